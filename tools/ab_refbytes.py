#!/usr/bin/env python3
"""Same-box A/B of SRCNN_MODE_REFBYTES' fix-up: ms per step (HIP events around K back-to-back steps on one stream) of the MFMA
mode and of REFBYTES at several threshold factors, with and without the device-side safety net, for ONE build of the library.
    python tools/ab_refbytes.py [--lib PATH] [--sizes 3840x2160,1920x1080] [--margins 4,6] [--steps 40]
Run it once per library (the binding is per process) and compare inside one gpurun call: boxes differ by 2-3 %."""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--sizes", default="3840x2160,1920x1080,7680x4320")
ap.add_argument("--margins", default="4,6")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--modes", default="refbytes")
ap.add_argument("--locals", default="0.4,0", help="per-pixel threshold factors to time (srcnn_set_fixup_local; 0 = the global threshold only)")
args = ap.parse_args()
if args.lib:
    S.use_library(args.lib)
blob = S.load_weights()
stream = torch.cuda.Stream()


def timed(fn, steps):
    for _ in range(max(10, int(0.4 / 1e-3 / 4))):      # clock ramp
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        for _ in range(steps):
            fn()
        b.record(stream)
        b.synchronize()
        best = min(best, a.elapsed_time(b) / steps)
    return best


with S.Context(0) as ctx:
    ctx.set_weights_blob(blob)
    ctx.set_stream(stream.cuda_stream)
    print(f"# library {S.library_path().name}")
    for size in args.sizes.split(","):
        w, h = (int(v) for v in size.split("x"))
        d_in = torch.from_numpy(synth_luma(w, h)).cuda()
        d_out = torch.zeros_like(d_in)
        torch.cuda.synchronize()
        step = lambda: ctx.forward_y_dev(d_in.data_ptr(), w, 0, d_out.data_ptr(), w, 0, w, h, 1)
        ctx.set_mode(S.MODE_MFMA)
        t0 = timed(step, args.steps)
        print(f"{size:>10} mfma                          {t0:8.4f} ms")
        for mode in args.modes.split(","):
            ctx.set_mode({"refbytes": S.MODE_REFBYTES, "refbytes16": S.MODE_REFBYTES16}[mode])
            for m in (float(v) for v in args.margins.split(",")):
                ctx.set_fixup_margin(m)
                for kl in ((float(v) for v in args.locals.split(",")) if hasattr(ctx, "set_fixup_local") else (None,)):
                    if kl is not None:
                        ctx.set_fixup_local(kl)
                    for strict in (True, False):
                        ctx.set_fixup_strict(strict)
                        before = ctx.fixup_stats()
                        calls = max(10, int(0.4 / 1e-3 / 4)) + 3 * args.steps
                        t = timed(step, args.steps)
                        st = ctx.fixup_stats()
                        n = (st["scattered_pixels"] - before["scattered_pixels"]) / calls
                        ratio = ctx.fixup_local_stats()[1] if kl is not None else float("nan")
                        print(f"{size:>10} {mode:<10} margin {m:g} local {kl if kl is not None else '-'} strict {int(strict)}  {t:8.4f} ms  x{t / t0:.3f}  "
                              f"(+{(t - t0) * 1e3:6.1f} us; {n:8.0f} px flagged per plane, delta {st['delta']:.3e}, max_dev {st['max_dev']:.2e}, "
                              f"max ratio {ratio:.3f}, reruns {st['exact_reruns']})")
        ctx.set_mode(S.MODE_MFMA)
