#!/usr/bin/env python3
"""Experiment (round 6): independent planes of a STREAM on TWO contexts / HIP streams of one GPU, alternately, each with seam
deferral -- against the same planes queued back to back on one stream.  A launch's idle tail (the CUs that finish first wait for
the slowest one: 2 % of a 3840x2160 step, more on small planes) and the launch boundary behind it could be filled by the other
lane's next kernel, whose workgroups are dispatched as soon as compute units free up.
usage: tools/two_lane_probe.py [WxH,WxH,...] [steps]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np
import torch

import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_batch

sizes = sys.argv[1] if len(sys.argv) > 1 else "3840x2160,1920x1080,1280x720,576x576"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
blob = S.load_weights()
PEAK = 157.3e12
for size in sizes.split(","):
    w, h = (int(v) for v in size.split("x"))
    frames = synth_batch(w, h, 4, first_frame=3)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    ctxs = [S.Context(0) for _ in range(2)]
    streams = [torch.cuda.Stream() for _ in range(2)]
    for c, st in zip(ctxs, streams):
        c.set_weights_blob(blob)
        c.set_stream(st.cuda_stream)
        c.set_seam_deferral(True)
    torch.cuda.synchronize()

    def run(n_lanes, n):
        for k in range(n):
            c = ctxs[k % n_lanes]
            f = k % 4
            c.forward_y_dev(d_in[f].data_ptr(), w, 0, d_out[f].data_ptr(), w, 0, w, h, 1)
        for c in ctxs[:n_lanes]:
            c.flush()

    res = {}
    for n_lanes in (1, 2, 1, 2):
        run(n_lanes, max(40, int(0.5 / 1e-3)))          # clock ramp
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            ev0.record(streams[0])
            if n_lanes == 2:
                streams[1].wait_event(ev0)
            run(n_lanes, steps)
            if n_lanes == 2:
                e2 = torch.cuda.Event()
                e2.record(streams[1])
                streams[0].wait_event(e2)
            ev1.record(streams[0])
            ev1.synchronize()
            best = min(best, ev0.elapsed_time(ev1) / steps)
        res.setdefault(n_lanes, []).append(best)
    ref = S.Context(0)
    ref.set_weights_blob(blob)
    want = np.stack([ref.forward_y(f) for f in frames])
    ok = bool(np.array_equal(d_out.cpu().numpy(), want))
    for n_lanes, v in res.items():
        ms = min(v)
        print(f"{size:>10} {n_lanes} lane(s): {ms:.4f} ms per plane = {w * h * 16064 / (ms * 1e-3) / PEAK:.4f} of the f32 MFMA peak  (runs: {', '.join(f'{x:.4f}' for x in v)})")
    print(f"{size:>10} outputs equal the one-context planes: {ok}")
    for c in ctxs + [ref]:
        c.close()
