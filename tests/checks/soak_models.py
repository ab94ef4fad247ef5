#!/usr/bin/env python3
"""SRCNN_MODE_REFBYTES with RANDOM MODELS (run on the GPU box; not part of pytest): the flag threshold follows the model
(fixup_delta(): 4 * 2^-24 * ||W3||_2 * bound of the layer-2 map + 4 * 2^-24 * 256), so other weights must give the reference's
bytes too, with the monitored deviation well inside the threshold -- in SRCNN_MODE_REFBYTES and, where the model fits the
f16 ranges of that mode, in SRCNN_MODE_REFBYTES16.  usage: python tests/checks/soak_models.py [seconds] [seed]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch  # noqa: F401
import os
import oracle, srcnn_cpp_amd as S
os.makedirs("gpurun_out", exist_ok=True)
from srcnn_cpp_amd.synth import synth_luma

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
n, n16, worst, worst16, reruns, t0 = 0, 0, 0.0, 0.0, 0, time.time()
oracle_flakes = []
while time.time() - t0 < budget:
    s1, s2, s3 = rng.uniform(0.03, 0.25), rng.uniform(0.03, 0.4), rng.uniform(0.005, 0.08)
    w1 = (rng.standard_normal(5184) * s1).astype(np.float32)
    b1 = (rng.standard_normal(64) * rng.uniform(1, 60)).astype(np.float32)
    w2 = (rng.standard_normal(2048) * s2).astype(np.float32)
    b2 = (rng.standard_normal(32) * rng.uniform(1, 30)).astype(np.float32)
    w3 = (rng.standard_normal(800) * s3).astype(np.float32)
    b3 = np.float32(rng.uniform(0, 200))
    if rng.random() < 0.3:                       # sparse / sign-structured models
        w2[rng.random(2048) < 0.5] = 0
        w3 = np.abs(w3) * np.float32(0.3)
    blob = np.concatenate([b1, w1, b2, w2, [b3], w3]).astype(np.float32)
    w, h = int(rng.integers(40, 700)), int(rng.integers(30, 500))
    kind = rng.integers(0, 3)
    y = synth_luma(w, h, frame=int(rng.integers(0, 50))) if kind == 0 else (
        rng.integers(0, 256, (h, w), dtype=np.uint8) if kind == 1 else np.full((h, w), int(rng.integers(0, 256)), np.uint8))
    r_out, r_pre = oracle.forward_y(y, blob)
    with S.Context(0) as ctx:
        ctx.set_weights(w1, b1, w2, b2, w3, b3)
        for mode in (S.MODE_REFBYTES, S.MODE_REFBYTES16):
            ctx.set_mode(mode)
            try:
                out = ctx.forward_y(y)
            except S.SrcnnError as e:
                assert mode == S.MODE_REFBYTES16 and e.code == S.ERR_STATE, e      # the model exceeds the f16 ranges: refused
                continue
            st = ctx.fixup_stats()
            if not np.array_equal(out, r_out):
                # Before blaming the kernels, ask the CHECKER again.  Round 5 met, twice in ~11,000 planes on the GPU boxes, an
                # oracle.forward_y whose result was wrong in one band of rows: a second run of the same call, the oracle recomputed
                # in the build container and the GPU's bytes all agreed with each other (profiles/r05/soak_long.txt).  Such a plane
                # is counted and reported, and judged by the second run; a result that differs from BOTH runs is the product's.
                r2 = oracle.forward_y(y, blob)[0]
                if np.array_equal(out, r2):
                    ys = np.nonzero((r_out != r2).any(axis=1))[0]
                    oracle_flakes.append((n, w, h, int((r_out != r2).sum()), int(ys.min()), int(ys.max())))
                    r_out = r2
                else:
                    ys, xs = np.nonzero(out != r2)
                    again = ctx.forward_y(y)
                    np.savez("gpurun_out/soak_models_failure.npz", blob=blob, y=y, out=out, r_out=r_out, r2=r2, again=again)
                    raise AssertionError((mode, n, w, h, int(kind), s1, s2, s3, len(ys), st, "rows", int(ys.min()), int(ys.max()), "cols", int(xs.min()),
                                          int(xs.max()), "repeat call differs from the oracle in", int((again != r2).sum()),
                                          "the two oracle runs differ in", int((r2 != r_out).sum())))
            # round 6: the largest deviation over the flagged pixel's OWN threshold (srcnn_fixup_local_stats); above 1/2 the
            # device-side net must have redone the launch (the bytes above are already checked)
            ratio = ctx.fixup_local_stats()[1]
            assert ratio < 0.5 or st["exact_reruns"] >= 1, (mode, n, st, ratio, s1, s2, s3)
            reruns += int(st["exact_reruns"] >= 1)
            if mode == S.MODE_REFBYTES:
                worst = max(worst, ratio)
            else:
                worst16, n16 = max(worst16, ratio), n16 + 1
    n += 1
print(f"soak_models ok: {n} random models x planes in {time.time() - t0:.0f} s ({n16} of them also in REFBYTES16); every plane bytewise equal "
      f"to the reference arithmetic; largest monitored deviation / the pixel's own threshold = {worst:.3f} (REFBYTES16: {worst16:.3f}); "
      f"{reruns} launches were over half a threshold and redone by the device-side net"
      + (f"; ORACLE ANOMALIES (first run of oracle.forward_y disagreed with a second run AND with the GPU; (n, w, h, bytes, rows)): {oracle_flakes}" if oracle_flakes else "")
      + f"; the oracle wrapper's own double runs disagreed {oracle.anomalies} times")
