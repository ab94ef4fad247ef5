#!/usr/bin/env python3
"""Does the CHECKER agree with itself on this box?  (round 5: tests/checks/soak_models.py twice met an `oracle.forward_y` whose
result differed from a second run of the same call in one band of rows -- the GPU output equalled the second run and the
oracle recomputed elsewhere.)  Runs the oracle twice on random planes and models and compares, with and without GPU work of the
library in between, so that a flaky host (or a library that scribbles over host memory) shows up as what it is.

usage: python tests/checks/oracle_selfcheck.py SECONDS [--gpu] [--threads N]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
use_gpu = "--gpu" in sys.argv
if "--threads" in sys.argv:
    oracle.set_threads(int(sys.argv[sys.argv.index("--threads") + 1]))
if use_gpu:
    import srcnn_cpp_amd as S
rng = np.random.default_rng(7)
n = bad = 0
t0 = time.time()
while time.time() - t0 < budget:
    s1, s2, s3 = rng.uniform(0.03, 0.25), rng.uniform(0.03, 0.4), rng.uniform(0.005, 0.08)
    blob = np.concatenate([(rng.standard_normal(64) * 20), rng.standard_normal(5184) * s1, rng.standard_normal(32) * 10,
                           rng.standard_normal(2048) * s2, [rng.uniform(0, 200)], rng.standard_normal(800) * s3]).astype(np.float32)
    w, h = int(rng.integers(40, 700)), int(rng.integers(30, 500))
    y = rng.integers(0, 256, (h, w), dtype=np.uint8)
    a = oracle.forward_y(y, blob)[0]
    if use_gpu:
        w1, b1, w2, b2, w3, b3 = S.split_weights(blob)
        with S.Context(0) as ctx:
            ctx.set_weights(w1, b1, w2, b2, w3, b3)
            ctx.set_mode(S.MODE_REFBYTES)
            g = ctx.forward_y(y)
    b = oracle.forward_y(y, blob)[0]
    if not np.array_equal(a, b):
        bad += 1
        ys = np.nonzero((a != b).any(axis=1))[0]
        print(f"iteration {n}: {w}x{h}: two oracle runs differ in {int((a != b).sum())} bytes, rows {ys.min()}..{ys.max()} ({len(ys)} rows)"
              + (f"; GPU equals run 1: {bool(np.array_equal(g, a))}, run 2: {bool(np.array_equal(g, b))}" if use_gpu else ""), flush=True)
    n += 1
print(f"oracle_selfcheck: {n} planes in {time.time() - t0:.0f} s, {'with' if use_gpu else 'WITHOUT'} GPU work in between: {bad} planes on which two oracle runs disagreed")
