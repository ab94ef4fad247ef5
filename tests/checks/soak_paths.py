#!/usr/bin/env python3
"""Randomised soak of the round-2 launch paths (run on the GPU box; not part of pytest).

usage: python tests/checks/soak_paths.py [seconds] [seed]
Random MID-SIZE planes (so that the work-item planner, row / column seams, item batches, row bands and the aligned
Convolution55 launch all engage with odd geometries); every result is compared BITWISE with the CPU model of the
kernels' arithmetic (oracle.gpuorder_*):
  * srcnn_forward_y with and without the pre-clamp plane (the latter pipelines row bands on large planes)
  * a batch of 2-5 frames through srcnn_forward_y_dev (item plan repeated per frame) with padded strides / pitches
  * the same plane row-striped over 2-3 contexts (srcnn_forward_y_striped)
  * Convolution99x11 -> Convolution55 through the device entry points (aligned 128-column strips in the second)
"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import oracle, srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 180.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
blob = S.load_weights()
ctxs = [S.Context(0) for _ in range(3)]
for c in ctxs:
    c.set_weights_blob(blob)
ctx = ctxs[0]
n, t0 = 0, time.time()
while time.time() - t0 < budget:
    w = int(rng.choice([rng.integers(130, 4100), rng.choice([128, 256, 260, 1020, 1024, 1028, 1920, 2047, 2052, 3840])]))
    h = int(rng.choice([rng.integers(40, 2300), rng.choice([64, 540, 1024, 1080, 2047, 2160])]))
    if w * h > 9_000_000:
        h = 9_000_000 // w
    y = synth_luma(w, h, frame=int(rng.integers(0, 99)))
    m_out, m_pre = oracle.gpuorder_forward_y(y, blob)
    pre = np.empty((h, w), np.float32)
    assert np.array_equal(ctx.forward_y(y, preclamp=pre), m_out) and np.array_equal(pre, m_pre), ("forward_y+pre", w, h)
    assert np.array_equal(ctx.forward_y(y), m_out), ("forward_y bands", w, h)
    # small batch on the device, padded strides
    nf = int(rng.integers(2, 6))
    if w * h * nf <= 24_000_000:
        pad = int(rng.integers(0, 3)) * 32
        frames = [y] + [synth_luma(w, h, frame=100 + k) for k in range(nf - 1)]
        d_in = torch.zeros((nf, h + 1, w + pad), dtype=torch.uint8, device="cuda")
        for k in range(nf):
            d_in[k, :h, :w] = torch.from_numpy(frames[k]).cuda()
        d_out = torch.full_like(d_in, 3)
        torch.cuda.synchronize()
        ctx.forward_y_dev(d_in.data_ptr(), w + pad, (h + 1) * (w + pad), d_out.data_ptr(), w + pad, (h + 1) * (w + pad), w, h, nf)
        ctx.synchronize()
        got = d_out.cpu().numpy()
        assert np.array_equal(got[0, :h, :w], m_out), ("batch frame 0", w, h, nf)
        k = nf - 1
        assert np.array_equal(got[k, :h, :w], oracle.gpuorder_forward_y(frames[k], blob)[0]), ("batch last frame", w, h, nf)
        assert (got[:, h, :] == 3).all() and (got[:, :, w:] == 3).all(), ("wrote outside", w, h, nf)
    # row stripes over contexts
    nc = int(rng.integers(2, 4))
    if h // nc >= 6:
        assert np.array_equal(S.forward_y_striped(ctxs[:nc], y), m_out), ("striped", w, h, nc)
    # the two reference functions on device memory
    if w * h <= 4_500_000:
        d_y = torch.from_numpy(y).cuda()
        d_planes = torch.empty((32, h, w), dtype=torch.float32, device="cuda")
        d_o = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.conv99x11_dev(d_y.data_ptr(), w, d_planes.data_ptr(), w, h * w, w, h)
        ctx.conv55_dev(d_planes.data_ptr(), w, h * w, d_o.data_ptr(), w, w, h)
        ctx.synchronize()
        assert np.array_equal(d_o.cpu().numpy(), m_out), ("conv99x11 -> conv55", w, h)
    n += 1
print(f"soak_paths ok: {n} random planes in {time.time() - t0:.0f} s (seed {seed})")
