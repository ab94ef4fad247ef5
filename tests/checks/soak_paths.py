#!/usr/bin/env python3
"""Randomised soak of the round-2 launch paths (run on the GPU box; not part of pytest).

usage: python tests/checks/soak_paths.py [seconds] [seed]
Random MID-SIZE planes (so that the work-item planner, row / column seams, item batches, row bands and the aligned
Convolution55 launch all engage with odd geometries); every result is compared BITWISE with the CPU model of the
kernels' arithmetic (oracle.gpuorder_*):
  * srcnn_forward_y with and without the pre-clamp plane (the latter pipelines row bands on large planes)
  * a batch of 2-5 frames through srcnn_forward_y_dev (item plan repeated per frame) with padded strides / pitches
  * the same plane row-striped over 2-3 contexts (srcnn_forward_y_striped: one launch per stripe, neighbours' rows read in place)
  * the same plane as 2-8 stripes through srcnn_forward_y_rows_halo_dev with the halo rows in separately allocated buffers of
    another row stride -- in the MFMA mode (against the model) and in SRCNN_MODE_REFBYTES (against the reference arithmetic),
    in strict mode every now and then with the threshold cut below the noise (the exact re-run path)
  * Convolution99x11 -> Convolution55 through the device entry points (aligned 128-column strips in the second)
  * seam deferral (round 5): a random stream of whole-plane and row-range launches on two planes of different sizes, outputs
    shared or fresh, every one complete and bitwise right after flush()
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
import os
os.makedirs("gpurun_out", exist_ok=True)
progress = open("gpurun_out/soak_paths_progress.txt", "w")          # what was running when a launch faulted
def note(*a):
    progress.write(" ".join(map(str, a)) + "\n"); progress.flush()
while time.time() - t0 < budget:
    w = int(rng.choice([rng.integers(130, 4100), rng.choice([128, 256, 260, 1020, 1024, 1028, 1920, 2047, 2052, 3840])]))
    h = int(rng.choice([rng.integers(40, 2300), rng.choice([64, 540, 1024, 1080, 2047, 2160])]))
    if w * h > 9_000_000:
        h = 9_000_000 // w
    y = synth_luma(w, h, frame=int(rng.integers(0, 99)))
    note("plane", n, w, h)
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
    # stripes with their halo rows in buffers of their own, both float32 modes
    ns = int(rng.integers(2, 9))
    if h // ns >= 6:
        r_out = oracle.forward_y(y, blob)[0] if w * h <= 1_500_000 else None
        for mode in ((S.MODE_MFMA, S.MODE_REFBYTES) if r_out is not None else (S.MODE_MFMA,)):
            ctx.set_mode(mode)
            strict = mode == S.MODE_REFBYTES and rng.random() < 0.3
            note("halo stripes", w, h, ns, mode, strict)
            if strict:
                ctx.set_fixup_margin(0.25)              # below the noise: the device-side re-run takes over (strict is the default)
            d_o = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
            keep = []
            for k in range(ns):
                r0, r1 = S.stripe_rows(h, ns, k)
                hs = w + int(rng.integers(0, 3)) * 16
                own = torch.from_numpy(np.ascontiguousarray(y[r0:r1])).cuda()
                top = bot = None
                if k > 0:
                    top = torch.zeros((6, hs), dtype=torch.uint8, device="cuda"); top[:, :w] = torch.from_numpy(np.ascontiguousarray(y[r0 - 6:r0])).cuda()
                if k < ns - 1:
                    bot = torch.zeros((6, hs), dtype=torch.uint8, device="cuda"); bot[:, :w] = torch.from_numpy(np.ascontiguousarray(y[r1:r1 + 6])).cuda()
                keep += [own, top, bot]
                torch.cuda.synchronize()
                ctx.forward_y_rows_halo_dev(own.data_ptr(), w, r0, r1 - r0, top.data_ptr() if top is not None else 0,
                                            bot.data_ptr() if bot is not None else 0, hs, d_o.data_ptr(), w, 0, w, h, r0, r1)
            ctx.synchronize()
            want = m_out if mode == S.MODE_MFMA else r_out
            assert np.array_equal(d_o.cpu().numpy(), want), ("halo stripes", mode, strict, w, h, ns)
            if strict:
                ctx.set_fixup_margin(4.0)
        ctx.set_mode(S.MODE_MFMA)
    # seam deferral: a stream of launches on this plane and a second one of another size -- whole planes and row ranges, into
    # one shared output buffer per plane or a fresh one, interleaved -- every output complete after flush()
    if w * h <= 4_500_000:
        note("deferral stream", w, h)
        w2, h2 = int(rng.choice([576, 960, 1280, 1920, w])), int(rng.choice([360, 540, 720, h]))
        y2 = synth_luma(w2, h2, frame=int(rng.integers(0, 99)))
        m2 = oracle.gpuorder_forward_y(y2, blob)[0]
        d1, d2 = torch.from_numpy(y).cuda(), torch.from_numpy(y2).cuda()
        o1, o2 = torch.zeros_like(d1), torch.zeros_like(d2)
        fresh = []
        torch.cuda.synchronize()
        ctx.set_seam_deferral(True)
        for _ in range(int(rng.integers(3, 8))):
            which = rng.integers(0, 4)
            if which == 0:
                ctx.forward_y_dev(d1.data_ptr(), w, 0, o1.data_ptr(), w, 0, w, h, 1)
            elif which == 1:
                ctx.forward_y_dev(d2.data_ptr(), w2, 0, o2.data_ptr(), w2, 0, w2, h2, 1)
            elif which == 2 and h >= 40:
                r0 = int(rng.integers(0, h - 20)); r1 = int(rng.integers(r0 + 7, h + 1))
                s0, s1 = max(0, r0 - 6), min(h, r1 + 6)
                ctx.forward_y_rows_dev(d1[s0:s1].data_ptr(), w, s0, o1.data_ptr(), w, 0, w, h, r0, r1)      # rows of plane 1 over its own output
            else:
                t = torch.zeros_like(d2)
                fresh.append(t)
                ctx.forward_y_dev(d2.data_ptr(), w2, 0, t.data_ptr(), w2, 0, w2, h2, 1)
        ctx.forward_y_dev(d1.data_ptr(), w, 0, o1.data_ptr(), w, 0, w, h, 1)
        ctx.forward_y_dev(d2.data_ptr(), w2, 0, o2.data_ptr(), w2, 0, w2, h2, 1)
        ctx.flush()
        ctx.synchronize()
        ctx.set_seam_deferral(False)
        assert np.array_equal(o1.cpu().numpy(), m_out), ("deferral stream, plane 1", w, h, w2, h2)
        assert np.array_equal(o2.cpu().numpy(), m2), ("deferral stream, plane 2", w, h, w2, h2)
        for t in fresh:
            assert np.array_equal(t.cpu().numpy(), m2), ("deferral stream, fresh output", w, h, w2, h2)
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
