#!/usr/bin/env python3
"""Randomised soak of the HIP path against the oracle (run on the GPU box; not part of pytest).

usage: python tests/checks/soak.py [seconds] [seed]
Random plane sizes (biased to strip / unit / item-planner boundaries), padded strides, three
content types; every result is checked: float32 MFMA mode bitwise against the FMA-order model and
within tolerance of the reference arithmetic, split-f16 mode within tolerance, exact mode bitwise,
the REFBYTES modes (float32 MFMA or split-f16 kernel + exact fix-up of the pixels next to a truncation boundary) BYTEWISE equal
to the reference arithmetic, with the margin their monitor reports.
"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import oracle, srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
blob = S.load_weights()
ctx = S.Context(0); ctx.set_weights_blob(blob)
TOL = 5e-3
edges = [1, 2, 5, 9, 31, 32, 33, 123, 124, 125, 127, 128, 129, 247, 248, 249, 372, 373, 496, 497, 620]
def dim(big):
    r = rng.random()
    if r < 0.5: return int(rng.choice(edges))
    return int(rng.integers(1, big))
n = 0; n_batches = 0; worst = {"mfma": 0.0, "split16": 0.0}; t0 = time.time()
while time.time() - t0 < budget:
    w, h = dim(900), dim(700)
    if rng.random() < 0.1: h = int(rng.integers(1500, 6000)); w = int(rng.choice([124, 125, 248, 300]))   # item planner
    kind = rng.integers(0, 3)
    if kind == 0: y = synth_luma(w, h, frame=int(rng.integers(0, 50)))
    elif kind == 1: y = rng.integers(0, 256, size=(h, w), dtype=np.uint8)
    else: y = np.full((h, w), int(rng.integers(0, 256)), np.uint8)
    stride = w + int(rng.integers(0, 3)) * int(rng.integers(0, 40))
    buf = np.zeros((h, stride), np.uint8); buf[:, :w] = y; yv = buf[:, :w]
    r_out, r_pre = oracle.forward_y(y, blob)
    m_out, m_pre = oracle.gpuorder_forward_y(y, blob)
    scale = max(1.0, float(np.abs(r_pre).max()) / 255.0)
    for name, mode in (("mfma", S.MODE_MFMA), ("split16", S.MODE_SPLIT16), ("exact", S.MODE_EXACT)):
        if name == "exact" and w * h > 200000: continue
        ctx.set_mode(mode)
        if name == "mfma":      # ... and the same plane in REFBYTES mode: the reference's bytes, no tolerance
            for rmode in (S.MODE_REFBYTES, S.MODE_REFBYTES16):
                ctx.set_mode(rmode)
                rb = ctx.forward_y(yv)
                assert np.array_equal(rb, r_out), ("refbytes", rmode, w, h, kind, int((rb != r_out).sum()))
            ctx.set_mode(mode)
        pre = np.full((h, w), np.nan, np.float32)
        out = ctx.forward_y(yv, preclamp=pre)
        assert np.isfinite(pre).all(), (name, w, h)
        if name == "exact":
            assert np.array_equal(out, r_out) and np.array_equal(pre, r_pre), (name, w, h, kind)
            continue
        if name == "mfma":
            assert np.array_equal(out, m_out) and np.array_equal(pre, m_pre), (name, w, h, kind)
        e = float(np.abs(pre - r_pre).max())
        worst[name] = max(worst[name], e / scale)
        assert e <= TOL * scale, (name, w, h, kind, e)
        d = np.abs(out.astype(int) - r_out.astype(int))
        assert d.max() <= 1, (name, w, h, kind)
        if d.any():
            assert np.abs(r_pre - np.rint(r_pre))[d != 0].max() <= TOL * scale, (name, w, h, kind)
    if n % 4 == 0:      # a batch of small planes through srcnn_forward_y_dev in a REFBYTES mode: ONE fix-up per 16 frames, odd pitches
        bw, bh, nf = int(rng.integers(5, 260)), int(rng.integers(3, 160)), int(rng.integers(2, 22))
        fr = []
        for k in range(nf):
            kk = rng.integers(0, 4)
            fr.append(synth_luma(bw, bh, frame=int(rng.integers(0, 50))) if kk == 0 else
                      rng.integers(0, 256, size=(bh, bw), dtype=np.uint8) if kk == 1 else
                      np.full((bh, bw), int(rng.choice([38, 54, int(rng.integers(0, 256))])), np.uint8) if kk == 2 else
                      np.where((np.add.outer(np.arange(bh) // 8, np.arange(bw) // 8) % 2) == 0, 16, 240).astype(np.uint8))
        ss, ds = bw + int(rng.integers(0, 9)), bw + int(rng.integers(0, 9))
        sp, dp = bh * ss + int(rng.integers(0, 100)), bh * ds + int(rng.integers(0, 100))
        d_in = torch.zeros(nf * sp, dtype=torch.uint8, device="cuda")
        for k in range(nf):
            d_in[k * sp:k * sp + bh * ss].view(bh, ss)[:, :bw] = torch.from_numpy(fr[k]).cuda()
        d_out = torch.full((nf * dp,), 7, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.set_mode(S.MODE_REFBYTES if (n // 4) % 2 == 0 else S.MODE_REFBYTES16)
        ctx.forward_y_dev(d_in.data_ptr(), ss, sp, d_out.data_ptr(), ds, dp, bw, bh, nf)
        ctx.synchronize()
        got = d_out.cpu().numpy()
        for k in range(nf):
            plane = got[k * dp:k * dp + bh * ds].reshape(bh, ds)
            assert np.array_equal(plane[:, :bw], oracle.forward_y(fr[k], blob)[0]), ("refbytes batch", bw, bh, nf, k)
            assert (plane[:, bw:] == 7).all() and (got[k * dp + bh * ds:(k + 1) * dp] == 7).all(), ("wrote outside", bw, bh, nf, k)
        n_batches += 1
    n += 1
ctx.set_mode(S.MODE_MFMA)
st = ctx.fixup_stats()
assert st["max_dev"] < 0.5 * st["delta"], st
k_eff, ratio = ctx.fixup_local_stats() if hasattr(ctx, "fixup_local_stats") else (0.0, 0.0)
assert ratio < 0.5 or st["exact_reruns"] >= 1, (k_eff, ratio, st)      # above half a pixel's own threshold the net must have acted
print(f"refbytes: every plane bytewise equal to the reference arithmetic; {st['scattered_pixels']} pixels recomputed one by one, "
      f"{st['dense_tiles']} tiles whole, {st['bytes_changed']} bytes changed; largest |v_mfma - v_ref| seen {st['max_dev']:.2e} "
      f"against delta {st['delta']:.2e}; largest deviation over the pixel's own threshold (k = {k_eff:.2f}) {ratio:.3f}; launches redone by the net: {st['exact_reruns']}")
print(f"soak ok: {n} random planes and {n_batches} REFBYTES batches of 2-21 frames in {time.time() - t0:.0f} s (seed {seed}); worst pre-clamp error / max(1, |ref|max/255): "
      f"mfma {worst['mfma']:.2e}, split16 {worst['split16']:.2e}")
