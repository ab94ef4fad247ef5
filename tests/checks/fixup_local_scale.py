#!/usr/bin/env python3
"""VERDICT r05 item 1, step A (CPU only): can SRCNN_MODE_REFBYTES flag against a PER-PIXEL threshold instead of one global delta?

The MFMA path's pre-truncation value v (oracle/srcnn_gpuorder.c: bitwise the kernels) differs from the reference's r
(oracle/srcnn_oracle.c) by rounding noise.  Today a pixel is flagged when |v - rint(v)| <= delta, ONE number per model sized from
the rigorous bound of the layer-2 map.  The noise of a pixel scales with ITS OWN activations; candidates for a local scale that the
strip kernel could carry at (almost) no MFMA cost:

  S1(x) = sum over the 5x5 window of U,  U = sum_c a_c * F_c,  a_c = max_tap |W3[c][tap]|   (one more layer-3 accumulator row + a box sum)
  S2(x) = sum_tap sum_c |W3[c][tap]| * F_c(x + tap)                                          (25 more rows: the exact abs-weight sum)

For every content class and for the adversarial windows this prints the largest |v - r| / (2^-24 * S) -- the factor k a threshold
k * 2^-24 * S + abs would need -- and, for thresholds with the SAME safety factor over the worst observed ratio as today's global
delta has over the worst observed deviation, the fraction of pixels flagged against today's.
usage: fixup_local_scale.py [megapixels per class]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

mp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
w = 1920
h = max(64, int(mp * 1e6 / w))
blob = S.load_weights()
w1, b1, w2, b2, w3, b3 = oracle.split_weights(blob)
rng = np.random.default_rng(7)
yy, xx = np.mgrid[0:h, 0:w]
EPS = 2.0 ** -24
ABS = 4 * EPS * 256            # the absolute term of fixup_delta()
a_c = np.abs(w3).reshape(32, 25).max(axis=1)
absw3 = np.abs(w3)             # [32][5][5]


def shift(a, dy, dx):
    """a[y + dy, x + dx] with replicate border"""
    hh, ww = a.shape
    ys = np.clip(np.arange(hh) + dy, 0, hh - 1)
    xs = np.clip(np.arange(ww) + dx, 0, ww - 1)
    return a[np.ix_(ys, xs)]


def scales(F):
    """F [32][h][w] (the kernels' layer-2 map) -> S1, S2"""
    U = np.tensordot(a_c, F, axes=(0, 0))
    S1 = sum(shift(U, m - 2, n - 2) for m in range(5) for n in range(5))
    S2 = np.zeros_like(U)
    for m in range(5):
        for n in range(5):
            S2 += shift(np.tensordot(absw3[:, m, n], F, axes=(0, 0)), m - 2, n - 2)
    S3 = np.zeros_like(U)
    for m in range(5):
        Um = np.tensordot(absw3[:, m, :].max(axis=1), F, axes=(0, 0))
        for n in range(5):
            S3 += shift(Um, m - 2, n - 2)
    return S1.astype(np.float64), S3.astype(np.float64)


def smooth(sigma_px, amp):
    f = rng.standard_normal((h, w)).astype(np.float32)
    Fq = np.fft.rfft2(f)
    ky = np.fft.fftfreq(h)[:, None]
    kx = np.fft.rfftfreq(w)[None, :]
    Fq *= np.exp(-0.5 * (ky ** 2 + kx ** 2) * (2 * np.pi * sigma_px) ** 2)
    g = np.fft.irfft2(Fq, s=(h, w))
    g = g / np.abs(g).max()
    return np.clip(128 + amp * g, 0, 255).astype(np.uint8)


classes = {
    "synthetic (bench generator)": synth_luma(w, h, frame=11),
    "band-limited sigma 6, full range": smooth(6, 127),
    "band-limited sigma 2, full range": smooth(2, 127),
    "band-limited sigma 12 + 4-bit noise": np.clip(smooth(12, 100).astype(int) + rng.integers(0, 16, (h, w)), 0, 255).astype(np.uint8),
    "white noise 0..255": rng.integers(0, 256, (h, w), dtype=np.uint8),
    "white noise 96..160": rng.integers(96, 161, (h, w), dtype=np.uint8),
    "checkerboard 8 px, 16/240": np.where(((yy // 8) + (xx // 8)) % 2 == 0, 16, 240).astype(np.uint8),
    "bright ramp 200..255 + 2-bit noise": np.clip(200 + (xx * 55 // w) + rng.integers(0, 4, (h, w)), 0, 255).astype(np.uint8),
    "text-like: sparse 255 on 30": np.where(rng.random((h, w)) < 0.03, 255, 30).astype(np.uint8),
    "flat 128": np.full((h, w), 128, np.uint8),
    "dark: 0..15 noise": rng.integers(0, 16, (h, w), dtype=np.uint8),
}

delta_now = float(S.fixup_delta(blob)) if hasattr(S, "fixup_delta") else 1.376e-3
rows = []
tot = {}
for name, y in classes.items():
    F = oracle.gpuorder_conv99x11(y, w1, b1, w2, b2)
    g_out, g_pre = oracle.gpuorder_conv55(F, w3, b3)
    r_out, r_pre = oracle.forward_y(y, blob)
    S1, S2 = scales(F)
    del F
    live = (g_pre > 0.5) & (g_pre < 255.5)
    d = np.abs(g_pre.astype(np.float64) - r_pre)[live]
    s1, s2 = S1[live], S2[live]
    rows.append((name, live.mean(), d, s1, s2))
    k1 = (np.maximum(d - ABS, 0) / (EPS * np.maximum(s1, 1e-30)))
    k2 = (np.maximum(d - ABS, 0) / (EPS * np.maximum(s2, 1e-30)))
    print(f"{name:38s} live {live.mean():6.1%} max|d| {d.max() if d.size else 0:.2e}  mean S1 {s1.mean() if d.size else 0:9.1f} S2 {s2.mean() if d.size else 0:9.1f}  "
          f"max k1 {k1.max() if d.size else 0:6.3f} p99.99 {np.quantile(k1, 0.9999) if d.size else 0:6.3f}  max k2 {k2.max() if d.size else 0:6.3f} p99.99 {np.quantile(k2, 0.9999) if d.size else 0:6.3f}",
          flush=True)

# the adversarial windows (tests/golden/): the centre pixel of a 13 x 13 window; its 5 x 5 feature window lies inside the window's own map
GOLD = Path(__file__).resolve().parent.parent / "golden"
adv = []
for fn in ("adversarial_windows.npz", "adversarial_windows_gpu.npz"):
    z = np.load(GOLD / fn)
    for key in z.files:
        a = z[key]
        if a.dtype == np.uint8 and a.ndim == 3 and a.shape[1:] == (13, 13):
            for win in a:
                adv.append((fn + ":" + key, win))
print(f"{len(adv)} adversarial windows")
adv_rows = []
for tag, win in adv:
    vr, vg = oracle.adv_point(win, blob)
    F = oracle.gpuorder_conv99x11(win, w1, b1, w2, b2)[:, 4:9, 4:9]
    s1 = float((a_c[:, None, None] * F).sum())
    s2 = float((absw3.max(axis=2)[:, :, None] * F).sum())
    adv_rows.append((tag, abs(vg - vr), s1, s2))
ad = np.array([r[1] for r in adv_rows]); a1 = np.array([r[2] for r in adv_rows]); a2 = np.array([r[3] for r in adv_rows])
print(f"adversarial: max|d| {ad.max():.2e}  S1 at worst {a1[ad.argmax()]:.1f}  S2 at worst {a2[ad.argmax()]:.1f}  "
      f"max k1 {(np.maximum(ad - ABS, 0) / (EPS * a1)).max():.3f}  max k2 {(np.maximum(ad - ABS, 0) / (EPS * a2)).max():.3f}")

# Thresholds with today's safety: the global delta is 3.1 x the worst content deviation and 1.73 x the adversarial worst.
D = np.concatenate([r[2] for r in rows]); A1 = np.concatenate([r[3] for r in rows]); A2 = np.concatenate([r[4] for r in rows])
for label, sc, asc in (("S1", A1, a1), ("S2", A2, a2)):
    kc = (np.maximum(D - ABS, 0) / (EPS * np.maximum(sc, 1e-30))).max()
    ka = (np.maximum(ad - ABS, 0) / (EPS * asc)).max()
    k = max(3.1 * kc, 1.73 * ka)
    print(f"{label}: worst k on content {kc:.3f}, adversarial {ka:.3f} -> k = {k:.3f} (3.1 x content, 1.73 x adversarial)")
    for name, lv, d, s1, s2 in rows:
        s = s1 if label == "S1" else s2
        thr = k * EPS * s + ABS
        print(f"    {name:38s} mean threshold {thr.mean():.3e} vs delta {delta_now:.3e}: flagged x {thr.mean() / delta_now:.3f}   (capped at delta: x {np.minimum(thr, delta_now).mean() / delta_now:.3f})")
