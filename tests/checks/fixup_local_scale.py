#!/usr/bin/env python3
"""VERDICT r05 item 1, step A (CPU only): can SRCNN_MODE_REFBYTES flag against a PER-PIXEL threshold instead of one global delta?

The MFMA path's pre-truncation value v (oracle/srcnn_gpuorder.c: bitwise the kernels) differs from the reference's r
(oracle/srcnn_oracle.c) by rounding noise.  Today a pixel is flagged when |v - rint(v)| <= delta, ONE number per model sized from
the rigorous bound of the layer-2 map.  The noise of a pixel scales with ITS OWN activations; candidates for a local scale that the
strip kernel could carry at (almost) no MFMA cost:

  S1(x) = sum over the 5x5 window of U,  U = sum_c a_c * F_c,  a_c = max_tap |W3[c][tap]|   (one more layer-3 accumulator row + a box sum)
  S2(x) = sum_tap sum_c |W3[c][tap]| * F_c(x + tap)                                          (25 more rows: the exact abs-weight sum)

For every content class this prints the k a threshold k * 2^-24 * S1 + abs needs to stay 2.5 x above every deviation of the class
(content then stays below 0.4 thr, short of the 1/2 at which the device-side net redoes a launch: rule R2 of srcnn_ctx.h), for several absolute terms, and what a (k, abs)
pair then flags against the global threshold.  (S2, the exact abs-weight sum, tracks S1 within a few per cent on every class:
round 6's first run of this script; it would cost 25 accumulator rows instead of 5 idle ones.)  The adversarial side of the
same question: tests/checks/fixup_adversarial_ratio.py.
usage: fixup_local_scale.py [megapixels per class] [k abs]"""
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

delta_now = 1.376e-3          # fixup_delta() of the shipped model at the default margin 4 (tests/test_refbytes_model.py)
GAIN_CONTENT = 2.5            # rule R2 of srcnn_ctx.h: content stays below 0.4 thr (the global delta happens to keep 3.1 on content)
ABS_CANDIDATES = [4 * EPS * 256, 8 * EPS * 256, 12 * EPS * 256, 16 * EPS * 256, 24 * EPS * 256]
rows = []
for name, y in classes.items():
    F = oracle.gpuorder_conv99x11(y, w1, b1, w2, b2)
    g_out, g_pre = oracle.gpuorder_conv55(F, w3, b3)
    r_out, r_pre = oracle.forward_y(y, blob)
    S1, _ = scales(F)
    del F
    live = (g_pre > 0.5) & (g_pre < 255.5)
    d = np.abs(g_pre.astype(np.float64) - r_pre)[live]
    s1 = S1[live]
    rows.append((name, live.mean(), d, s1))
    kn = [(np.maximum(GAIN_CONTENT * d - a, 0) / (EPS * np.maximum(s1, 1e-30))).max() if d.size else 0.0 for a in ABS_CANDIDATES]
    print(f"{name:38s} live {live.mean():6.1%} max|d| {d.max() if d.size else 0:.2e}  mean S1 {s1.mean() if d.size else 0:9.1f}  "
          f"k needed (thr = k 2^-24 S1 + abs >= {GAIN_CONTENT} |d|) at abs = " + ", ".join(f"{a:.2e}: {k:.3f}" for a, k in zip(ABS_CANDIDATES, kn)), flush=True)

print("\nlargest k needed over all classes:")
for a in ABS_CANDIDATES:
    k = max((np.maximum(GAIN_CONTENT * d - a, 0) / (EPS * np.maximum(s1, 1e-30))).max() for _, _, d, s1 in rows if d.size)
    print(f"  abs {a:.3e}: k >= {k:.3f}")

# what a (k, abs) pair flags: the mean threshold over the live pixels of a class against the global delta (flagged fraction ~ 2 x mean threshold)
K_ABS = [(float(sys.argv[2]), float(sys.argv[3]))] if len(sys.argv) > 3 else [(2.4, 4 * EPS * 256), (1.8, 8 * EPS * 256), (1.6, 16 * EPS * 256)]
for k, a in K_ABS:
    print(f"\nthr = min(delta, {k} * 2^-24 * S1 + {a:.3e}):  mean threshold / delta per class (= flagged pixels against the global threshold's)")
    for name, lv, d, s1 in rows:
        if not d.size:
            continue
        thr = np.minimum(delta_now, k * EPS * s1 + a)
        print(f"    {name:38s} {thr.mean() / delta_now:.3f}    largest |d| / thr {np.max(d / thr):.3f}")
