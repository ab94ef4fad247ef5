#!/usr/bin/env python3
"""How far can the MFMA path's pre-truncation value be from the reference's?  (CPU only: the model of the kernels'
arithmetic, oracle/srcnn_gpuorder.c, is bitwise the GPU; oracle/srcnn_oracle.c is the reference arithmetic.)
Prints, per content class, max / quantiles of |v_gpu - v_ref| over the pixels whose value can still change a byte
(0.5 < v < 255.5), and how many pixels a fix-up threshold delta would flag.  This is what SRCNN_MODE_REFBYTES' delta
(srcnn_model.cpp: fixup_delta()) is chosen from: profiles/r03/fixup_margin.txt.
usage: fixup_margin.py [megapixels per class]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

mp = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
w = 1920
h = max(64, int(mp * 1e6 / w))
blob = S.load_weights()
rng = np.random.default_rng(7)
yy, xx = np.mgrid[0:h, 0:w]


def smooth(sigma_px, amp):
    """band-limited noise: natural-image-like 1/f content"""
    f = rng.standard_normal((h, w)).astype(np.float32)
    F = np.fft.rfft2(f)
    ky = np.fft.fftfreq(h)[:, None]
    kx = np.fft.rfftfreq(w)[None, :]
    F *= np.exp(-0.5 * (ky ** 2 + kx ** 2) * (2 * np.pi * sigma_px) ** 2)
    g = np.fft.irfft2(F, s=(h, w))
    g = g / np.abs(g).max()
    return np.clip(128 + amp * g, 0, 255).astype(np.uint8)


classes = {
    "synthetic (bench generator)": synth_luma(w, h, frame=11),
    "band-limited noise, sigma 6 px, full range": smooth(6, 127),
    "band-limited noise, sigma 2 px, full range": smooth(2, 127),
    "band-limited sigma 12 + 4-bit noise": np.clip(smooth(12, 100).astype(int) + rng.integers(0, 16, (h, w)), 0, 255).astype(np.uint8),
    "white noise 0..255 (saturates the output)": rng.integers(0, 256, (h, w), dtype=np.uint8),
    "white noise 96..160": rng.integers(96, 161, (h, w), dtype=np.uint8),
    "checkerboard 8 px, 16/240": np.where(((yy // 8) + (xx // 8)) % 2 == 0, 16, 240).astype(np.uint8),
    "bright ramp 200..255 + 2-bit noise": np.clip(200 + (xx * 55 // w) + rng.integers(0, 4, (h, w)), 0, 255).astype(np.uint8),
    "text-like: sparse 255 strokes on 30": np.where(rng.random((h, w)) < 0.03, 255, 30).astype(np.uint8),
}
worst = 0.0
for name, y in classes.items():
    g_out, g_pre = oracle.gpuorder_forward_y(y, blob)
    r_out, r_pre = oracle.forward_y(y, blob)
    live = (g_pre > 0.5) & (g_pre < 255.5)
    d = np.abs(g_pre - r_pre)[live]
    frac = np.abs(g_pre - np.rint(g_pre))[live]
    mism = int((g_out != r_out).sum())
    q = np.quantile(d, [0.5, 0.99, 0.9999]) if d.size else [0, 0, 0]
    flagged = {dl: float((frac <= dl).mean()) if d.size else 0.0 for dl in (5e-4, 1e-3, 2e-3, 3.5e-3)}
    # the mismatching bytes must all be flagged by delta: distance of v_gpu to an integer at the mismatches
    mm = np.abs(g_pre - np.rint(g_pre))[g_out != r_out]
    print(f"{name:45s} live {live.mean():6.1%}  max|d| {d.max() if d.size else 0:.2e}  median {q[0]:.1e}  p99 {q[1]:.1e}  p99.99 {q[2]:.1e}  "
          f"byte mismatches {mism} (max |v - rint v| there {mm.max() if mm.size else 0:.1e})  "
          f"flagged at 5e-4/1e-3/2e-3/3.5e-3: " + "/".join(f"{flagged[k]:.2%}" for k in flagged), flush=True)
    worst = max(worst, float(d.max()) if d.size else 0.0)
print(f"worst |v_gpu - v_ref| over {len(classes)} x {w}x{h} pixels: {worst:.3e}")
