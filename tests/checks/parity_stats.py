#!/usr/bin/env python3
"""Print the measured parity of the MFMA path against the reference arithmetic
(oracle) on a full 3840x2160 synthetic frame -- the numbers DESIGN.md section 6 quotes."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np
import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

w, h = 3840, 2160
blob = S.load_weights()
y = synth_luma(w, h)
ctx = S.Context(0)
ctx.set_weights_blob(blob)
pre = np.empty((h, w), np.float32)
out = ctx.forward_y(y, preclamp=pre)
r_out, r_pre = oracle.forward_y(y, blob)
d = np.abs(out.astype(int) - r_out.astype(int))
print(f"frame {w}x{h}: pre-clamp max|d| = {np.abs(pre - r_pre).max():.3e}, mean|d| = {np.abs(pre - r_pre).mean():.3e}")
print(f"u8: max|d| = {d.max()}, mismatching pixels = {(d != 0).sum()} of {d.size} = {(d != 0).mean():.3e}")
near = np.abs(r_pre - np.rint(r_pre))[d != 0]
print(f"distance of the reference pre-truncation value to an integer at the mismatches: max {near.max():.3e}")
w1, b1, w2, b2, w3, b3 = S.split_weights(blob)
ys = y[:400, :600].copy()
buf = np.zeros((32, 400, 600), np.float32)
ctx.conv99x11(ys, [buf[k] for k in range(32)], w1, b1, w2, b2)
ref = oracle.conv99x11(ys, w1, b1, w2, b2)
rel = np.abs(buf - ref) / np.maximum(1.0, np.abs(ref))
print(f"32-channel map (600x400 crop): max |d|/max(1,|ref|) = {rel.max():.3e}, max|ref| = {np.abs(ref).max():.1f}")
