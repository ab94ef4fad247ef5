#!/usr/bin/env python3
"""Accuracy of SRCNN_MODE_SPLIT16 and SRCNN_MODE_MFMA against the oracle on the 4K bench frame."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np, torch
import oracle, srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma
W, H = 3840, 2160
blob = S.load_weights()
y = synth_luma(W, H)
r_out, r_pre = oracle.forward_y(y, blob)
ctx = S.Context(0); ctx.set_weights_blob(blob)
for name, mode in [("mfma (f32)", S.MODE_MFMA), ("split16", S.MODE_SPLIT16)]:
    ctx.set_mode(mode)
    pre = np.empty((H, W), np.float32)
    out = ctx.forward_y(y, preclamp=pre)
    e = np.abs(pre - r_pre)
    print(f"{name:12s} pre-clamp |d| max {e.max():.3e} mean {e.mean():.3e} p99.9 {np.quantile(e, 0.999):.3e};  u8 mismatches {(out != r_out).sum()} of {out.size} (max {np.abs(out.astype(int) - r_out.astype(int)).max()} LSB)")
