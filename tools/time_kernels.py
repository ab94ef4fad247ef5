#!/usr/bin/env python3
"""Time the layer-1/2 kernel (srcnn_conv99x11_dev) and the layer-3 kernel (srcnn_conv55_dev) alone."""
import os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_batch
W, H = 3840, 2160
ctx = S.Context(0); ctx.set_weights_blob(S.load_weights())
st = torch.cuda.Stream(); torch.cuda.set_stream(st); ctx.set_stream(st.cuda_stream)
d_in = torch.from_numpy(synth_batch(W, H, 1)).cuda()
d_pl = torch.empty((32, H, W), dtype=torch.float32, device="cuda")
d_out = torch.empty((H, W), dtype=torch.uint8, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(st)
    for _ in range(n): fn()
    b.record(st); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("L12 ms:", round(t(lambda: ctx.conv99x11_dev(d_in.data_ptr(), W, d_pl.data_ptr(), W, W * H, W, H)), 4),
      " L3 ms:", round(t(lambda: ctx.conv55_dev(d_pl.data_ptr(), W, W * H, d_out.data_ptr(), W, W, H)), 4),
      " fused ms:", round(t(lambda: ctx.forward_y_dev(d_in.data_ptr(), W, W * H, d_out.data_ptr(), W, W * H, W, H, 1)), 4))
