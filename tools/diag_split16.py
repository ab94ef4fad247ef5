#!/usr/bin/env python3
"""Diagnostics: DIAG build of the split-f16 kernel (SRCNN_DEBUG_TUNE=2): where a wave's cycles go."""
import ctypes, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["SRCNN_DEBUG_TUNE"] = os.environ.get("SRCNN_DEBUG_TUNE", "2")
import numpy as np, torch
import srcnn_cpp_amd as S
S.use_library(S.tuning_library_path())      # the stamped kernels and srcnn_debug_read_sink live in the tuning build
from srcnn_cpp_amd.synth import synth_batch
W, H = 3840, 2160
ctx = S.Context(0); ctx.set_weights_blob(S.load_weights()); ctx.set_mode(S.MODE_SPLIT16)
d_in = torch.from_numpy(synth_batch(W, H, 1)).cuda(); d_out = torch.zeros_like(d_in)
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(100):
        ctx.forward_y_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, 1)
    ctx.synchronize()
n = ctx.query_plan(W, H, 1)["workgroups"] * 4
buf = np.zeros(1 << 17, np.uint64)
lib = S.load_library()
lib.srcnn_debug_read_sink.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
assert lib.srcnn_debug_read_sink(ctx._h, buf.ctypes.data, buf.nbytes) == 0
raw = buf[128:128 + n * 8].reshape(n, 8)
tot, real = raw[:, 0].astype(float), raw[:, 1].astype(float)
rows = (raw[:, 6] & np.uint64(0xffff)).astype(float)
bar = (raw[:, 7] >> np.uint64(32)).astype(float)
slot = (raw[:, 7] & np.uint64(0xf)).astype(int)
print(f"waves {n} rows/wave {rows.mean():.1f}; in-kernel clock GHz median {np.median(tot/real*100e6)/1e9:.3f}")
print(f"cycles per row per wave: median {np.median(tot/rows):.0f}  (MFMA only: 42 x 32 = 1344; one wave per SIMD)")
for name, col, n_mfma in [("MFMA 1-16  (split of the layer-1 result, vertical sums of the previous row)", 2, 16),
                          ("MFMA 17-23 (end of layer 2)", 3, 7), ("MFMA 24-38 (rescale + split of the layer-2 result, B reads)", 4, 15),
                          ("MFMA 39-42 (Y staging, output row)", 5, 4)]:
    v = raw[:, col].astype(float) / rows
    print(f"  {name:78s} median {np.median(v):6.0f} = {np.median(v) / n_mfma:5.1f} per MFMA")
print(f"  {'barrier + loop head':78s} median {np.median(bar/rows):6.0f}")
print("(each section boundary costs one s_memtime + s_waitcnt in this DIAG build)")
