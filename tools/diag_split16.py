#!/usr/bin/env python3
"""Diagnostics: DIAG build of the split-f16 kernel (SRCNN_DEBUG_TUNE=2): where a wave's cycles go."""
import ctypes, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["SRCNN_DEBUG_TUNE"] = os.environ.get("SRCNN_DEBUG_TUNE", "2")
import numpy as np, torch
import srcnn_cpp_amd as S
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
print(f"cycles per row per wave: median {np.median(tot/rows):.0f}  (MFMA only: 42 x 32 = 1344; two waves share a SIMD)")
for name, col in [("Y prefetch + B reads + layer 1 (24 MFMA)", 2), ("ReLU/split + layer 2 (12 MFMA)", 3), ("bias/ReLU/split + layer 3 (6 MFMA)", 4), ("vertical sums, F tile, Y staging", 5)]:
    v = raw[:, col].astype(float) / rows
    print(f"  {name:44s} median {np.median(v):7.0f}  slot0 {np.median(v[slot == 0]):7.0f}  slot1 {np.median(v[slot != 0]):7.0f}")
print(f"  {'barrier':44s} median {np.median(bar/rows):7.0f}")
print("lifetime slot0 / slot1:", np.median(tot[slot == 0]), np.median(tot[slot != 0]))
