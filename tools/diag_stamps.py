#!/usr/bin/env python3
"""Diagnostics: run the fused kernel's DIAG build (SRCNN_DEBUG_TUNE=2) and print
where a wave's cycles go and the in-kernel clock (s_memtime / s_memrealtime)."""
import ctypes, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["SRCNN_DEBUG_TUNE"] = os.environ.get("SRCNN_DEBUG_TUNE", "2")
import numpy as np, torch
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_batch

W, H = 3840, 2160
ctx = S.Context(0); ctx.set_weights_blob(S.load_weights())
d_in = torch.from_numpy(synth_batch(W, H, 1)).cuda(); d_out = torch.zeros_like(d_in)
t0 = time.time()
while time.time() - t0 < float(os.environ.get("WARM_S", "2.0")):      # hold the chip under load first
    for _ in range(50):
        ctx.forward_y_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, 1)
    ctx.synchronize()
plan = ctx.query_plan(W, H, 1)
n = int(os.environ.get("DIAG_BLOCKS", plan["workgroups"])) * 4
buf = np.zeros(1 << 17, np.uint64)
lib = S.load_library()
lib.srcnn_debug_read_sink.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
assert lib.srcnn_debug_read_sink(ctx._h, buf.ctypes.data, buf.nbytes) == 0
raw = buf[128:128 + n * 8].reshape(n, 8)
st = raw.astype(np.float64)
tot, real, top, l1, l23, bar, _, _ = st.T
rows = (raw[:, 6] & np.uint64(0xffff)).astype(np.float64)
start = (raw[:, 6] >> np.uint64(32)).astype(np.int64)
hwid = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.float64)
xcc = ((raw[:, 7] >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
clk = tot / real * 100e6
print(f"waves {n}  rows/wave {rows.mean():.1f}")
print(f"in-kernel clock GHz: median {np.median(clk)/1e9:.3f}  min {clk.min()/1e9:.3f} max {clk.max()/1e9:.3f}")
print(f"kernel cycles per wave: median {np.median(tot):.0f}  max {tot.max():.0f}  -> per row {np.median(tot/rows):.0f}")
for name, v in [("top(loop head->L1)", top), ("L1 (82 MFMA + horizontal sum of previous row)", l1), ("L2+L3+vertical chains+F write", l23), ("barrier wait", bar)]:
    print(f"  {name:48s} per row: median {np.median(v/rows):8.0f}   mean {np.mean(v/rows):8.0f}")
print("ideal MFMA cycles per row: L1 5248, L2+L3 3072, total 8320 (x2 waves per SIMD when 2 WG/CU)")
slot = (hwid.astype(np.int64) & 0xF)
for sl in np.unique(slot):
    m = slot == sl
    print(f"  wave slot {sl}: {m.sum()} waves, total cycles median {np.median(tot[m]):.0f}, L1/row {np.median((l1/rows)[m]):.0f}, bar/row {np.median((bar/rows)[m]):.0f}")
# dispatch-order check: do the first n_cu blocks (hardware order = blockIdx) take wave slot 0 of distinct CUs?
blk = np.arange(n) // 4
first = blk < 256
print(f"blocks <256 in slot 0: {(slot[first] == 0).mean():.3f};  blocks >=256 in slot 1: {(slot[~first] == 1).mean():.3f}")
cu = (hwid.astype(np.int64) >> 8) & 0xF
se = (hwid.astype(np.int64) >> 13) & 0x7
print("lifetime (cycles) by block range: <256:", np.median(tot[first]), " >=256:", np.median(tot[~first]))

# which blocks share a CU?  CU identity = (XCC_ID, SE, SH, CU) of wave 0 of each block
h = hwid.astype(np.int64)[::4]
cuid = (xcc[::4] << 12) | (((h >> 13) & 7) << 5) | (((h >> 12) & 1) << 4) | ((h >> 8) & 0xF)
nb = n // 4
print("distinct CUs:", len(np.unique(cuid)), " blocks:", nb, " xcc of blocks 0..15:", xcc[::4][:16].tolist())
by = {}
for b in range(nb):
    by.setdefault(int(cuid[b]), []).append(b)
pairs = sorted(v for v in by.values())
cnt = {}
for v in pairs:
    cnt[len(v)] = cnt.get(len(v), 0) + 1
print("blocks per CU histogram:", cnt)
d = [v[1] - v[0] for v in pairs if len(v) == 2]
if d:
    vals, c = np.unique(d, return_counts=True)
    print("second block id - first block id on the same CU (value: count):", dict(zip(vals.tolist(), c.tolist())))
print("first 24 CU pairs:", pairs[:24])
st0 = start[::4] - start[::4].min()
print("start time (100 MHz ticks) of blocks 0,64,128,192,255,256,300,400,511:", [int(st0[min(b, nb - 1)]) for b in (0, 64, 128, 192, 255, 256, 300, 400, 511)])
tb = tot[::4]; rb = rows[::4]
life_cu = {k: max(int(st0[b] * 23.8 + tb[b]) for b in v) for k, v in by.items()}
print("per-CU finish (cycles from first start): min %d median %d max %d" % (min(life_cu.values()), np.median(list(life_cu.values())), max(life_cu.values())))
