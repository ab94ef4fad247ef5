#!/usr/bin/env python3
"""Where the fixed cost of a single-plane launch goes: four wall-clock stamps per wave (s_memrealtime, 100 MHz) of the
production fused launch (SRCNN_DEBUG_TUNE=16: entry, loop start, loop end, exit) -- no per-row stamps, and the stamped
instantiation runs the same FAST row body as the shipped kernel (since round 3), so the timing is the shipped kernel's.  usage: tools/diag_light.py [W H]"""
import ctypes, os, sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
os.environ["SRCNN_DEBUG_TUNE"] = "16"
import numpy as np, torch
import srcnn_cpp_amd as S
S.use_library(S.tuning_library_path())      # the stamped kernels and srcnn_debug_read_sink live in the tuning build
from srcnn_cpp_amd.synth import synth_batch

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
ctx = S.Context(0); ctx.set_weights_blob(S.load_weights())
d_in = torch.from_numpy(synth_batch(W, H, 1)).cuda(); d_out = torch.zeros_like(d_in)
t0 = time.time()
while time.time() - t0 < 2.0:                       # hold the chip under load first
    for _ in range(50):
        ctx.forward_y_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, 1)
    ctx.synchronize()
plan = ctx.query_plan(W, H, 1)
nb = plan["workgroups"]
n = nb * 4
buf = np.zeros(1 << 17, np.uint64)
lib = S.load_library()
lib.srcnn_debug_read_sink.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
assert lib.srcnn_debug_read_sink(ctx._h, buf.ctypes.data, buf.nbytes) == 0
raw = buf[128:128 + n * 8].reshape(n, 8)
t = raw[:, :4].astype(np.int64)
rows = raw[:, 4].astype(np.int64)
hw = (raw[:, 5] & np.uint64(0xffffffff)).astype(np.int64)
xcc = ((raw[:, 5] >> np.uint64(32)) & np.uint64(0xf)).astype(np.int64)
us = lambda ticks: np.asarray(ticks, dtype=np.float64) / 100.0
k0 = t[:, 0].min()
entry, loop0, loop1, exit_ = (us(t[:, i] - k0) for i in range(4))
blk = np.arange(n) // 4
first = blk < nb // 2
print(f"{W}x{H}: {nb} work items, rows per item mean {rows.mean():.1f} (min {rows.min()}, max {rows.max()})")
print(f"kernel span (first entry -> last exit): {exit_.max():.1f} us")
for name, m in (("first half of the blocks", first), ("second half", ~first)):
    print(f"  {name}: entry median {np.median(entry[m]):.2f} us (max {entry[m].max():.2f});  "
          f"prologue (entry -> loop) median {np.median((loop0 - entry)[m]):.2f} us (max {(loop0 - entry)[m].max():.2f});  "
          f"loop median {np.median((loop1 - loop0)[m]):.1f} us = {np.median(((loop1 - loop0) / rows)[m]):.3f} us/row;  "
          f"epilogue median {np.median((exit_ - loop1)[m]):.2f} us;  exit median {np.median(exit_[m]):.1f} (min {exit_[m].min():.1f}, max {exit_[m].max():.1f})")
cuid = (xcc[::4] << 12) | (((hw[::4] >> 13) & 7) << 5) | (((hw[::4] >> 12) & 1) << 4) | ((hw[::4] >> 8) & 0xF)
fin_blk = exit_.reshape(nb, 4).max(axis=1)
by = {}
for b in range(nb):
    by.setdefault(int(cuid[b]), []).append(b)
fin_cu = np.array([max(fin_blk[b] for b in v) for v in by.values()])
one_left = np.array([abs(fin_blk[v[0]] - fin_blk[v[1]]) for v in by.values() if len(v) == 2])
print(f"CUs used: {len(by)};  per-CU finish: min {fin_cu.min():.1f} median {np.median(fin_cu):.1f} max {fin_cu.max():.1f} us;  "
      f"mean idle tail per CU {np.mean(fin_cu.max() - fin_cu):.1f} us;  time a CU runs ONE workgroup at the end: median {np.median(one_left):.1f} us")
rows_cu = np.array([sum(rows[4 * b] for b in v) for v in by.values()])
strip_ys = raw[::4, 6]
order = np.argsort(-fin_cu)
keys = list(by.keys())
print("slowest / fastest CUs: finish us | XCC | blocks: (block id, strip, first row, rows, loop us/row, exit us)")
for j in list(order[:12]) + list(order[-6:]):
    v = by[keys[j]]
    desc = "  ".join(f"({b}, s{int(strip_ys[b]) & 0xffffffff}, y{int(strip_ys[b]) >> 32}, {rows[4 * b]}r, {((loop1 - loop0)[4 * b] / max(1, rows[4 * b])):.2f}, {fin_blk[b]:.0f})" for b in v)
    print(f"  {fin_cu[j]:7.1f} | xcc {keys[j] >> 12} | {desc}")
if os.environ.get("DIAG_DUMP"):      # per-CU table for fitting the planner's model: rows and times of the two workgroups of every CU
    with open(os.environ["DIAG_DUMP"], "a") as fh:
        for k, v in by.items():
            if len(v) != 2:
                continue
            f, s2 = (v[0], v[1]) if v[0] < v[1] else (v[1], v[0])       # first-dispatched block, the one that joined it
            fh.write(f"{W} {H} {k >> 12} {rows[4 * f]} {rows[4 * s2]} {loop0.reshape(nb, 4)[f].max():.2f} {fin_blk[f]:.2f} "
                     f"{loop0.reshape(nb, 4)[s2].max():.2f} {fin_blk[s2]:.2f} {f} {s2}\n")
xc = np.array([k >> 12 for k in keys])
print("per-XCC median CU finish:", {int(x): round(float(np.median(fin_cu[xc == x])), 1) for x in np.unique(xc)})
print(f"rows per CU: min {rows_cu.min()} max {rows_cu.max()};  loop us/row by CU pair total: median {np.median(fin_cu / rows_cu * 2):.3f}")
print(f"ideal MFMA time per row (130 MFMA x 64 cycles x 2 waves / 2.4 GHz): {130 * 64 * 2 / 2400:.3f} us")
