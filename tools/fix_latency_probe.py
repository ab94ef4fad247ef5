#!/usr/bin/env python3
"""What fix_apply_kernel costs as a function of how many items it has (round 6): SRCNN_MODE_REFBYTES steps on planes whose
flagged-pixel count is controlled -- a constant plane (none), the synthetic plane cropped to WxH -- for rocprofv3 --kernel-trace.
usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/fix_latency_probe.py KIND W H [steps]     KIND = const | synth"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np, torch
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

kind, w, h = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
y = np.full((h, w), 131, np.uint8) if kind == "const" else synth_luma(w, h)
with S.Context(0) as ctx:
    ctx.set_weights_blob(S.load_weights())
    ctx.set_mode(S.MODE_REFBYTES)
    d_in = torch.from_numpy(y).cuda()
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    for _ in range(steps):
        ctx.forward_y_dev(d_in.data_ptr(), w, 0, d_out.data_ptr(), w, 0, w, h, 1)
    ctx.synchronize()
    st = ctx.fixup_stats()
    print(kind, w, h, "flagged per step", st["scattered_pixels"] / steps, "dense tiles per step", st["dense_tiles"] / steps)
