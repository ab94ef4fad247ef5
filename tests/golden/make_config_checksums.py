#!/usr/bin/env python3
"""Per-frame checksums for the full-size GPU tests of BASELINE configs[2] and configs[4], computed in the BUILD
container with the CPU model of the HIP kernels' summation order (oracle/srcnn_gpuorder.c) -- the GPU output of
every frame must reproduce its sha256 bit for bit:

  c2_3840x2160   frames 0..63 of the synthetic 3840x2160 stream (srcnn_cpp_amd/synth.py)   configs[2]
  c4_5760x3240   frames 0..7  of the synthetic 5760x3240 stream (3840x2160 x1.5)            configs[4]
  c3_7680x4320   frame 0 of the synthetic 7680x4320 stream (3840x2160 x2.0), the plane       configs[3]
                 configs[3] row-stripes over 8 GPUs; also the sha256 of each of its 8 stripes
                 of 540 rows (a stripe-level mismatch names the rank)

usage: make_config_checksums.py [key ...]     (default: all; named keys are recomputed, the others kept)

Takes ~15 minutes on 8 cores.  Regenerate after any change of the kernels' summation order
(tests/golden/make_4k_checksums.py holds frame 0 of the first set together with the reference-arithmetic sha)."""
import hashlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

blob = S.load_weights()
OUT = Path(__file__).resolve().parent / "config_checksums.json"
rec = json.loads(OUT.read_text()) if OUT.exists() and len(sys.argv) > 1 else {}
ALL = {"c2_3840x2160": (3840, 2160, 64), "c4_5760x3240": (5760, 3240, 8), "c3_7680x4320": (7680, 4320, 1)}
for key, (w, h, n) in ALL.items():
    if len(sys.argv) > 1 and key not in sys.argv[1:]:
        continue
    shas = []
    for f in range(n):
        y = synth_luma(w, h, frame=f)
        g, _ = oracle.gpuorder_forward_y(y, blob)
        shas.append(hashlib.sha256(g.tobytes()).hexdigest())
        print(key, f, shas[-1][:16], flush=True)
    rec[key] = {"width": w, "height": h, "frames": n, "gpuorder_sha256": shas}
    if key == "c3_7680x4320":
        rec[key]["stripes"] = 8
        rec[key]["stripe_gpuorder_sha256"] = [hashlib.sha256(g[k * h // 8:(k + 1) * h // 8].tobytes()).hexdigest() for k in range(8)]
OUT.write_text(json.dumps(rec, indent=1) + "\n")
