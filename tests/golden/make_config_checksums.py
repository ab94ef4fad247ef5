#!/usr/bin/env python3
"""Per-frame checksums for the full-size GPU tests of BASELINE configs[2] and configs[4], computed in the BUILD
container with the CPU model of the HIP kernels' summation order (oracle/srcnn_gpuorder.c) -- the GPU output of
every frame must reproduce its sha256 bit for bit:

  c2_3840x2160   frames 0..63 of the synthetic 3840x2160 stream (srcnn_cpp_amd/synth.py)   configs[2]
  c4_5760x3240   frames 0..7  of the synthetic 5760x3240 stream (3840x2160 x1.5)            configs[4]

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
rec = {}
for key, (w, h, n) in {"c2_3840x2160": (3840, 2160, 64), "c4_5760x3240": (5760, 3240, 8)}.items():
    shas = []
    for f in range(n):
        y = synth_luma(w, h, frame=f)
        g, _ = oracle.gpuorder_forward_y(y, blob)
        shas.append(hashlib.sha256(g.tobytes()).hexdigest())
        print(key, f, shas[-1][:16], flush=True)
    rec[key] = {"width": w, "height": h, "frames": n, "gpuorder_sha256": shas}
(Path(__file__).resolve().parent / "config_checksums.json").write_text(json.dumps(rec, indent=1) + "\n")
