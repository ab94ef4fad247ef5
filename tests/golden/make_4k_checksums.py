#!/usr/bin/env python3
"""Checksums of the conv path on the synthetic 3840x2160 frame (BASELINE configs[1] input,
srcnn_cpp_amd/synth.py, frame 0), computed in the BUILD container with the CPU oracle:

  oracle     sha256 of the u8 plane produced by the reference arithmetic (oracle/srcnn_oracle.c)
  gpuorder   sha256 of the u8 plane produced by the FMA-order model of the HIP kernels
             (oracle/srcnn_gpuorder.c) -- the GPU output must reproduce this one bit for bit

Regenerate after any change of the kernels' summation order."""
import hashlib, json, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

blob = S.load_weights()
y = synth_luma(3840, 2160)
o, _ = oracle.forward_y(y, blob)
g, _ = oracle.gpuorder_forward_y(y, blob)
rec = {"input_sha256": hashlib.sha256(y.tobytes()).hexdigest(),
       "oracle_sha256": hashlib.sha256(o.tobytes()).hexdigest(), "oracle_sum": int(o.astype(np.int64).sum()),
       "gpuorder_sha256": hashlib.sha256(g.tobytes()).hexdigest(), "gpuorder_sum": int(g.astype(np.int64).sum()),
       "u8_mismatches_between_them": int((o != g).sum())}
(Path(__file__).resolve().parent / "synthetic_4k_checksums.json").write_text(json.dumps(rec, indent=1) + "\n")
print(rec)
