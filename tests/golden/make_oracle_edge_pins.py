#!/usr/bin/env python3
"""Secondary pins of the oracle: the G2 edge set of SURVEY.md section 8(c) -- planes narrower and lower than
the 9 x 9 / 5 x 5 windows, sizes divisible by no tile, constant 0 and constant 255 -- as INPUT / OUTPUT vectors of
oracle/srcnn_oracle.c at the time the oracle reproduced the reference's published picture bit for bit
(tests/test_pipeline_oracle.py).  The picture pins the interior arithmetic; these pin the BORDER behaviour for W, H < 9 and the
saturating ends, which a later edit of the oracle could break while still reproducing the picture.
tests/test_oracle_golden.py checks the C oracle AND the independent numpy float32 restatement against them.
Run from the repo root:  python tests/golden/make_oracle_edge_pins.py   (rewrites tests/golden/oracle_edge_pins.npz)"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402
import srcnn_cpp_amd as S  # noqa: E402
from srcnn_cpp_amd.synth import synth_luma  # noqa: E402

SIZES = [(1, 1), (3, 3), (9, 5), (5, 9), (17, 4), (2, 40), (40, 2), (8, 8), (33, 9), (13, 25)]


def planes():
    for k, (w, h) in enumerate(SIZES):
        yield f"synth_{w}x{h}", synth_luma(w, h, frame=k)
    yield "const0_13x7", np.zeros((7, 13), np.uint8)
    yield "const255_13x7", np.full((7, 13), 255, np.uint8)
    yield "const255_4x3", np.full((3, 4), 255, np.uint8)
    rng = np.random.default_rng(8)
    yield "noise_11x6", rng.integers(0, 256, (6, 11), dtype=np.uint8)


if __name__ == "__main__":
    blob = S.load_weights()
    out = {}
    for name, y in planes():
        u8, pre = oracle.forward_y(y, blob)
        out[name + ".in"] = y
        out[name + ".out"] = u8
        out[name + ".pre"] = pre
    np.savez_compressed(ROOT / "tests" / "golden" / "oracle_edge_pins.npz", **out)
    print(f"wrote {len(out) // 3} planes")
