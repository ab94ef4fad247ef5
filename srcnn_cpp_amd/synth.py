"""Deterministic synthetic luma planes.

Integer-only, so every host generates identical bytes:

    Y(x,y,f) = min(255, 2*tri(x+5f,61) + tri(y+3f,89) + tri(x+y,23) + (h32(seed,f,y,x) >> 29))
    tri(t,P) = |(t mod 2P) - P|
    h32      = lowbias32 integer hash of  seed ^ ((f*H + y)*W + x)  (mod 2^32):
               h ^= h>>16; h *= 0x7feb352d; h ^= h>>15; h *= 0x846ca68b; h ^= h>>16

Range 0..241: smooth ramps plus 3-bit noise, which keeps the SRCNN output away
from saturation (i.i.d. uniform bytes drive it to 0/255 and are useless for
parity).  Generated directly at conv-plane resolution.
"""
from __future__ import annotations

import numpy as np

DEFAULT_SEED = 12345


def _tri(t, p):
    return np.abs((t % (2 * p)) - p)


def synth_luma(width: int, height: int, frame: int = 0, seed: int = DEFAULT_SEED, rows=None) -> np.ndarray:
    """One height x width uint8 plane for frame index `frame`; rows = (r0, r1): only those rows of it (a rank's stripe)."""
    x = np.arange(width, dtype=np.int64)[None, :]
    y = np.arange(*(rows if rows is not None else (0, height)), dtype=np.int64)[:, None]
    f = int(frame)
    idx = ((f * height + y) * width + x) & 0xFFFFFFFF
    h = (seed ^ idx) & 0xFFFFFFFF
    h ^= h >> 16
    h = (h * 0x7FEB352D) & 0xFFFFFFFF
    h ^= h >> 15
    h = (h * 0x846CA68B) & 0xFFFFFFFF
    h ^= h >> 16
    v = 2 * _tri(x + 5 * f, 61) + _tri(y + 3 * f, 89) + _tri(x + y, 23) + (h >> 29)
    return np.minimum(v, 255).astype(np.uint8)


def synth_batch(width: int, height: int, n_frames: int, first_frame: int = 0,
                seed: int = DEFAULT_SEED) -> np.ndarray:
    """[n_frames, height, width] uint8."""
    return np.stack([synth_luma(width, height, first_frame + k, seed) for k in range(n_frames)])
