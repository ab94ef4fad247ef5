"""CPU check of the NUMERICAL CLAIM behind SRCNN_MODE_SPLIT16 (srcnn_cpp_amd/csrc/srcnn_split16.hip).

The mode replaces every float32 operand by an f16 (hi, lo) pair and drops the lo*lo product.  This
test restates that arithmetic in numpy -- same power-of-two scales, round-toward-zero hi part for
the activations, round-to-nearest for the weights, optionally with every f16 denormal flushed to
zero (the worst case for an MFMA implementation) -- and checks it against the ORACLE (reference
arithmetic, src/srcnn.cpp:254-325 + :189-243) with the tolerance the GPU tests use.  It does not
touch the GPU path; tests/test_gpu_split16.py does.
"""
import numpy as np
import pytest

import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

TOL_PRE_ABS = 5e-3


def _split_rn(a, scale, ftz):
    a = (np.asarray(a, np.float32) * np.float32(scale)).astype(np.float32)
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float32)).astype(np.float16)
    return _flush(hi, ftz), _flush(lo, ftz)


def _flush(h, ftz):
    if ftz:
        h = np.where(np.abs(h) < np.float16(6.1035e-5), np.float16(0), h)
    return h.astype(np.float64)


def _relu_split_rtz(x, ftz):
    """relu_split_pair(): hi = max(rtz_f16(x), 0), lo = clamp(f16(x - hi), 0, 1)."""
    x = np.asarray(x, np.float32)
    hi = x.astype(np.float16)
    over = np.abs(hi.astype(np.float32)) > np.abs(x)
    hi = np.where(over, np.nextafter(hi, np.float16(0)), hi)
    hi = np.maximum(hi, np.float16(0))
    lo = np.clip((x - hi.astype(np.float32)).astype(np.float16), np.float16(0), np.float16(1))
    return _flush(hi, ftz), _flush(lo, ftz)


def split16_model(y, blob, ftz):
    w1, b1, w2, b2, w3, b3 = S.split_weights(blob)
    h, w = y.shape
    yp = np.pad(y.astype(np.float32), 4, mode="edge")
    cols = np.stack([yp[i:i + h, j:j + w] for i in range(9) for j in range(9)], 0).reshape(81, -1)
    xs = (cols * np.float32(2.0 ** -14)).astype(np.float16).astype(np.float64)        # exact
    w1h, w1l = _split_rn(w1.reshape(64, 81), 2.0 ** 11, ftz)
    b1h, b1l = _split_rn(b1, 0.125, ftz)
    a1 = (w1h @ xs + w1l @ xs + (b1h + b1l)[:, None]).astype(np.float32)               # layer-1 map / 8
    assert a1.max() < 1024
    ah, al = _relu_split_rtz(a1, ftz)
    w2h, w2l = _split_rn(w2, 2.0 ** 14, ftz)
    d2 = (w2h @ ah + w2l @ ah + w2h @ al).astype(np.float32)
    a2 = (d2.astype(np.float64) * 2.0 ** -15 + (b2 * np.float32(0.0625))[:, None]).astype(np.float32)   # layer-2 map / 16
    assert a2.max() < 1024
    dh, dl = _relu_split_rtz(a2, ftz)
    w3h, w3l = _split_rn(w3.reshape(32, 25).T.copy(), 2.0 ** 14, ftz)
    t = (w3h @ dh + w3l @ dh + w3h @ dl).astype(np.float32).reshape(25, h, w)
    tp = np.pad(t, ((0, 0), (2, 2), (2, 2)), mode="edge")
    acc = np.zeros((h, w), np.float32)
    for n in range(5):
        fn = np.zeros((h, w), np.float32)
        for m in range(5):
            fn = fn + tp[5 * m + n, m:m + h, n:n + w]
        acc = acc + fn
    return (acc.astype(np.float64) * 2.0 ** -10 + b3).astype(np.float32)


@pytest.mark.parametrize("ftz", [False, True])
def test_split16_arithmetic_matches_reference_arithmetic(ftz):
    blob = S.load_weights()
    y = synth_luma(160, 96, frame=1)
    r_out, r_pre = oracle.forward_y(y, blob)
    pre = split16_model(y, blob, ftz)
    err = np.abs(pre - r_pre)
    assert err.max() <= TOL_PRE_ABS
    assert err.max() <= 1e-3 and err.mean() <= 2e-4        # in fact at the float32 rounding level
    out = np.clip(np.trunc(pre), 0, 255).astype(np.uint8)
    d = np.abs(out.astype(int) - r_out.astype(int))
    assert d.max() <= 1
    if d.any():
        assert np.abs(r_pre - np.rint(r_pre))[d != 0].max() <= TOL_PRE_ABS


def test_split16_ranges_hold_for_any_8bit_input():
    """Rigorous bounds behind the f16 ranges (split16_range_ok() in csrc/srcnn_model.cpp)."""
    w1, b1, w2, b2, w3, _ = S.split_weights(S.load_weights())
    a1 = np.maximum(255.0 * np.maximum(w1.reshape(64, 81), 0).sum(1) + b1, 0)
    a2 = (np.maximum(w2, 0) * a1[None, :]).sum(1) + b2
    assert a1.max() < 8 * 1024 and a2.max() < 16 * 1024
    assert np.abs(w1).max() * 2 ** 11 < 65504 and max(np.abs(w2).max(), np.abs(w3).max()) * 2 ** 14 < 65504
