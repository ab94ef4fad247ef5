"""CPU check of the PRINCIPLE behind SRCNN_MODE_REFBYTES (csrc/srcnn_exact.hip), with the two CPU
restatements only -- oracle/srcnn_gpuorder.c is bitwise the GPU's float32 MFMA path, oracle/srcnn_oracle.c the reference
arithmetic: a byte of the MFMA path can differ from the reference's only where its pre-truncation value v lies within delta of
an integer (and 0.5 < v < 255.5), so replacing exactly those pixels by the reference's gives the reference's plane.  The GPU
tests (tests/test_gpu_refbytes.py) check the kernels; this checks the selection rule and the threshold the library derives
from the model, on the CPU suite's budget."""
import numpy as np
import pytest

import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma


def shipped_delta(blob, margin=4.0):
    """fixup_delta() of srcnn_model.cpp: margin (default 4; rounds 3-4: 6) * 2^-24 * ||W3||_2 * (rigorous bound of the layer-2 map for any 8-bit input), plus
    4 * 2^-24 * 256 for the roundings at the output's own magnitude (the b3 additions)."""
    w1, b1, w2, b2, w3, _ = S.split_weights(blob)
    w1, w2, w3 = np.asarray(w1, np.float64).reshape(64, 81), np.asarray(w2, np.float64).reshape(32, 64), np.asarray(w3, np.float64)
    a1 = np.maximum(0.0, 255.0 * np.maximum(w1, 0).sum(1) + np.asarray(b1, np.float64))
    m2 = (np.maximum(w2, 0) @ a1 + np.asarray(b2, np.float64)).max()
    return margin * 2.0 ** -24 * np.sqrt((w3 ** 2).sum()) * m2 + 4.0 * 2.0 ** -24 * 256.0


K_LOCAL, ABS_TERM, EPS = 4.0 * 0.4, 16.0 * 2.0 ** -24 * 256.0, 2.0 ** -24      # srcnn_ctx.h: kFixMargin * kFixLocal, kFixAbsLocal


def local_threshold(y, blob, delta, k=K_LOCAL):
    """The PER-PIXEL threshold of round 6 (srcnn_kernels.h fix_threshold(), l3_row_is_scale()): min(delta, k * 2^-24 * S1 + abs)
    with S1(y, x) = the sum over the pixel's 5 x 5 feature window (replicate border) of U = sum_c max_tap|W3[c][tap]| * F_c on the
    kernels' own layer-2 map.  (The kernels sum S1 in another order; it scales a threshold, its last bits decide nothing.)"""
    w1, b1, w2, b2, w3, _ = oracle.split_weights(blob)
    F = oracle.gpuorder_conv99x11(y, w1, b1, w2, b2)
    U = np.tensordot(np.abs(w3).reshape(32, 25).max(axis=1).astype(np.float64), F.astype(np.float64), axes=(0, 0))
    h, w = U.shape
    S1 = np.zeros_like(U)
    for dy in range(-2, 3):
        ys = np.clip(np.arange(h) + dy, 0, h - 1)
        for dx in range(-2, 3):
            S1 += U[np.ix_(ys, np.clip(np.arange(w) + dx, 0, w - 1))]
    return np.minimum(delta, k * EPS * S1 + ABS_TERM)


def planes():
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:150, 0:260]
    yield "synthetic", synth_luma(260, 150, frame=4)
    yield "white noise", rng.integers(0, 256, (150, 260), dtype=np.uint8)
    yield "bright ramp", np.clip(200 + xx * 55 // 260 + rng.integers(0, 4, (150, 260)), 0, 255).astype(np.uint8)
    yield "checkerboard", np.where(((yy // 8) + (xx // 8)) % 2 == 0, 16, 240).astype(np.uint8)
    yield "constant", np.full((60, 90), 128, np.uint8)
    yield "tiny", synth_luma(5, 3, frame=1)


def test_threshold_of_the_shipped_model(weights_blob):
    assert abs(shipped_delta(weights_blob) - 1.3759e-3) < 2e-6          # what srcnn_fixup_stats reports on the GPU


@pytest.mark.parametrize("name,y", list(planes()), ids=[n for n, _ in planes()])
def test_flagged_pixels_are_all_that_can_differ(weights_blob, name, y):
    delta = shipped_delta(weights_blob)
    g_out, g_pre = oracle.gpuorder_forward_y(y, weights_blob)          # the MFMA path, bit for bit
    r_out, r_pre = oracle.forward_y(y, weights_blob)                   # the reference arithmetic
    flagged = (np.abs(g_pre - np.rint(g_pre)) <= delta) & (g_pre > 0.5) & (g_pre < 255.5)
    fixed = np.where(flagged, r_out, g_out)                            # recompute exactly the flagged pixels
    assert np.array_equal(fixed, r_out), f"{name}: {(fixed != r_out).sum()} bytes differ outside the flagged set"
    # the margin: the noise stays far below the threshold, and the threshold stays selective on textured content
    live = (g_pre > 0.5) & (g_pre < 255.5)
    if live.any():
        assert np.abs(g_pre - r_pre)[live].max() < 0.5 * delta
    if name in ("synthetic", "white noise", "bright ramp"):
        assert flagged.mean() < 0.012


@pytest.mark.parametrize("name,y", list(planes()), ids=[n for n, _ in planes()])
def test_locally_flagged_pixels_are_all_that_can_differ(weights_blob, name, y):
    """Round 6: the same principle against the PER-PIXEL threshold -- fewer pixels flagged, still every byte that can differ among
    them, and no deviation above half its own pixel's threshold (what the device-side net acts on)."""
    delta = shipped_delta(weights_blob)
    thr = local_threshold(y, weights_blob, delta)
    g_out, g_pre = oracle.gpuorder_forward_y(y, weights_blob)
    r_out, r_pre = oracle.forward_y(y, weights_blob)
    live = (g_pre > 0.5) & (g_pre < 255.5)
    flagged = (np.abs(g_pre - np.rint(g_pre)) <= thr) & live
    assert np.array_equal(np.where(flagged, r_out, g_out), r_out), f"{name}: bytes differ outside the flagged set"
    if live.any():
        assert (np.abs(g_pre.astype(np.float64) - r_pre) / thr)[live].max() < 0.5
    if name in ("synthetic", "white noise"):
        glob = (np.abs(g_pre - np.rint(g_pre)) <= delta) & live
        assert thr[live].mean() < 0.66 * delta and flagged.sum() < 0.8 * glob.sum()      # (small planes: the counts are noisy)


def test_model_with_small_weights_and_a_large_bias():
    """The weight-proportional part of the threshold nearly vanishes for such a model, while the roundings at the output's own
    magnitude (b3 is added last) do not: the absolute term of fixup_delta() covers them (tests/checks/soak_models.py met a
    deviation of 0.86 x the threshold before that term existed)."""
    rng = np.random.default_rng(11)
    thin = 0
    for _ in range(3):
        w1 = (rng.standard_normal(5184) * rng.uniform(0.02, 0.05)).astype(np.float32)
        b1 = (rng.standard_normal(64) * 2).astype(np.float32)
        w2 = (rng.standard_normal(2048) * rng.uniform(0.02, 0.06)).astype(np.float32)
        b2 = (rng.standard_normal(32) * 2).astype(np.float32)
        w3 = (rng.standard_normal(800) * rng.uniform(0.005, 0.05)).astype(np.float32)
        blob = np.concatenate([b1, w1, b2, w2, [np.float32(rng.uniform(129, 250))], w3]).astype(np.float32)
        delta = shipped_delta(blob)
        assert delta < 3e-4                                                # the absolute term is most of it
        y = rng.integers(0, 256, (160, 400), dtype=np.uint8)
        g_out, g_pre = oracle.gpuorder_forward_y(y, blob)
        r_out, r_pre = oracle.forward_y(y, blob)
        flagged = (np.abs(g_pre - np.rint(g_pre)) <= delta) & (g_pre > 0.5) & (g_pre < 255.5)
        assert np.array_equal(np.where(flagged, r_out, g_out), r_out)
        assert np.abs(g_pre - r_pre).max() < 0.5 * delta
        thin += int(np.abs(g_pre - r_pre).max() > 0.5 * (delta - 4.0 * 2.0 ** -24 * 256.0))
    assert thin > 0, "the case is meant to leave the weight-proportional term alone without its margin"
