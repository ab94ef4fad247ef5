"""GPU parity tests of the opt-in split-f16 mode (SRCNN_MODE_SPLIT16).

The mode runs the fused forward pass (src/srcnn.cpp:254-325 + :189-243) on f16 MFMAs with every
float32 operand split into an f16 (hi, lo) pair (srcnn_cpp_amd/csrc/srcnn_split16.hip).  It has no
bitwise CPU model, so everything here is checked against the ORACLE (reference arithmetic) with
the tolerance of the float32 MFMA mode, stated once in tests/test_gpu_parity.py:

    pre-clamp f32 output   max |d| <= 5e-3
    u8 output              <= 1 LSB, only where the reference value is within 5e-3 of an integer,
                           and on <= 1e-3 of the pixels of large planes

plus structural properties that are exact: launch-geometry independence (plane alone == plane in a
batch == row stripes), determinism, and 13x13 locality.
"""
import numpy as np
import pytest

import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma, synth_batch

from test_gpu_parity import EDGE_SIZES, TILE_SIZES, TOL_PRE_ABS, check_u8

pytestmark = pytest.mark.gpu


@pytest.fixture()
def split_ctx(gpu_ctx):
    gpu_ctx.set_mode(S.MODE_SPLIT16)
    try:
        yield gpu_ctx
    finally:
        gpu_ctx.set_mode(S.MODE_MFMA)


@pytest.mark.parametrize("w,h", EDGE_SIZES + TILE_SIZES)
def test_forward_split16(split_ctx, weights_blob, w, h):
    y = synth_luma(w, h, frame=2)
    pre = np.empty((h, w), np.float32)
    out = split_ctx.forward_y(y, preclamp=pre)
    r_out, r_pre = oracle.forward_y(y, weights_blob)
    assert np.isfinite(pre).all()
    assert np.abs(pre - r_pre).max() <= TOL_PRE_ABS
    check_u8(out, r_out, r_pre)


@pytest.mark.parametrize("value", [0, 1, 128, 255])
def test_split16_constant_images(split_ctx, weights_blob, value):
    y = np.full((40, 150), value, np.uint8)
    pre = np.empty(y.shape, np.float32)
    out = split_ctx.forward_y(y, preclamp=pre)
    r_out, r_pre = oracle.forward_y(y, weights_blob)
    assert np.abs(pre - r_pre).max() <= TOL_PRE_ABS
    check_u8(out, r_out, r_pre)


def test_split16_saturating_input(split_ctx, weights_blob):
    """i.i.d. uniform bytes drive the output far outside 0..255: clamping and the f16 ranges hold."""
    rng = np.random.default_rng(5)
    y = rng.integers(0, 256, size=(64, 200), dtype=np.uint8)
    pre = np.empty(y.shape, np.float32)
    out = split_ctx.forward_y(y, preclamp=pre)
    r_out, r_pre = oracle.forward_y(y, weights_blob)
    assert np.isfinite(pre).all()
    assert np.abs(pre - r_pre).max() <= TOL_PRE_ABS * 4      # |values| reach several hundred here
    assert np.abs(out.astype(int) - r_out.astype(int)).max() <= 1


def test_split16_butterfly_fixture(split_ctx, weights_blob):
    """configs[0]: the reference's own example (Y plane of butterfly.png x1.5, 576x576)."""
    from pathlib import Path
    g = Path(__file__).resolve().parent / "golden"
    y = np.fromfile(g / "butterfly_y_in_576.u8", np.uint8).reshape(576, 576)
    y_ref = np.fromfile(g / "butterfly_y_ref_576.u8", np.uint8).reshape(576, 576)   # Y of the reference's own PNG
    out = split_ctx.forward_y(y)
    r_out, r_pre = oracle.forward_y(y, weights_blob)
    check_u8(out, r_out, r_pre)
    d = out.astype(np.float64) - y_ref
    assert 10 * np.log10(255.0 ** 2 / np.mean(d * d)) >= 50.0


def test_split16_is_at_least_as_close_to_the_reference_as_mfma(gpu_ctx, weights_blob):
    """Accuracy claim of the mode: the split keeps 22 bits per operand, so the error against the
    reference arithmetic stays at the float32 level (the two modes are within a factor 2)."""
    w, h = 640, 360
    y = synth_luma(w, h, frame=4)
    _, r_pre = oracle.forward_y(y, weights_blob)
    pre_m = np.empty((h, w), np.float32)
    gpu_ctx.forward_y(y, preclamp=pre_m)
    gpu_ctx.set_mode(S.MODE_SPLIT16)
    try:
        pre_s = np.empty((h, w), np.float32)
        gpu_ctx.forward_y(y, preclamp=pre_s)
    finally:
        gpu_ctx.set_mode(S.MODE_MFMA)
    e_m, e_s = np.abs(pre_m - r_pre), np.abs(pre_s - r_pre)
    assert e_s.max() <= TOL_PRE_ABS
    assert e_s.max() <= 2 * e_m.max() + 1e-4
    assert e_s.mean() <= 2 * e_m.mean() + 1e-5


@pytest.mark.parametrize("w,h", [(3840, 2160), (1920, 1080), (992, 1700), (125, 6400)])
def test_split16_launch_geometry_independence(split_ctx, w, h):
    """Plane alone (per-block work items) == plane inside a batch (regular grid) == row stripe,
    bit for bit: the arithmetic does not depend on the decomposition."""
    import torch
    y = synth_luma(w, h, frame=3)
    d_in = torch.from_numpy(np.stack([y, y])).cuda()
    d_batch = torch.zeros_like(d_in)
    d_one = torch.full((h, w), 7, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    split_ctx.forward_y_dev(d_in.data_ptr(), w, h * w, d_batch.data_ptr(), w, h * w, w, h, 2)
    split_ctx.forward_y_dev(d_in.data_ptr(), w, h * w, d_one.data_ptr(), w, h * w, w, h, 1)
    split_ctx.synchronize()
    batch, one = d_batch.cpu().numpy(), d_one.cpu().numpy()
    assert np.array_equal(batch[0], batch[1])
    assert np.array_equal(one, batch[0])
    r0, r1 = h // 3 + 1, h - h // 5
    d_stripe = torch.full((r1 - r0, w), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    split_ctx.forward_y_rows_dev(d_in.data_ptr(), w, 0, d_stripe.data_ptr(), w, r0, w, h, r0, r1)
    split_ctx.synchronize()
    assert np.array_equal(d_stripe.cpu().numpy(), one[r0:r1])


def test_split16_full_size_4k_frame(split_ctx, weights_blob):
    """configs[1] at full size against the reference arithmetic (oracle on all host cores), and
    against the committed checksum of the oracle's output for this synthetic frame."""
    import hashlib, json
    from pathlib import Path
    w, h = 3840, 2160
    y = synth_luma(w, h)
    pre = np.empty((h, w), np.float32)
    out = split_ctx.forward_y(y, preclamp=pre)
    pins = json.loads((Path(__file__).resolve().parent / "golden" / "synthetic_4k_checksums.json").read_text())
    assert hashlib.sha256(y.tobytes()).hexdigest() == pins["input_sha256"]
    r_out, r_pre = oracle.forward_y(y, weights_blob)
    assert hashlib.sha256(r_out.tobytes()).hexdigest() == pins["oracle_sha256"]
    assert np.abs(pre - r_pre).max() <= TOL_PRE_ABS
    check_u8(out, r_out, r_pre)
    # 13x13 locality: crops reproduce the frame bit for bit
    for (r0, c0) in [(0, 0), (1000, 2000), (h - 80, w - 200)]:
        r1, c1 = min(h, r0 + 80), min(w, c0 + 200)
        c_out = split_ctx.forward_y(np.ascontiguousarray(y[r0:r1, c0:c1]))
        ir0, ic0 = (0 if r0 == 0 else 6), (0 if c0 == 0 else 6)
        ir1, ic1 = (r1 - r0 if r1 == h else r1 - r0 - 6), (c1 - c0 if c1 == w else c1 - c0 - 6)
        assert np.array_equal(c_out[ir0:ir1, ic0:ic1], out[r0 + ir0:r0 + ir1, c0 + ic0:c0 + ic1])


def test_split16_deterministic_and_batched(split_ctx, weights_blob):
    frames = synth_batch(300, 77, 6, first_frame=20)
    a = split_ctx.forward_y_frames(frames)
    b = split_ctx.forward_y_frames(frames)
    assert np.array_equal(a, b)
    for k in range(6):
        assert np.array_equal(a[k], split_ctx.forward_y(frames[k]))
        r_out, r_pre = oracle.forward_y(frames[k], weights_blob)
        check_u8(a[k], r_out, r_pre)


def test_split16_refuses_weights_outside_its_range(gpu_ctx, weights_blob):
    """The mode's f16 ranges hold for the shipped model (rigorous bounds 2,065 and 8,786 on the
    layer maps); weights that break them are refused loudly instead of overflowing."""
    y = synth_luma(64, 32)
    big = weights_blob.copy()
    big[64:5248] *= 64.0                       # layer-1 weights x64: the map bound leaves the range
    gpu_ctx.set_weights_blob(big)
    gpu_ctx.set_mode(S.MODE_SPLIT16)
    try:
        with pytest.raises(S.SrcnnError) as e:
            gpu_ctx.forward_y(y)
        assert e.value.code == -5
        gpu_ctx.set_mode(S.MODE_MFMA)
        gpu_ctx.forward_y(y)                   # the float32 mode takes any weights
    finally:
        gpu_ctx.set_mode(S.MODE_MFMA)
        gpu_ctx.set_weights_blob(weights_blob)
