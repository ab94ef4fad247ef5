"""GPU parity tests of the steps either side of the conv path (SURVEY.md section
8f ranks 1-2) and of the whole pipeline region src/srcnn.cpp:505-659, through
the C ABI.  These are integer/byte kernels: BIT-EXACT against
oracle/opencv_steps.c.  The whole pipeline is bit-exact in SRCNN_MODE_EXACT and,
in the default MFMA mode, bitwise equal to the oracle chain with the FMA-order
model on the Y plane."""
from pathlib import Path

import numpy as np
import pytest

import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"


def synth_bgr(w, h, seed=0):
    return np.stack([synth_luma(w, h, frame=seed + k, seed=99 + k) for k in range(3)], axis=2)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (64, 64), (257, 33), (300, 201)])
def test_colour_kernels_bit_exact(gpu_ctx, w, h):
    rng = np.random.default_rng(w * 31 + h)
    bgr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    y, cr, cb = gpu_ctx.bgr2ycrcb(bgr)
    oy, ocr, ocb = oracle.bgr2ycrcb(bgr)
    assert np.array_equal(y, oy) and np.array_equal(cr, ocr) and np.array_equal(cb, ocb)
    # back-conversion on arbitrary (also out-of-gamut) triples: exercises the saturation
    planes = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(3)]
    assert np.array_equal(gpu_ctx.ycrcb2bgr(*planes), oracle.ycrcb2bgr(*planes))


def test_colour_kernels_padded_rows(gpu_ctx):
    buf = np.zeros((20, 40, 3), np.uint8)
    buf[:, :33] = synth_bgr(33, 20)
    y, cr, cb = gpu_ctx.bgr2ycrcb(buf[:, :33])
    oy, ocr, ocb = oracle.bgr2ycrcb(np.ascontiguousarray(buf[:, :33]))
    assert np.array_equal(y, oy) and np.array_equal(cr, ocr) and np.array_equal(cb, ocb)


@pytest.mark.parametrize("sw,sh,dw,dh", [(1, 1, 1, 1), (1, 1, 5, 3), (4, 4, 6, 6), (31, 23, 62, 46), (31, 23, 31, 23),
                                         (101, 77, 131, 100), (384, 384, 576, 576), (200, 120, 100, 60),
                                         (97, 61, 291, 183), (640, 360, 1280, 720)])
def test_resize_cubic_bit_exact(gpu_ctx, sw, sh, dw, dh):
    rng = np.random.default_rng(sw + 7 * dw)
    src = rng.integers(0, 256, (sh, sw), dtype=np.uint8)
    assert np.array_equal(gpu_ctx.resize_cubic(src, dw, dh), oracle.resize_cubic(src, dw, dh))
    smooth = synth_luma(sw, sh)
    assert np.array_equal(gpu_ctx.resize_cubic(smooth, dw, dh), oracle.resize_cubic(smooth, dw, dh))


# (output widths that are multiples of four take the two fused launches around the conv path -- colour conversion inside the
# resize kernels -- the others the three separate kernels: both forms are covered)
@pytest.mark.parametrize("w,h,scale", [(40, 30, 2.0), (97, 61, 1.5), (33, 20, 3.0), (64, 48, 1.3), (320, 180, 2.0),
                                       (200, 100, 2.4), (130, 70, 4.0)])
def test_whole_pipeline(gpu_ctx, weights_blob, w, h, scale):
    bgr = synth_bgr(w, h)
    assert S.scaled_size(w, h, scale) == oracle.scaled_size(w, h, scale)
    out = gpu_ctx.process_bgr(bgr, scale)
    model = oracle.process_bgr(bgr, scale, weights_blob, y_path=oracle.gpuorder_forward_y)
    assert np.array_equal(out, model)                 # MFMA mode == FMA-order model, bitwise
    ref = oracle.process_bgr(bgr, scale, weights_blob)
    d = np.abs(out.astype(int) - ref.astype(int))
    assert d.max() <= 3                               # a 1-LSB luma flip moves B/G/R by <= 2 after descale
    assert (d.max(axis=2) != 0).mean() <= 2e-3
    gpu_ctx.set_mode(S.MODE_EXACT)
    try:
        exact = gpu_ctx.process_bgr(bgr, scale)
    finally:
        gpu_ctx.set_mode(S.MODE_MFMA)
    assert np.array_equal(exact, ref)                 # EXACT mode == reference arithmetic, bitwise


def test_butterfly_example_on_gpu(gpu_ctx, weights_blob):
    """configs[0] end to end on the GPU: the reference's README example."""
    z = np.load(GOLD / "butterfly_bgr.npz")
    src, ref = z["src_bgr"], z["ref_bgr"]
    out = gpu_ctx.process_bgr(src, 1.5)
    d = np.abs(out.astype(int) - ref.astype(int))
    assert out.shape == ref.shape and d.max() <= 2
    assert (d.max(axis=2) == 0).mean() >= 0.9995          # MFMA mode: a handful of pixels on a truncation boundary
    assert np.array_equal(out, oracle.process_bgr(src, 1.5, weights_blob, y_path=oracle.gpuorder_forward_y))
    # SRCNN_MODE_EXACT on the GPU reproduces the reference's published picture bit for bit, all 995,328 bytes
    gpu_ctx.set_mode(S.MODE_EXACT)
    try:
        exact = gpu_ctx.process_bgr(src, 1.5)
    finally:
        gpu_ctx.set_mode(S.MODE_MFMA)
    assert np.array_equal(exact, ref)


def test_pipeline_device_entry_point(gpu_ctx, weights_blob):
    import torch
    w, h, scale = 160, 90, 2.0
    bgr = synth_bgr(w, h, seed=5)
    ow, oh = S.scaled_size(w, h, scale)
    d_in = torch.from_numpy(bgr).cuda()
    d_out = torch.zeros((oh, ow, 3), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.process_bgr_dev(d_in.data_ptr(), 3 * w, w, h, scale, d_out.data_ptr(), 3 * ow)
    gpu_ctx.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), gpu_ctx.process_bgr(bgr, scale))


def test_pipeline_error_paths(gpu_ctx):
    with pytest.raises(S.SrcnnError):
        gpu_ctx.process_bgr(np.zeros((3, 3, 3), np.uint8), 0.2)       # (int)(3*0.2) == 0: src/srcnn.cpp:485-495
    with pytest.raises(TypeError):
        gpu_ctx.process_bgr(np.zeros((3, 3), np.uint8), 2.0)
