"""BASELINE.json configs[2] and configs[4] at FULL size on one MI355X (parity, not timing).

configs[2]  batch of 64 x 3840x2160 through the fused kernel AND through the materialising im2col+MFMA path
            (layer-1/2 kernel -> 32 f32 planes per frame in HBM, 68 GB -> layer-3 kernel);
configs[4]  5760x3240 frames (3840x2160 x1.5) as a host-frame stream and as a device batch.

Every output frame must reproduce, bit for bit, the sha256 the CPU model of the kernels' arithmetic gave in the
build container (tests/golden/config_checksums.json, made by tests/golden/make_config_checksums.py); three frames
per config are also compared with the reference arithmetic (oracle) within the stated tolerance."""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_batch

pytestmark = pytest.mark.gpu


def pins(key):
    return json.loads((Path(__file__).resolve().parent / "golden" / "config_checksums.json").read_text())[key]

TOL_PRE_ABS = 5e-3


def check_against_reference(out, frame, blob):
    """u8 within 1 LSB, only next to a truncation boundary, on <= 1e-3 of the pixels."""
    r_out, r_pre = oracle.forward_y(frame, blob)
    d = np.abs(out.astype(np.int16) - r_out.astype(np.int16))
    assert d.max() <= 1
    bad = d != 0
    assert bad.mean() <= 1e-3
    frac = np.abs(r_pre[bad] - np.rint(r_pre[bad]))
    assert (frac <= TOL_PRE_ABS).all() or ((r_pre[bad] < 0) | (r_pre[bad] > 255)).all()


def shas(planes):
    return [hashlib.sha256(np.ascontiguousarray(p).tobytes()).hexdigest() for p in planes]


def test_config2_batch_of_64_fused_and_unfused(gpu_ctx, weights_blob):
    import torch
    pin = pins("c2_3840x2160")
    w, h, n = pin["width"], pin["height"], pin["frames"]
    assert (w, h, n) == (3840, 2160, 64)
    frames = synth_batch(w, h, n)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    gpu_ctx.forward_y_dev(d_in.data_ptr(), w, h * w, d_out.data_ptr(), w, h * w, w, h, n)
    gpu_ctx.synchronize()
    fused = d_out.cpu().numpy()
    assert shas(fused) == pin["gpuorder_sha256"]
    # the materialising path named by configs[2]: 64 x 32 f32 planes = 68 GB of HBM
    d_out.zero_()
    d_work = torch.empty((n, 32, h, w), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.forward_y_unfused_dev(d_in.data_ptr(), w, h * w, d_out.data_ptr(), w, h * w, w, h, n, d_work.data_ptr())
    gpu_ctx.synchronize()
    unfused = d_out.cpu().numpy()
    assert shas(unfused) == pin["gpuorder_sha256"]
    # the 32-channel map of one frame against the reference arithmetic (1e-3 relative)
    w1, b1, w2, b2, _, _ = S.split_weights(weights_blob)
    k = 37
    ref_map = oracle.conv99x11(frames[k][:64], w1, b1, w2, b2)             # top 64 rows: rows >= 4 from the cut are exact
    got_map = d_work[k, :, :60].cpu().numpy()
    assert np.abs(got_map - ref_map[:, :60]).max() <= 1e-3 * max(1.0, float(np.abs(ref_map).max()))
    del d_work
    torch.cuda.empty_cache()
    for k in (0, 31, 63):
        check_against_reference(fused[k], frames[k], weights_blob)


def test_config4_5760x3240_frame_stream(gpu_ctx, weights_blob):
    import torch
    pin = pins("c4_5760x3240")
    w, h, n = pin["width"], pin["height"], pin["frames"]
    assert (w, h) == (5760, 3240) and n >= 8
    frames = synth_batch(w, h, n)
    # host frames, uploads / kernels / downloads overlapped on two lanes (srcnn_forward_y_frames)
    streamed = gpu_ctx.forward_y_frames(frames)
    assert shas(streamed) == pin["gpuorder_sha256"]
    # the same frames as one device batch
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    gpu_ctx.forward_y_dev(d_in.data_ptr(), w, h * w, d_out.data_ptr(), w, h * w, w, h, n)
    gpu_ctx.synchronize()
    assert shas(d_out.cpu().numpy()) == pin["gpuorder_sha256"]
    # one frame alone (the single-plane work-item launch with seams)
    assert shas([gpu_ctx.forward_y(frames[3])]) == [pin["gpuorder_sha256"][3]]
    for k in (0, 4, 7):
        check_against_reference(streamed[k], frames[k], weights_blob)


@pytest.mark.parametrize("n_frames", [3, 20])
def test_batch_launch_forms_equal_single_plane(gpu_ctx, weights_blob, n_frames):
    """How a batch is launched depends on its size (csrc/srcnn_api.cpp): a few large planes run as one single-plane launch
    per frame, 17-31 of them as ONE launch that repeats the plane's work items -- row seams, column seams, the merged seam
    kernel over all frames -- frame after frame, larger batches on the regular grid.  Whatever the form, every frame must
    come out as it does launched alone (which the other tests pin to the model of the kernels' arithmetic)."""
    import torch
    w, h = 2304, 1830                               # 4.2 MPix: 18 strips, ~64-row items
    frames = synth_batch(w, h, n_frames, first_frame=40)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.full_like(d_in, 3)
    d_one = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    gpu_ctx.forward_y_dev(d_in.data_ptr(), w, h * w, d_out.data_ptr(), w, h * w, w, h, n_frames)
    gpu_ctx.synchronize()
    out = d_out.cpu().numpy()
    for k in sorted({0, 1, n_frames // 2, n_frames - 1}):
        gpu_ctx.forward_y_dev(d_in[k].data_ptr(), w, h * w, d_one.data_ptr(), w, h * w, w, h, 1)
        gpu_ctx.synchronize()
        assert np.array_equal(out[k], d_one.cpu().numpy()), k
    model, _ = oracle.gpuorder_forward_y(frames[n_frames - 1][:200], weights_blob)
    assert np.array_equal(out[n_frames - 1][:190], model[:190])         # 13x13 locality: the top rows of a crop are the frame's
