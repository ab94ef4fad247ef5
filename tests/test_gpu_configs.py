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
    """How a batch is launched depends on its size (csrc/srcnn_plan.cpp, frames_per_launch): a few large planes run as one single-plane launch
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


def test_config3_7680x4320_as_8_stripes_of_540_rows(weights_blob):
    """configs[3] at size and in ITS partition: the 7680x4320 plane (3840x2160 x2.0) as 8 row stripes of 540 rows --
    (a) every stripe through srcnn_forward_y_rows_dev from exactly the rows a rank would hold (its own 540 + 6 halo rows
    per interior side), (b) the whole step through srcnn_forward_y_striped_dev with 8 contexts on cuda:0 (ONE launch per
    context, the neighbours' edge rows read where they lie), (c) every stripe through srcnn_forward_y_rows_halo_dev from its
    own 540 rows and two separately allocated 6-row halo buffers -- what a rank of a one-process-per-GPU job launches.  All
    must reproduce the sha256 the CPU model of
    the kernels' arithmetic gave in the build container, plane and stripe by stripe; two stripe EDGES (rows either side
    of a cut, where a halo error would show) are compared with the reference arithmetic."""
    import torch
    pin = pins("c3_7680x4320")
    w, h, n = pin["width"], pin["height"], pin["stripes"]
    assert (w, h, n) == (7680, 4320, 8)
    from srcnn_cpp_amd.synth import synth_luma
    y = synth_luma(w, h, frame=0)
    bounds = [S.stripe_rows(h, n, k) for k in range(n)]
    assert all(b - a == 540 for a, b in bounds)
    ctxs = [S.Context(0) for _ in range(n)]
    try:
        for c in ctxs:
            c.set_weights_blob(weights_blob)
        # (a) stripe by stripe, each from its own halo-extended rows only
        out_a = np.empty_like(y)
        for k, (r0, r1) in enumerate(bounds):
            s0, s1 = max(0, r0 - 6), min(h, r1 + 6)
            d_in = torch.from_numpy(np.ascontiguousarray(y[s0:s1])).cuda()
            d_out = torch.zeros((r1 - r0, w), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ctxs[k].forward_y_rows_dev(d_in.data_ptr(), w, s0, d_out.data_ptr(), w, r0, w, h, r0, r1)
            ctxs[k].synchronize()
            out_a[r0:r1] = d_out.cpu().numpy()
        assert shas([out_a[a:b] for a, b in bounds]) == pin["stripe_gpuorder_sha256"]
        assert shas([out_a]) == pin["gpuorder_sha256"]
        # (b) the striped step: 8 contexts, each holding ONLY its 540 rows
        d_ins = [torch.from_numpy(np.ascontiguousarray(y[a:b])).cuda() for a, b in bounds]
        d_outs = [torch.zeros((b - a, w), dtype=torch.uint8, device="cuda") for a, b in bounds]
        torch.cuda.synchronize()
        for _ in range(2):                                   # twice, queued back to back
            S.forward_y_striped_dev(ctxs, [t.data_ptr() for t in d_ins], w, [t.data_ptr() for t in d_outs], w, w, h)
        for c in ctxs:
            c.synchronize()
        out_b = np.concatenate([t.cpu().numpy() for t in d_outs], axis=0)
        assert shas([out_b[a:b] for a, b in bounds]) == pin["stripe_gpuorder_sha256"]
        assert shas([out_b]) == pin["gpuorder_sha256"]
        # (c) a rank's one launch: own rows + 6-row halo buffers of their own
        out_c = np.empty_like(y)
        for k, (r0, r1) in enumerate(bounds):
            top = torch.from_numpy(np.ascontiguousarray(y[r0 - 6:r0])).cuda() if k > 0 else None
            bot = torch.from_numpy(np.ascontiguousarray(y[r1:r1 + 6])).cuda() if k < n - 1 else None
            d_out = torch.zeros((r1 - r0, w), dtype=torch.uint8, device="cuda")
            torch.cuda.synchronize()
            ctxs[k].forward_y_rows_halo_dev(d_ins[k].data_ptr(), w, r0, r1 - r0, top.data_ptr() if top is not None else 0,
                                            bot.data_ptr() if bot is not None else 0, w, d_out.data_ptr(), w, r0, w, h, r0, r1)
            ctxs[k].synchronize()
            out_c[r0:r1] = d_out.cpu().numpy()
        assert shas([out_c[a:b] for a, b in bounds]) == pin["stripe_gpuorder_sha256"]
        assert shas([out_c]) == pin["gpuorder_sha256"]
        # reference arithmetic around two cuts: rows [cut - 40, cut + 40) -- the oracle on a crop is exact >= 6 rows from its ends
        for cut in (bounds[1][0], bounds[5][0]):
            a, b = cut - 46, cut + 46
            r_out, r_pre = oracle.forward_y(y[a:b], weights_blob)
            got = out_b[a + 6:b - 6]
            d = np.abs(got.astype(np.int16) - r_out[6:-6].astype(np.int16))
            assert d.max() <= 1 and (d != 0).mean() <= 1e-3
            frac = np.abs(r_pre[6:-6][d != 0] - np.rint(r_pre[6:-6][d != 0]))
            assert (frac <= TOL_PRE_ABS).all()
    finally:
        for c in ctxs:
            c.close()
