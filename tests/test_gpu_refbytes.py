"""SRCNN_MODE_REFBYTES: the reference's BYTES from the MFMA path (VERDICT r02 "missing" 2).

The float32 MFMA kernel differs from the reference arithmetic by rounding noise (<= 4.4e-4 before truncation on 54 MPix of varied content), so a byte can
differ only where the pre-truncation value lies next to an integer (src/srcnn.cpp:238-240 truncates).  In this mode the fused
kernel flags those pixels (|v - rint(v)| <= delta, delta derived from the model: 1.38e-3 for the shipped one, ~0.3 % of the
pixels) and ONE kernel behind it recomputes exactly them -- whole 12 x 12 tiles where flat or periodic content flags a region --
in the reference's arithmetic (csrc/srcnn_exact.hip), watches the margin and leaves a verdict that a second kernel acts on: a
launch whose monitored deviation exceeds delta / 2 is redone in the reference's arithmetic on every pixel, on the device.
Every test here compares with ``oracle.forward_y`` BIT FOR BIT: no tolerance anywhere in this file.
"""
import hashlib
import json
from pathlib import Path

import numpy as np
import pytest

import oracle
import srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_batch, synth_luma

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"

EDGE_SIZES = [(1, 1), (3, 3), (9, 5), (5, 9), (17, 4), (2, 40), (40, 2), (31, 7), (33, 9), (12, 12), (13, 25), (24, 11)]
TILE_SIZES = [(123, 11), (124, 13), (127, 6), (128, 20), (129, 10), (249, 8), (300, 70), (97, 61), (640, 360), (1000, 333)]


@pytest.fixture(params=[S.MODE_REFBYTES, S.MODE_REFBYTES16], ids=["f32-mfma", "split-f16"])
def ref_ctx(weights_blob, request):
    """Both flag-and-recompute modes: behind the float32 MFMA kernel (SRCNN_MODE_REFBYTES) and, opt-in, behind the split-f16
    kernel (SRCNN_MODE_REFBYTES16, threshold 8/6 of the former's)."""
    ctx = S.Context(0)
    ctx.set_weights_blob(weights_blob)
    ctx.set_mode(request.param)
    ctx.mode_under_test = request.param
    yield ctx
    ctx.close()


@pytest.mark.parametrize("w,h", EDGE_SIZES + TILE_SIZES)
def test_reference_bytes_on_every_size(ref_ctx, weights_blob, w, h):
    for frame in (0, 5):
        y = synth_luma(w, h, frame=frame)
        assert np.array_equal(ref_ctx.forward_y(y), oracle.forward_y(y, weights_blob)[0])


@pytest.mark.parametrize("value", [0, 1, 16, 77, 128, 200, 235, 255])
def test_constant_planes_take_the_dense_tile_path(ref_ctx, weights_blob, value):
    """A flat region gives ONE pre-truncation value: either no pixel of it is flagged or every one -- whole tiles are then
    recomputed (a letterbox must not cost 25 feature positions per pixel)."""
    y = np.full((150, 333), value, np.uint8)
    before = ref_ctx.fixup_stats()
    assert np.array_equal(ref_ctx.forward_y(y), oracle.forward_y(y, weights_blob)[0])
    after = ref_ctx.fixup_stats()
    assert after["scattered_pixels"] - before["scattered_pixels"] <= 10 * 333 * 150 // 144 + 2000   # at most 10 per tile


def test_periodic_and_saturating_content(ref_ctx, weights_blob):
    rng = np.random.default_rng(3)
    yy, xx = np.mgrid[0:400, 0:700]
    planes = {
        "checkerboard": np.where(((yy // 8) + (xx // 8)) % 2 == 0, 16, 240).astype(np.uint8),
        "stripes": np.where((xx // 3) % 2 == 0, 30, 220).astype(np.uint8),
        "white noise": rng.integers(0, 256, (400, 700), dtype=np.uint8),
        "bright ramp": np.clip(200 + xx * 55 // 700 + rng.integers(0, 4, (400, 700)), 0, 255).astype(np.uint8),
        "letterbox": np.concatenate([np.full((60, 700), 16, np.uint8), synth_luma(700, 280, frame=2), np.full((60, 700), 16, np.uint8)]),
    }
    for name, y in planes.items():
        assert np.array_equal(ref_ctx.forward_y(y), oracle.forward_y(y, weights_blob)[0]), name


def test_full_4k_frame_has_the_oracle_sha(ref_ctx, weights_blob):
    """The plane the MFMA mode gets 259 bytes 'wrong' (tests/golden/synthetic_4k_checksums.json): here sha256 == the
    reference arithmetic's, and the statistics say what it took."""
    pin = json.loads((GOLD / "synthetic_4k_checksums.json").read_text())
    y = synth_luma(3840, 2160)
    assert hashlib.sha256(y.tobytes()).hexdigest() == pin["input_sha256"]
    before = ref_ctx.fixup_stats()
    out = ref_ctx.forward_y(y)
    st = ref_ctx.fixup_stats()
    assert hashlib.sha256(out.tobytes()).hexdigest() == pin["oracle_sha256"]
    flagged = st["scattered_pixels"] - before["scattered_pixels"]
    changed = st["bytes_changed"] - before["bytes_changed"]
    if ref_ctx.mode_under_test == S.MODE_REFBYTES:
        # round 6: flagged against the PER-PIXEL threshold min(delta, k * 2^-24 * S1 + abs), mean 0.57 delta on this plane
        # (+ the 3 % of pixels around strip boundaries, which keep delta): ~0.15 % of the pixels instead of 0.28 %
        assert 0.0010 * y.size < flagged < 0.0020 * y.size
        assert changed == pin["u8_mismatches_between_them"]     # exactly the bytes the MFMA mode differs on
        assert abs(st["delta"] - 1.376e-3) < 2e-5
        k, ratio = ref_ctx.fixup_local_stats()
        assert abs(k - 1.6) < 1e-3 and 0 < ratio < 0.5, (k, ratio)      # no deviation above half its own pixel's threshold
        # ... and against the one global threshold of rounds 3-5 (srcnn_set_fixup_local(ctx, 0)): same bytes, ~2 x the pixels
        ref_ctx.set_fixup_local(0.0)
        out0 = ref_ctx.forward_y(y)
        st0 = ref_ctx.fixup_stats()
        ref_ctx.set_fixup_local(0.4)
        assert np.array_equal(out0, out)
        flagged0 = st0["scattered_pixels"] - st["scattered_pixels"]
        assert 0.002 * y.size < flagged0 < 0.008 * y.size and flagged < 0.62 * flagged0, (flagged, flagged0)
    else:
        assert 0.0015 * y.size < flagged < 0.005 * y.size
        assert 200 <= changed <= 450 and abs(st["delta"] - 1.834e-3) < 2e-5
        assert abs(ref_ctx.fixup_local_stats()[0] - 2.15) < 1e-3      # the split-f16 kernel's own factor (its noise is wider)
    assert 0 < st["max_dev"] < 0.5 * st["delta"], st            # the margin: |v_fast - v_ref| seen on the flagged sample


_ORACLE_8K = {}


def test_configs3_plane_and_its_540_row_stripes(ref_ctx, weights_blob):
    """BASELINE configs[3]'s plane (7680 x 4320, 33 MPix) in the REFBYTES modes: one launch on device memory, the host entry
    point (row bands with overlapped transfers, one fix-up per band) and the 8 x 540-row stripes of the multi-GPU split each
    from its own halo-extended rows -- every byte equal to the reference arithmetic (computed once on the host's cores)."""
    import torch
    w, h = 7680, 4320
    y = synth_luma(w, h)
    if "out" not in _ORACLE_8K:
        _ORACLE_8K["out"] = oracle.forward_y(y, weights_blob)[0]
    want = _ORACLE_8K["out"]
    d_in = torch.from_numpy(y).cuda()
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    ref_ctx.forward_y_dev(d_in.data_ptr(), w, 0, d_out.data_ptr(), w, 0, w, h, 1)
    ref_ctx.synchronize()
    assert np.array_equal(d_out.cpu().numpy(), want), "one launch"
    assert np.array_equal(ref_ctx.forward_y(y), want), "host entry point (row bands)"
    out = np.zeros_like(y)
    for k in range(8):
        r0, r1 = 540 * k, 540 * (k + 1)
        s0, s1 = max(0, r0 - 6), min(h, r1 + 6)
        d_s = d_in[s0:s1].contiguous()
        d_o = torch.zeros((540, w), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ref_ctx.forward_y_rows_dev(d_s.data_ptr(), w, s0, d_o.data_ptr(), w, r0, w, h, r0, r1)
        ref_ctx.synchronize()
        out[r0:r1] = d_o.cpu().numpy()
    assert np.array_equal(out, want), "8 x 540-row stripes"
    st = ref_ctx.fixup_stats()
    assert 0 < st["max_dev"] < 0.5 * st["delta"], st


def test_batches_stripes_and_device_entry_points(ref_ctx, weights_blob):
    import torch
    w, h, n = 1920, 1080, 5
    frames = synth_batch(w, h, n, first_frame=20)
    want = [oracle.forward_y(f, weights_blob)[0] for f in frames]
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    ref_ctx.forward_y_dev(d_in.data_ptr(), w, w * h, d_out.data_ptr(), w, w * h, w, h, n)
    ref_ctx.synchronize()
    got = d_out.cpu().numpy()
    for k in range(n):
        assert np.array_equal(got[k], want[k]), k
    # host frame stream (two lanes, overlapped transfers)
    streamed = ref_ctx.forward_y_frames(frames)
    assert all(np.array_equal(streamed[k], want[k]) for k in range(n))
    # row stripes of frame 0, each from its own halo-extended rows, thin and odd ones included
    out = np.zeros_like(frames[0])
    bounds = [0, 7, 200, 201, 540, 1073, h]
    for r0, r1 in zip(bounds[:-1], bounds[1:]):
        s0, s1 = max(0, r0 - 6), min(h, r1 + 6)
        d_s = torch.from_numpy(np.ascontiguousarray(frames[0][s0:s1])).cuda()
        d_o = torch.zeros((r1 - r0, w), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ref_ctx.forward_y_rows_dev(d_s.data_ptr(), w, s0, d_o.data_ptr(), w, r0, w, h, r0, r1)
        ref_ctx.synchronize()
        out[r0:r1] = d_o.cpu().numpy()
    assert np.array_equal(out, want[0])
    # strided planes
    d_big = torch.zeros((h, w + 96), dtype=torch.uint8, device="cuda")
    d_big[:, :w] = d_in[1]
    d_bo = torch.full((h, w + 160), 9, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ref_ctx.forward_y_dev(d_big.data_ptr(), w + 96, 0, d_bo.data_ptr(), w + 160, 0, w, h, 1)
    ref_ctx.synchronize()
    res = d_bo.cpu().numpy()
    assert np.array_equal(res[:, :w], want[1]) and (res[:, w:] == 9).all()


def test_fixup_batches_span_frames_with_any_pitch(ref_ctx, weights_blob):
    """srcnn_forward_y_dev finishes up to 16 frames of a batch with ONE fix-up (pixel codes carry the frame; the lists, the flag
    planes and the item draw are shared): 19 frames = a full batch and a partial one, frame pitches that are not width x height
    and differ between input and output, a constant frame (dense tiles) between textured ones, a scattered group that straddles
    two frames."""
    import torch
    w, h, n = 203, 119, 19
    rng = np.random.default_rng(77)
    frames = synth_batch(w, h, n, first_frame=3)
    frames[4] = 38                                         # a flat 38 comes out at 37.99993: every interior tile is dense
    frames[11] = rng.integers(0, 256, (h, w), dtype=np.uint8)
    want = [oracle.forward_y(f, weights_blob)[0] for f in frames]
    sp, dp, ss, ds = h * (w + 5) + 64, h * (w + 9) + 192, w + 5, w + 9
    d_in = torch.zeros(n * sp, dtype=torch.uint8, device="cuda")
    for k in range(n):
        d_in[k * sp:k * sp + h * ss].view(h, ss)[:, :w] = torch.from_numpy(frames[k]).cuda()
    d_out = torch.full((n * dp,), 5, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    before = ref_ctx.fixup_stats()
    ref_ctx.forward_y_dev(d_in.data_ptr(), ss, sp, d_out.data_ptr(), ds, dp, w, h, n)
    ref_ctx.synchronize()
    st = ref_ctx.fixup_stats()
    got = d_out.cpu().numpy()
    for k in range(n):
        plane = got[k * dp:k * dp + h * ds].reshape(h, ds)
        assert np.array_equal(plane[:, :w], want[k]), k
        assert (plane[:, w:] == 5).all() and (got[k * dp + h * ds:(k + 1) * dp] == 5).all(), k      # nothing outside the planes
    assert st["dense_tiles"] - before["dense_tiles"] >= (w // 12 - 2) * (h // 12 - 2)            # the constant frame's interior
    assert st["scattered_pixels"] > before["scattered_pixels"]


def test_preclamp_request_gets_the_reference_float(ref_ctx, weights_blob):
    y = synth_luma(300, 70, frame=1)
    pre = np.empty(y.shape, np.float32)
    out = ref_ctx.forward_y(y, preclamp=pre)
    r_out, r_pre = oracle.forward_y(y, weights_blob)
    assert np.array_equal(out, r_out) and np.array_equal(pre, r_pre)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_models_scale_their_threshold(seed):
    """delta follows the model (4 * 2^-24 * ||W3|| * bound of the layer-2 map): other weights, other magnitudes, same result --
    the reference's bytes."""
    rng = np.random.default_rng(seed)
    w1 = (rng.standard_normal(5184) * 0.12).astype(np.float32)
    b1 = (rng.standard_normal(64) * 20 + 10).astype(np.float32)
    w2 = (rng.standard_normal(2048) * 0.15 * seed).astype(np.float32)
    b2 = (rng.standard_normal(32) * 10).astype(np.float32)
    w3 = (rng.standard_normal(800) * 0.02).astype(np.float32)
    b3 = np.float32(20.0)
    blob = np.concatenate([b1, w1, b2, w2, [b3], w3]).astype(np.float32)
    y = synth_luma(500, 260, frame=seed)
    with S.Context(0) as ctx:
        ctx.set_weights(w1, b1, w2, b2, w3, b3)
        ctx.set_mode(S.MODE_REFBYTES)
        out = ctx.forward_y(y)
        st = ctx.fixup_stats()
    assert np.array_equal(out, oracle.forward_y(y, blob)[0])
    assert st["max_dev"] < 0.5 * st["delta"], st


@pytest.mark.parametrize("mode", [S.MODE_REFBYTES, S.MODE_REFBYTES16], ids=["f32-mfma", "split-f16"])
def test_pipeline_reproduces_the_reference_picture(weights_blob, mode):
    """BGR in, BGR out (src/srcnn.cpp:505-659) in REFBYTES mode: the reference's own butterfly-srcnn.png, all 995,328 bytes --
    what only SRCNN_MODE_EXACT did before, at a third of the speed."""
    fx = np.load(GOLD / "butterfly_bgr.npz")
    with S.Context(0) as ctx:
        ctx.set_weights_blob(weights_blob)
        ctx.set_mode(mode)
        out = ctx.process_bgr(fx["src_bgr"], 1.5)
    assert np.array_equal(out, fx["ref_bgr"])


# ---- the threshold under attack, and the monitor acting ------------------------------------------------

def test_adversarial_windows_keep_the_reference_bytes(ref_ctx, weights_blob):
    """The windows an adversarial search drove to the largest |v_mfma - v_reference| (tests/checks/fixup_adversarial.py,
    150,000 restarts on the CPU models; profiles/r04/fixup_adversarial.txt), tiled into one plane: both REFBYTES modes return
    the reference's bytes, the monitor stays below half the threshold (the worst of THESE windows sits at 0.47 of round 5's delta;
    the GPU-side search of round 5 went further: next test), and in the MFMA mode the centre pixels carry exactly the
    value the search predicted (the CPU model it searched IS the kernel's arithmetic)."""
    from test_adversarial import FIX, tile_windows
    fx = np.load(FIX)
    plane, cy, cx = tile_windows(fx["shipped_windows"])
    r_out, _ = oracle.forward_y(plane, weights_blob)
    assert np.array_equal(ref_ctx.forward_y(plane), r_out)
    st = ref_ctx.fixup_stats()
    assert 0 < st["max_dev"] < 0.5 * st["delta"], st
    if ref_ctx.mode_under_test == S.MODE_REFBYTES:
        with S.Context(0) as ctx:
            ctx.set_weights_blob(weights_blob)
            pre = np.empty(plane.shape, np.float32)
            ctx.forward_y(plane, preclamp=pre)
        assert np.array_equal(pre[cy, cx], fx["shipped_vals"][:, 1])
        # the worst window found sits at 0.47 delta: the monitor sees deviations of that size when such a pixel is flagged
        assert np.abs(pre[cy, cx] - fx["shipped_vals"][:, 0]).max() > 0.4 * st["delta"]


def test_windows_of_the_gpu_side_search_keep_the_reference_bytes(weights_blob):
    """tests/checks/adversarial_gpu.py (round 5) searched on the GPU itself, 35 million window evaluations per kernel: the float32
    MFMA kernel reaches 7.9e-4 = 0.58 of REFBYTES' threshold, the split-f16 kernel 1.13e-3 = 0.62 of REFBYTES16's.  The committed
    windows (tests/golden/adversarial_windows_gpu.npz), tiled into one plane: each kernel shows exactly the recorded deviation at
    its windows' centres, and both byte-exact modes return the reference's bytes."""
    from test_adversarial import FIX_GPU, tile_windows
    fx = np.load(FIX_GPU)
    wins = np.concatenate([fx["mfma_windows"], fx["split16_windows"]])
    n_m = len(fx["mfma_windows"])
    plane, cy, cx = tile_windows(wins)
    r_out, r_pre = oracle.forward_y(plane, weights_blob)
    with S.Context(0) as ctx:
        ctx.set_weights_blob(weights_blob)
        for mode, sl, rec in ((S.MODE_MFMA, slice(0, n_m), fx["mfma_dev"]), (S.MODE_SPLIT16, slice(n_m, None), fx["split16_dev"])):
            ctx.set_mode(mode)
            pre = np.empty(plane.shape, np.float32)
            ctx.forward_y(plane, preclamp=pre)
            dev = np.abs(pre[cy, cx].astype(np.float64) - r_pre[cy, cx])[sl]
            assert np.allclose(dev, rec, rtol=0, atol=1e-7), (mode, dev, rec)
    for mode, rec in ((S.MODE_REFBYTES, fx["mfma_dev"]), (S.MODE_REFBYTES16, fx["split16_dev"])):
        with S.Context(0) as ctx:
            ctx.set_weights_blob(weights_blob)
            ctx.set_mode(mode)
            assert np.array_equal(ctx.forward_y(plane), r_out), mode
            st = ctx.fixup_stats()
            # the largest deviation any search has produced for the mode's kernel keeps a factor 1.5 below the mode's threshold
            assert st["exact_reruns"] == 0 and float(rec.max()) < st["delta"] / 1.5, (mode, st)


@pytest.mark.parametrize("k", range(8))
def test_adversarial_windows_of_random_models(k):
    from test_adversarial import FIX, tile_windows
    fx = np.load(FIX)
    if k >= len(fx["random_blobs"]):
        pytest.skip("fewer models in the fixture")
    blob = fx["random_blobs"][k]
    plane, _, _ = tile_windows(fx["random_windows"][k])
    r_out, _ = oracle.forward_y(plane, blob)
    with S.Context(0) as ctx:
        ctx.set_weights_blob(blob)
        for mode in (S.MODE_REFBYTES, S.MODE_REFBYTES16):
            ctx.set_mode(mode)
            try:
                out = ctx.forward_y(plane)
            except S.SrcnnError as e:
                assert mode == S.MODE_REFBYTES16 and e.code == S.ERR_STATE      # the model exceeds that mode's f16 ranges
                continue
            assert np.array_equal(out, r_out), mode
            st = ctx.fixup_stats()
            assert st["max_dev"] < 0.5 * st["delta"], (mode, st)


def test_strict_mode_redoes_a_launch_whose_margin_is_gone(weights_blob):
    """The safety net (srcnn_set_fixup_strict, ON by default): with the threshold cut to its floor (srcnn_set_fixup_margin(0.25):
    delta = 1.5e-4, below the noise) the monitor of nearly every launch exceeds delta / 2 -- the last workgroup of fix_apply_kernel
    notices and fix_rerun_kernel, queued behind it, redoes the launch in the reference's arithmetic on every pixel, all on the
    device: the reference's bytes although the threshold no longer covers the noise, on a plane, a batch, row stripes from one
    buffer and row stripes with their halo rows in buffers of their own.  With the net switched off the same threshold lets
    wrong bytes through (that is what the margin is for)."""
    import torch
    w, h = 1000, 600
    frames = synth_batch(w, h, 3, first_frame=2)
    want = np.stack([oracle.forward_y(f, weights_blob)[0] for f in frames])
    with S.Context(0) as ctx:
        ctx.set_weights_blob(weights_blob)
        ctx.set_mode(S.MODE_REFBYTES)
        with pytest.raises(S.SrcnnError):
            ctx.set_fixup_margin(0.0)
        ctx.set_fixup_margin(0.25)
        assert ctx.fixup_stats()["delta"] < 1.6e-4
        ctx.set_fixup_strict(False)
        loose = np.stack([ctx.forward_y(f) for f in frames])
        n_wrong = int((loose != want).sum())
        assert ctx.fixup_stats()["exact_reruns"] == 0
        ctx.set_fixup_strict(True)
        assert np.array_equal(ctx.forward_y(frames[0]), want[0])
        r1 = ctx.fixup_stats()["exact_reruns"]
        assert r1 >= 1, "the monitor must have tripped: delta is below the measured noise"
        assert n_wrong > 0 or r1 >= 1
        # a device batch (one fix-up for its frames), row stripes, stripes with separate halo buffers
        d_in = torch.from_numpy(frames).cuda()
        d_out = torch.zeros_like(d_in)
        torch.cuda.synchronize()
        ctx.forward_y_dev(d_in.data_ptr(), w, w * h, d_out.data_ptr(), w, w * h, w, h, 3)
        ctx.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want)
        out = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
        for k in range(3):
            r0, r1_ = S.stripe_rows(h, 3, k)
            s0, s1 = max(0, r0 - 6), min(h, r1_ + 6)
            ext = d_in[1, s0:s1].contiguous()
            torch.cuda.synchronize()
            ctx.forward_y_rows_dev(ext.data_ptr(), w, s0, out.data_ptr(), w, 0, w, h, r0, r1_)
        ctx.synchronize()
        assert np.array_equal(out.cpu().numpy(), want[1])
        out.zero_()
        for k in range(3):
            r0, r1_ = S.stripe_rows(h, 3, k)
            own = d_in[2, r0:r1_].contiguous()
            top = d_in[2, r0 - 6:r0].contiguous() if k > 0 else None
            bot = d_in[2, r1_:r1_ + 6].contiguous() if k < 2 else None
            torch.cuda.synchronize()
            ctx.forward_y_rows_halo_dev(own.data_ptr(), w, r0, r1_ - r0, top.data_ptr() if k > 0 else 0, bot.data_ptr() if k < 2 else 0,
                                        w, out.data_ptr(), w, 0, w, h, r0, r1_)
        ctx.synchronize()
        assert np.array_equal(out.cpu().numpy(), want[2])
        assert ctx.fixup_stats()["exact_reruns"] > r1
        # back at the default margin nothing trips
        ctx.set_fixup_margin(4.0)
        before = ctx.fixup_stats()["exact_reruns"]
        assert np.array_equal(ctx.forward_y(frames[0]), want[0])
        assert ctx.fixup_stats()["exact_reruns"] == before


@pytest.mark.parametrize("w,h,n", [(300, 203, 3), (1000, 97, 4), (2050, 333, 5)])
def test_strict_rerun_of_stripes_with_any_row_count(weights_blob, w, h, n):
    """The device-side re-run of a row range: every 12 x 12 tile of the launch's rows as a dense tile, the last band cut by
    row_end, stripes of 67 / 68 rows etc. with their halo rows in buffers of their own (fix_src_at): the reference's bytes and no
    out-of-bounds read (round 4's workspace form faulted here on ranges no multiple of 4 tall)."""
    import torch
    y = synth_luma(w, h, frame=6)
    want = oracle.forward_y(y, weights_blob)[0]
    with S.Context(0) as ctx:
        ctx.set_weights_blob(weights_blob)
        ctx.set_mode(S.MODE_REFBYTES)
        ctx.set_fixup_margin(0.25)
        out = torch.zeros((h, w), dtype=torch.uint8, device="cuda")
        keep = []
        for k in range(n):
            r0, r1 = S.stripe_rows(h, n, k)
            own = torch.from_numpy(np.ascontiguousarray(y[r0:r1])).cuda()
            top = torch.from_numpy(np.ascontiguousarray(y[r0 - 6:r0])).cuda() if k > 0 else None
            bot = torch.from_numpy(np.ascontiguousarray(y[r1:r1 + 6])).cuda() if k < n - 1 else None
            keep += [own, top, bot]
            torch.cuda.synchronize()
            ctx.forward_y_rows_halo_dev(own.data_ptr(), w, r0, r1 - r0, top.data_ptr() if top is not None else 0,
                                        bot.data_ptr() if bot is not None else 0, w, out.data_ptr(), w, 0, w, h, r0, r1)
        ctx.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        assert ctx.fixup_stats()["exact_reruns"] >= 1


@pytest.mark.parametrize("mode", [S.MODE_REFBYTES, S.MODE_REFBYTES16], ids=["f32-mfma", "split-f16"])
def test_rerun_in_a_queued_frame_stream(weights_blob, mode):
    """The re-run never leaves the device, so it works inside the two-lane host frame stream (srcnn_forward_y_frames alternates
    between two streams; round 4's host-side re-run shared one workspace between them -- advisor, round 4) and behind a batch
    whose frames share ONE fix-up: six frames with the threshold below the noise, every byte the reference's, and the re-run
    counted per fix-up."""
    w, h, n = 640, 360, 6
    frames = synth_batch(w, h, n, first_frame=9)
    want = [oracle.forward_y(f, weights_blob)[0] for f in frames]
    with S.Context(0) as ctx:
        ctx.set_weights_blob(weights_blob)
        ctx.set_mode(mode)
        ctx.set_fixup_margin(0.25)
        got = ctx.forward_y_frames(frames)
        assert all(np.array_equal(got[k], want[k]) for k in range(n))
        r = ctx.fixup_stats()["exact_reruns"]
        assert r >= n - 1, r                    # (a frame whose sampled deviation happens to stay below delta / 2 needs none)
        ctx.set_fixup_margin(4.0)
        got = ctx.forward_y_frames(frames)
        assert all(np.array_equal(got[k], want[k]) for k in range(n))
        assert ctx.fixup_stats()["exact_reruns"] == r


def test_forced_rerun_recomputes_every_pixel(weights_blob):
    """fix_rerun_kernel alone: the tuning build's SRCNN_DEBUG_FORCE_RERUN makes the verdict 'redo' for every launch whatever the
    monitor saw.  Constant, textured and odd-sized planes (segments and tiles cut by the plane's edges, strides that are no
    multiple of 4) come out as the reference's bytes, and the counter says every launch was redone."""
    import os
    import subprocess
    import sys
    code = r"""
import numpy as np, oracle, srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_luma
S.use_library(S.tuning_library_path())      # the knob exists in the tuning build only
blob = S.load_weights()
with S.Context(0) as ctx:
    ctx.set_weights_blob(blob)
    ctx.set_mode(S.MODE_REFBYTES)
    n = 0
    for (w, h) in [(1, 1), (13, 25), (97, 61), (203, 119), (640, 360), (1000, 333)]:
        for y in (synth_luma(w, h, frame=3), np.full((h, w), 38, np.uint8)):
            assert np.array_equal(ctx.forward_y(y), oracle.forward_y(y, blob)[0]), (w, h)
            n += 1
    st = ctx.fixup_stats()
    assert st["exact_reruns"] == n, (st, n)
print("ok")
"""
    env = dict(os.environ, SRCNN_DEBUG_FORCE_RERUN="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600,
                       cwd=str(Path(__file__).resolve().parent.parent))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_both_forms_of_fix_apply_give_the_reference_bytes(weights_blob):
    """Round 6: fix_apply_kernel exists twice -- weights through scalar loads (5 workgroups per CU; planes above 2.4 MPix) and both
    weight tables in LDS (3 per CU; small planes, whose items are one round of the draw).  The library picks by plane size; the
    tuning build's SRCNN_DEBUG_FIX_LDS forces either form: a textured plane, a letterboxed one (dense tiles) and a batch of three
    frames come out as the reference's bytes both ways, with the same statistics."""
    import os
    import subprocess
    import sys
    code = r"""
import json, numpy as np, oracle, srcnn_cpp_amd as S
from srcnn_cpp_amd.synth import synth_batch, synth_luma
S.use_library(S.tuning_library_path())
blob = S.load_weights()
out = {}
with S.Context(0) as ctx:
    ctx.set_weights_blob(blob)
    ctx.set_mode(S.MODE_REFBYTES)
    y = synth_luma(1000, 333, frame=4)
    box = np.concatenate([np.full((48, 700), 38, np.uint8), synth_luma(700, 200, frame=2), np.full((48, 700), 38, np.uint8)])
    for name, p in (("textured", y), ("letterbox", box)):
        assert np.array_equal(ctx.forward_y(p), oracle.forward_y(p, blob)[0]), name
    frames = synth_batch(640, 360, 3, first_frame=7)
    got = ctx.forward_y_frames(frames)
    assert all(np.array_equal(got[k], oracle.forward_y(frames[k], blob)[0]) for k in range(3))
    st = ctx.fixup_stats()
    print(json.dumps({k: st[k] for k in ("scattered_pixels", "dense_tiles", "bytes_changed", "exact_reruns")}))
"""
    stats = []
    for lds in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SRCNN_DEBUG_FIX_LDS=lds), capture_output=True, text=True,
                           timeout=600, cwd=str(Path(__file__).resolve().parent.parent))
        assert r.returncode == 0, r.stdout + r.stderr[-2000:]
        stats.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert stats[0] == stats[1] and stats[0]["dense_tiles"] > 0 and stats[0]["scattered_pixels"] > 0 and stats[0]["exact_reruns"] == 0
