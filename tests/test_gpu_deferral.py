"""Seam deferral (srcnn_set_seam_deferral / srcnn_flush): fused float32 launches queued back to back carry
the seam blocks of the launch before them (srcnn_strip_fold_kernel) instead of paying a seam launch each.  Speed only: every
test here compares bytes -- with the launch-by-launch form, with oracle/srcnn_gpuorder.c's model of the kernels, and with the
committed checksums of the full-size planes."""
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


@pytest.fixture
def ctx(weights_blob):
    c = S.Context(0)
    c.set_weights_blob(weights_blob)
    yield c
    c.close()


def test_a_stream_of_planes_equals_launch_by_launch(ctx, weights_blob):
    """Eight DIFFERENT 1920x1080 planes queued back to back into eight outputs, deferral on: every output equals the plain form's
    (and frame 0 the CPU model of the kernels); the last plane is complete only after flush()."""
    import torch
    w, h, n = 1920, 1080, 8
    frames = synth_batch(w, h, n, first_frame=11)
    d_in = torch.from_numpy(frames).cuda()
    want = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    for k in range(n):
        ctx.forward_y_dev(d_in[k].data_ptr(), w, 0, want[k].data_ptr(), w, 0, w, h, 1)
    ctx.synchronize()
    want = want.cpu().numpy()
    assert np.array_equal(want[0], oracle.gpuorder_forward_y(frames[0], weights_blob)[0])
    got = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    ctx.set_seam_deferral(True)
    for k in range(n):
        ctx.forward_y_dev(d_in[k].data_ptr(), w, 0, got[k].data_ptr(), w, 0, w, h, 1)
    torch.cuda.synchronize()                   # everything QUEUED so far has run: the last plane's seam pixels have not been queued
    part = got.cpu().numpy()
    assert all(np.array_equal(part[k], want[k]) for k in range(n - 1))
    assert not np.array_equal(part[n - 1], want[n - 1]), "the last launch's seam work must still be pending"
    ctx.flush()
    ctx.synchronize()
    assert np.array_equal(got.cpu().numpy(), want)
    ctx.set_seam_deferral(False)


def test_any_other_call_completes_the_pending_plane(ctx, weights_blob):
    """The contract's safety net: ANY other entry point queues the pending seam work first -- a mode change, a host-buffer call,
    a change of stream, a geometry that cannot carry it (another plan), srcnn_synchronize."""
    import torch
    w, h = 1280, 720
    y = synth_luma(w, h, frame=2)
    want = oracle.gpuorder_forward_y(y, weights_blob)[0]
    d_in = torch.from_numpy(y).cuda()
    small = synth_luma(300, 70, frame=1)
    d_small, d_small_out = torch.from_numpy(small).cuda(), torch.zeros((70, 300), dtype=torch.uint8, device="cuda")
    other = torch.cuda.Stream()
    ctx.set_seam_deferral(True)
    actions = {
        "synchronize": lambda: ctx.synchronize(),
        "set_mode": lambda: (ctx.set_mode(S.MODE_EXACT), ctx.set_mode(S.MODE_MFMA)),
        "host call": lambda: ctx.forward_y(small),
        "another geometry": lambda: ctx.forward_y_dev(d_small.data_ptr(), 300, 0, d_small_out.data_ptr(), 300, 0, 300, 70, 1),
        "set_stream": lambda: (ctx.set_stream(other.cuda_stream), ctx.set_stream(0)),
        "fixup_stats": lambda: ctx.fixup_stats(),
    }
    for name, act in actions.items():
        d_out = torch.zeros_like(d_in)
        torch.cuda.synchronize()
        ctx.forward_y_dev(d_in.data_ptr(), w, 0, d_out.data_ptr(), w, 0, w, h, 1)
        act()
        torch.cuda.synchronize()
        ctx.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want), name
    assert np.array_equal(d_small_out.cpu().numpy(), oracle.gpuorder_forward_y(small, weights_blob)[0])


def test_striped_plane_steps_with_deferral(ctx, weights_blob):
    """The steps of a row-striped plane (one launch per rank per step, halo rows in buffers of their own): four stripes of a
    1920x1080 plane, three steps each, deferral on -- every stripe equals its rows of the whole plane."""
    import torch
    w, h, n = 1920, 1080, 4
    y = synth_luma(w, h, frame=7)
    want = oracle.gpuorder_forward_y(y, weights_blob)[0]
    d_y = torch.from_numpy(y).cuda()
    ctx.set_seam_deferral(True)
    out = torch.zeros_like(d_y)
    torch.cuda.synchronize()
    for _ in range(3):
        for k in range(n):
            r0, r1 = S.stripe_rows(h, n, k)
            own = d_y[r0:r1]
            top = d_y[r0 - 6:r0] if k > 0 else None
            bot = d_y[r1:r1 + 6] if k < n - 1 else None
            ctx.forward_y_rows_halo_dev(own.data_ptr(), w, r0, r1 - r0, top.data_ptr() if top is not None else 0,
                                        bot.data_ptr() if bot is not None else 0, w, out.data_ptr(), w, 0, w, h, r0, r1)
    ctx.flush()
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy(), want)


def test_batches_fold_their_own_seam_launches(ctx, weights_blob):
    """srcnn_forward_y_dev on a batch that runs as one launch per frame folds frame k's seam blocks into frame k + 1's launch by
    itself and returns with every frame's work queued (no deferral asked for): 5 x 2560x1440 equal the frames one by one."""
    import torch
    w, h, n = 2560, 1440, 5
    frames = synth_batch(w, h, n, first_frame=3)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    ctx.forward_y_dev(d_in.data_ptr(), w, w * h, d_out.data_ptr(), w, w * h, w, h, n)
    torch.cuda.synchronize()                   # no flush: the call itself queued everything
    got = d_out.cpu().numpy()
    for k in (0, n - 1):
        assert np.array_equal(got[k], oracle.gpuorder_forward_y(frames[k], weights_blob)[0]), k
    one = torch.zeros_like(d_in[2])
    ctx.forward_y_dev(d_in[2].data_ptr(), w, 0, one.data_ptr(), w, 0, w, h, 1)
    ctx.synchronize()
    assert np.array_equal(one.cpu().numpy(), got[2])


def test_full_size_planes_with_deferral(ctx):
    """configs[1] and configs[3] at size: the 3840x2160 plane and the 7680x4320 plane as 8 stripes of 540 rows, 3 steps each with
    deferral on, against the committed sha256 of the kernels' model."""
    import torch
    pin4k = json.loads((GOLD / "synthetic_4k_checksums.json").read_text())
    y = synth_luma(3840, 2160)
    d_in = torch.from_numpy(y).cuda()
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    ctx.set_seam_deferral(True)
    for _ in range(3):
        ctx.forward_y_dev(d_in.data_ptr(), 3840, 0, d_out.data_ptr(), 3840, 0, 3840, 2160, 1)
    ctx.flush()
    ctx.synchronize()
    assert hashlib.sha256(d_out.cpu().numpy().tobytes()).hexdigest() == pin4k["gpuorder_sha256"]
    pin = json.loads((GOLD / "config_checksums.json").read_text())["c3_7680x4320"]
    w, h = 7680, 4320
    d_y = torch.from_numpy(synth_luma(w, h)).cuda()
    out = torch.zeros_like(d_y)
    torch.cuda.synchronize()
    for _ in range(2):
        for k in range(8):
            r0, r1 = 540 * k, 540 * (k + 1)
            own = d_y[r0:r1]
            ctx.forward_y_rows_halo_dev(own.data_ptr(), w, r0, 540, d_y[r0 - 6:r0].data_ptr() if k else 0,
                                        d_y[r1:r1 + 6].data_ptr() if k < 7 else 0, w, out.data_ptr(), w, 0, w, h, r0, r1)
    ctx.flush()
    ctx.synchronize()
    assert hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest() == pin["gpuorder_sha256"][0]


def test_overlapping_outputs_of_different_launches_are_not_folded(ctx, weights_blob):
    """A deferred launch's seam blocks write ITS seam pixels while the next launch's work items run.  When the next launch writes
    another geometry into the same buffer (here: rows 200..899 of a second plane over the first plane's output) a stale seam
    pixel could land on a finished pixel -- the library queues the pending seam launch first instead of folding it in."""
    import torch
    w, h = 1920, 1080
    a, b = synth_luma(w, h, frame=1), synth_luma(w, h, frame=9)
    want_a, want_b = (oracle.gpuorder_forward_y(p, weights_blob)[0] for p in (a, b))
    d_a, d_b = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    d_out = torch.zeros_like(d_a)
    torch.cuda.synchronize()
    ctx.set_seam_deferral(True)
    ctx.forward_y_dev(d_a.data_ptr(), w, 0, d_out.data_ptr(), w, 0, w, h, 1)
    ctx.forward_y_rows_dev(d_b[194:906].data_ptr(), w, 194, d_out.data_ptr(), w, 0, w, h, 200, 900)
    ctx.flush()
    ctx.synchronize()
    got = d_out.cpu().numpy()
    assert np.array_equal(got[200:900], want_b[200:900]) and np.array_equal(got[:200], want_a[:200]) and np.array_equal(got[900:], want_a[900:])


def test_entry_points_that_read_their_own_launches_never_defer(ctx, weights_blob):
    """Deferral is a contract with the DIRECT caller of a device launch.  The host-buffer paths, the two-lane frame stream, the
    several-context calls and the BGR pipeline call those launches themselves and read the output right behind them: with
    deferral switched on they must still return complete planes."""
    ctx.set_seam_deferral(True)
    frames = synth_batch(1920, 1080, 4, first_frame=5)
    want = [oracle.gpuorder_forward_y(f, weights_blob)[0] for f in frames[:2]]
    assert np.array_equal(ctx.forward_y(frames[0]), want[0])                      # row bands over srcnn_forward_y_rows_dev
    streamed = ctx.forward_y_frames(frames)                                       # lanes over srcnn_forward_y_dev
    assert np.array_equal(streamed[0], want[0]) and np.array_equal(streamed[1], want[1])
    one = ctx.forward_y(frames[3])
    assert np.array_equal(streamed[3], one)
    fx = np.load(GOLD / "butterfly_bgr.npz")
    plain = S.Context(0)
    plain.set_weights_blob(weights_blob)
    assert np.array_equal(ctx.process_bgr(fx["src_bgr"], 2.0), plain.process_bgr(fx["src_bgr"], 2.0))
    others = [S.Context(0) for _ in range(2)]
    for c in others:
        c.set_weights_blob(weights_blob)
        c.set_seam_deferral(True)
    assert np.array_equal(S.forward_y_striped([ctx] + others, frames[2]), plain.forward_y(frames[2]))
    for c in others + [plain]:
        c.close()


@pytest.mark.parametrize("w,h", [(576, 576), (960, 540), (1280, 720), (700, 300), (3840, 200)])
def test_small_and_narrow_planes_defer_too(ctx, weights_blob, w, h):
    """Plans without column seams (576 and 960 columns keep their two halo columns: row seams only), plans whose seam windows
    are not kept apart, planes too small for work items at all: whatever the plan, a stream of five planes with deferral on
    equals the planes one by one -- deferred where the plan allows it, launch by launch where not."""
    import torch
    frames = synth_batch(w, h, 5, first_frame=40)
    d_in = torch.from_numpy(frames).cuda()
    got = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    ctx.set_seam_deferral(True)
    for k in range(5):
        ctx.forward_y_dev(d_in[k].data_ptr(), w, 0, got[k].data_ptr(), w, 0, w, h, 1)
    ctx.flush()
    ctx.synchronize()
    got = got.cpu().numpy()
    for k in range(5):
        assert np.array_equal(got[k], oracle.gpuorder_forward_y(frames[k], weights_blob)[0]), k


def test_a_batch_behind_a_pending_plane_does_not_free_its_seam_scratch(ctx, weights_blob):
    """a launch that does not defer shares a seam-scratch set with the pending deferred launch and may have to
    GROW it (a 4-frame batch in one launch needs four planes' seam exports) -- the pending seam launch must be queued before the
    buffer is freed, not read from freed memory afterwards.  Two deferred planes, then the batch, against the plain form."""
    import torch
    w, h = 1920, 1080
    frames = synth_batch(w, h, 6, first_frame=21)
    d_in = torch.from_numpy(frames).cuda()
    want = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    for k in range(6):
        ctx.forward_y_dev(d_in[k].data_ptr(), w, 0, want[k].data_ptr(), w, 0, w, h, 1)
    ctx.synchronize()
    want = want.cpu().numpy()
    for pre in (None, torch.zeros((h, w), dtype=torch.float32, device="cuda")):
        fresh = S.Context(0)                   # a context whose scratch has only ever held ONE plane's exports
        fresh.set_weights_blob(weights_blob)
        fresh.set_seam_deferral(True)
        got = torch.zeros_like(d_in)
        torch.cuda.synchronize()
        for k in range(2):
            fresh.forward_y_dev(d_in[k].data_ptr(), w, 0, got[k].data_ptr(), w, 0, w, h, 1)
        if pre is None:
            fresh.forward_y_dev(d_in[2].data_ptr(), w, w * h, got[2].data_ptr(), w, w * h, w, h, 4)
        else:                                  # ... a pre-clamp request never defers either (and a taller plane needs more scratch)
            big = torch.from_numpy(synth_luma(w, 2 * h, frame=5)).cuda()
            big_out = torch.zeros_like(big)
            big_pre = torch.zeros((2 * h, w), dtype=torch.float32, device="cuda")
            fresh.forward_y_dev(big.data_ptr(), w, 0, big_out.data_ptr(), w, 0, w, 2 * h, 1, d_preclamp=big_pre.data_ptr())
        fresh.flush()
        fresh.synchronize()
        got = got.cpu().numpy()
        n_checked = 6 if pre is None else 2
        for k in range(n_checked):
            assert np.array_equal(got[k], want[k]), (k, pre is None)
        fresh.close()


def test_a_chain_reads_complete_planes(ctx, weights_blob):
    """with deferral on, a launch whose INPUT is the pending launch's output (a two-pass chain X -> B -> C, or
    B handed over as a halo buffer) must see B's seam pixels finished: the pending seam launch is queued first, not folded in."""
    import torch
    w, h = 1920, 1080
    x = synth_luma(w, h, frame=3)
    b_want = oracle.gpuorder_forward_y(x, weights_blob)[0]
    c_want = oracle.gpuorder_forward_y(b_want, weights_blob)[0]
    d_x = torch.from_numpy(x).cuda()
    ctx.set_seam_deferral(True)
    for _ in range(3):                         # (a race would not show every time)
        d_b, d_c = torch.zeros_like(d_x), torch.zeros_like(d_x)
        torch.cuda.synchronize()
        ctx.forward_y_dev(d_x.data_ptr(), w, 0, d_b.data_ptr(), w, 0, w, h, 1)
        ctx.forward_y_dev(d_b.data_ptr(), w, 0, d_c.data_ptr(), w, 0, w, h, 1)
        ctx.flush()
        ctx.synchronize()
        assert np.array_equal(d_b.cpu().numpy(), b_want)
        assert np.array_equal(d_c.cpu().numpy(), c_want)
    # ... and as halo rows: the lower half of a plane whose top halo buffer is the last 6 rows of the pending launch's output
    top_src = synth_luma(w, 546, frame=8)
    top_out = oracle.gpuorder_forward_y(top_src, weights_blob)[0]            # 546 rows; its last 6 rows become the halo
    lower = synth_luma(w, 540, frame=9)
    whole = np.concatenate([top_out[-6:], lower])                             # rows 534.. of a virtual 1080-row plane
    virt = np.concatenate([np.zeros((534, w), np.uint8), whole])
    want = oracle.gpuorder_forward_y(virt, weights_blob)[0][540:]
    d_top_src, d_lower = torch.from_numpy(top_src).cuda(), torch.from_numpy(lower).cuda()
    for _ in range(3):
        d_top_out = torch.zeros((546, w), dtype=torch.uint8, device="cuda")
        d_out = torch.zeros((540, w), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        ctx.forward_y_dev(d_top_src.data_ptr(), w, 0, d_top_out.data_ptr(), w, 0, w, 546, 1)
        ctx.forward_y_rows_halo_dev(d_lower.data_ptr(), w, 540, 540, d_top_out[540:].data_ptr(), 0, w,
                                    d_out.data_ptr(), w, 540, w, 1080, 540, 1080)
        ctx.flush()
        ctx.synchronize()
        assert np.array_equal(d_out.cpu().numpy(), want)


@pytest.mark.parametrize("w,h,n_lanes", [(576, 576, 2), (1280, 720, 2), (1920, 1080, 3), (300, 70, 2)])
def test_planes_of_a_stream_on_lanes_equal_plane_by_plane(weights_blob, w, h, n_lanes):
    """srcnn_forward_y_lanes_dev (round 6): nine different planes alternately on two or three contexts of the one GPU, each a lane
    with its own stream and its own deferred seam work -- every plane equals the CPU model of the kernels, the call returns with
    everything queued and flushed (a synchronize per context is all the caller owes), and the lanes' contexts come back with their
    own deferral setting untouched."""
    import torch
    n = 9
    frames = synth_batch(w, h, n, first_frame=60)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    ctxs = [S.Context(0) for _ in range(n_lanes)]
    streams = [torch.cuda.Stream() for _ in range(n_lanes)]
    try:
        for c, st in zip(ctxs, streams):
            c.set_weights_blob(weights_blob)
            c.set_stream(st.cuda_stream)
        torch.cuda.synchronize()
        for _ in range(2):
            S.forward_y_lanes_dev(ctxs, [d_in[k].data_ptr() for k in range(n)], w, [d_out[k].data_ptr() for k in range(n)], w, w, h)
        for c in ctxs:
            c.synchronize()
        got = d_out.cpu().numpy()
        for k in range(n):
            assert np.array_equal(got[k], oracle.gpuorder_forward_y(frames[k], weights_blob)[0]), k
        # deferral was the call's own business: a plain launch afterwards is complete after a synchronize of the stream alone
        one = torch.zeros_like(d_in[0])
        ctxs[0].forward_y_dev(d_in[4].data_ptr(), w, 0, one.data_ptr(), w, 0, w, h, 1)
        streams[0].synchronize()
        assert np.array_equal(one.cpu().numpy(), got[4])
        with pytest.raises(S.SrcnnError):
            S.forward_y_lanes_dev([ctxs[0], ctxs[0]], [d_in[0].data_ptr()], w, [d_out[0].data_ptr()], w, w, h)      # the same context twice
        with pytest.raises(S.SrcnnError):
            S.forward_y_lanes_dev(ctxs, [d_in[0].data_ptr(), 0], w, [d_out[0].data_ptr(), d_out[1].data_ptr()], w, w, h)    # a null plane
    finally:
        for c in ctxs:
            c.close()
