"""CPU test of the work-item planner (plan_items() in srcnn_cpp_amd/csrc/srcnn_plan.cpp) through a test hook that only the
TUNING build of the library holds (libsrcnn_amd_tuning.so; the product exports the ABI of include/srcnn_amd.h and nothing
else) -- host logic only, no device needed.

A plane launched alone is cut into per-block work items {strip, row range, seam above, seam below}.
Whatever heights the planner picks for speed, the items must tile every strip exactly once, in
order, and the seam ids must describe exactly the boundaries between vertically adjacent items:
the kernels rely on that (an item leaves the two output rows next to a seam to the seam kernel).
"""
import ctypes as C

import numpy as np
import pytest

import srcnn_cpp_amd as S

ITEM_INTS = 5
_tuning = None


def tuning_lib():
    global _tuning
    if _tuning is None:
        from srcnn_cpp_amd import build as B
        B.build()
        _tuning = C.CDLL(str(S.tuning_library_path()))
    return _tuning


def plan(n_cu, n_strips, r0, r1, skew=10, per_cu=2, seams=True):
    lib = tuning_lib()
    fn = lib.srcnn_debug_plan_items
    fn.restype = C.c_int
    items = (C.c_int * (ITEM_INTS * 4096))()
    seam = (C.c_int * (2 * 4096))()
    n_seams = C.c_int(0)
    n = fn(n_cu, n_strips, r0, r1, skew, per_cu, int(seams), items, 4096, seam, 4096, C.byref(n_seams))
    assert n >= 0
    it = np.array(items[:ITEM_INTS * n], dtype=np.int64).reshape(n, ITEM_INTS)
    se = np.array(seam[:2 * n_seams.value], dtype=np.int64).reshape(n_seams.value, 2)
    return it, se


GEOMETRIES = [(256, 5, 0, 360), (256, 4, 0, 400), (256, 2, 0, 75), (256, 30, 0, 2160), (256, 31, 0, 2160), (256, 15, 0, 1080), (256, 60, 0, 4320), (256, 62, 540, 1080),
              (256, 10, 0, 720), (256, 1, 0, 6400), (256, 2, 0, 6400), (256, 45, 100, 3340), (32, 8, 0, 1080),
              (256, 4, 0, 540), (304, 31, 0, 2160), (256, 256, 0, 2160), (256, 200, 0, 600)]


@pytest.mark.parametrize("n_cu,n_strips,r0,r1", GEOMETRIES)
@pytest.mark.parametrize("per_cu", [1, 2])
@pytest.mark.parametrize("seams", [False, True])
def test_items_tile_every_strip_exactly(n_cu, n_strips, r0, r1, per_cu, seams):
    it, se = plan(n_cu, n_strips, r0, r1, per_cu=per_cu, seams=seams)
    if len(it) == 0:
        return                                   # geometry does not qualify: the regular grid is used
    assert ((it[:, 2] - it[:, 1]) >= 0).all()
    if (it[:, 2] == it[:, 1]).any():
        # placeholders of the balanced two-per-CU plan: a CU whose rows end at a strip boundary holds a single piece
        assert per_cu == 2 and seams and len(it) == 2 * n_cu
        assert (it[it[:, 2] == it[:, 1]][:, 3:] == -1).all()
        assert (it[:n_cu, 2] > it[:n_cu, 1]).all(), "the first-dispatched block of a CU is never the empty one"
        it = it[it[:, 2] > it[:, 1]]
    elif len(it) != per_cu * n_cu:               # every workgroup slot gets exactly one item ...
        # ... except on a plane too small for that: one item per CU, fewer items than CUs, none shorter than 10 rows
        assert per_cu == 1 and seams and len(it) < n_cu and len(it) % n_strips == 0
        assert ((it[:, 2] - it[:, 1]) >= 10).all()
    used_seams = set()
    for s in range(n_strips):
        mine = it[it[:, 0] == s]
        mine = mine[np.argsort(mine[:, 1])]
        assert len(mine) >= 1
        assert mine[0, 1] == r0 and mine[-1, 2] == r1
        assert (mine[1:, 1] == mine[:-1, 2]).all(), "gap or overlap between the items of a strip"
        assert (mine[:, 2] > mine[:, 1]).all()
        assert mine[0, 3] == -1 and mine[-1, 4] == -1, "seam at the edge of the launch"
        for a, b in zip(mine[:-1], mine[1:]):
            assert a[4] == b[3], "the two items at a boundary name different seams"
            if a[4] >= 0:
                assert tuple(se[a[4]]) == (s, a[2])
                assert a[4] not in used_seams
                used_seams.add(int(a[4]))
    if len(se):
        assert used_seams == set(range(len(se)))
        assert ((it[:, 2] - it[:, 1]) >= 8).all(), "an item next to a seam must hold the 4 rows it hands over"
    else:
        assert (it[:, 3:] == -1).all()


def finish_estimate(fast_rows, slow_rows, kf=6.40, ks=8.40, ka=3.76, sf=3.63, ss=5.44):
    """The planner's model (cu_finish_estimate in srcnn_plan.cpp; constants fitted to stamped launches,
    profiles/r03/planner_fit.txt): us until a CU has finished both of its items -- paired they take kf / ks us per row, the one
    left alone ka."""
    tf, ts = sf + fast_rows * kf, ss + slow_rows * ks
    if tf <= ts:
        return tf + max(0.0, slow_rows - (tf - ss) / ks) * ka
    return ts + max(0.0, fast_rows - (ts - sf) / kf) * ka


def test_fast_and_slow_items_pair_up_per_cu():
    """Two workgroups per CU: block i and block n_cu + i share a CU (measured).  The launch ends with the slowest CU, so
    the planner equalises the estimated finish times: nearly the same number of rows everywhere, split so that both
    items of a CU end together (the first-dispatched block is the faster one and gets the taller item)."""
    for (n_strips, rows) in [(30, 2160), (60, 4320), (15, 1080), (45, 3240), (30, 1080), (31, 2160)]:
        it, _ = plan(256, n_strips, 0, rows)
        assert len(it) == 512
        h = it[:, 2] - it[:, 1]
        per_cu = h[:256] + h[256:]
        # (3 %: every second strip is shifted by 16 rows so that the seam windows of neighbouring strips stay apart -- one seam
        # launch instead of two -- which costs a single CU a few rows; the estimated finish times below are what counts)
        assert per_cu.max() - per_cu.min() <= max(5, 0.03 * per_cu.mean())
        fin = np.array([finish_estimate(f, s) for f, s in zip(h[:256], h[256:])])
        assert fin.max() <= 1.005 * np.median(fin) + 3.76         # within one row of the median CU
        assert h[:256].mean() > 1.1 * h[256:].mean()          # first-dispatched blocks are the taller ones
        assert (h >= 10).all()


def test_seam_windows_of_neighbouring_strips_stay_apart():
    """One seam launch instead of two (srcnn_seams_merged_kernel) rests on this: the four-row windows around the seams of
    NEIGHBOURING strips share no row, so the block that finishes a row seam can take the neighbours' column-seam values of
    those rows from the strip kernel's exports.  The balanced planner guarantees it for the geometries it marks separated;
    the full-size ones must be among them (and keep their balance: test_fast_and_slow_items_pair_up_per_cu)."""
    for (n_strips, rows) in [(30, 2160), (60, 4320), (15, 1080), (45, 3240), (31, 2160), (30, 1080)]:
        it, se = plan(256, n_strips, 0, rows)
        assert len(it) == 512 and len(se) > 0
        for s in range(1, n_strips):
            a, b = se[se[:, 0] == s - 1][:, 1], se[se[:, 0] == s][:, 1]
            if len(a) and len(b):
                assert np.abs(a[:, None] - b[None, :]).min() >= 4, (n_strips, rows, s)
        for s in range(n_strips):                      # ... and inside a strip two seams are at least an item apart
            b = np.sort(se[se[:, 0] == s][:, 1])
            assert (np.diff(b) >= 10).all()


def test_worker_pool_of_the_multi_gpu_entry_points():
    """srcnn_forward_y_striped* / srcnn_forward_y_frames_multi park one persistent host thread per context beyond the first
    between calls (no thread spawn per step).  The hook runs thousands of rounds over a growing number of workers: every task
    exactly once per call, the first non-zero code returned, n - 1 threads in all.  (ASan / UBSan: tests/test_sanitizers.py.)"""
    import srcnn_cpp_amd as S
    lib = tuning_lib()
    lib.srcnn_debug_worker_pool.restype = int
    assert lib.srcnn_debug_worker_pool(8, 3000) == 0
    assert lib.srcnn_debug_worker_pool(1, 10) == 0
    assert lib.srcnn_debug_worker_pool(0, 1) == -1


def _finish_estimate(fast_rows, slow_rows):
    """cu_finish_estimate() of srcnn_plan.cpp with its default rates."""
    pf, ps, alone, sf, ss = 6.40, 8.40, 3.76, 3.63, 5.44
    tf, ts = sf + fast_rows * pf, ss + slow_rows * ps
    if tf <= ts:
        return tf + max(0.0, slow_rows - (tf - ss) / ps) * alone
    return ts + max(0.0, fast_rows - (ts - sf) / pf) * alone


@pytest.mark.parametrize("n_strips,r0,r1,top,bot", [(60, 540, 1080, 2, 2), (60, 0, 540, 0, 2), (60, 3780, 4320, 2, 0), (30, 1080, 2160, 2, 2)])
def test_open_ends_of_a_stripe_weigh_in_the_balance(n_strips, r0, r1, top, bot):
    """A launch on rows of a taller plane computes two more feature rows at either open end (srcnn_mfma.hip: f_lo, f_hi).  The
    planner counts them as work of the first / last item of every strip: the plan still tiles the rows exactly, and the CU that
    finishes last -- with the edge rows counted -- finishes earlier than under the plan that ignores them."""
    n_cu = 256

    def slowest(it):
        work = (it[:, 2] - it[:, 1]).copy()
        work[it[:, 1] == r0] += top
        work[it[:, 2] == r1] += bot
        return max(_finish_estimate(work[c], work[n_cu + c]) for c in range(n_cu))

    plain, _ = plan(n_cu, n_strips, r0, r1)
    edged, se = plan(n_cu, n_strips, r0, r1, per_cu=2 + 16 * top + 256 * bot)
    assert len(plain) == len(edged) == 2 * n_cu
    for s in range(n_strips):
        rows = sorted((int(a), int(b)) for _, a, b, _, _ in edged[edged[:, 0] == s] if b > a)
        assert rows[0][0] == r0 and rows[-1][1] == r1
        assert all(rows[i][1] == rows[i + 1][0] for i in range(len(rows) - 1))
    assert slowest(edged) < slowest(plain) - 3.0            # microseconds of the model: about one row pair
