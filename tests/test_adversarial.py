"""The attack on SRCNN_MODE_REFBYTES' flag threshold, CPU side (oracle/adversarial.c, tests/checks/fixup_adversarial.py).

* the one-pixel evaluator of the search is, bit for bit, the centre pixel of the two whole-plane restatements -- so a
  deviation the search reports is one the GPU path would show (oracle/srcnn_gpuorder.c is bitwise the float32 MFMA kernels);
* the committed worst windows (tests/golden/adversarial_windows.npz) reproduce their recorded deviation, and every one of
  them -- the largest |v_gpu - v_ref| a search of 150,000 restarts could drive the shipped model to, and the same for random
  models -- stays below HALF the threshold the library derives for its model;
* on a plane tiled with those windows the selection rule of the mode still yields the reference's bytes.
tests/test_gpu_refbytes.py runs the same planes through the kernels in both REFBYTES modes."""
from pathlib import Path

import numpy as np
import pytest

import oracle
import srcnn_cpp_amd as S

from test_refbytes_model import shipped_delta

FIX = Path(__file__).resolve().parent / "golden" / "adversarial_windows.npz"
FIX_GPU = Path(__file__).resolve().parent / "golden" / "adversarial_windows_gpu.npz"     # tests/checks/adversarial_gpu.py (round 5)


def tile_windows(wins, cell=24):
    """Every 13 x 13 window in the middle of its own cell, the cell's other pixels replicating the window's edge: the window's
    centre pixel sees exactly the window.  -> (plane, centre rows, centre columns)"""
    n = len(wins)
    cols = int(np.ceil(np.sqrt(n)))
    rows = (n + cols - 1) // cols
    plane = np.zeros((rows * cell, cols * cell), np.uint8)
    cy, cx = [], []
    off = (cell - 13) // 2
    for k, w in enumerate(wins):
        r, c = divmod(k, cols)
        plane[r * cell:(r + 1) * cell, c * cell:(c + 1) * cell] = np.pad(w, ((off, cell - 13 - off), (off, cell - 13 - off)), mode="edge")
        cy.append(r * cell + off + 6)
        cx.append(c * cell + off + 6)
    return plane, np.array(cy), np.array(cx)


def test_point_evaluator_is_the_centre_pixel_of_both_restatements(weights_blob):
    rng = np.random.default_rng(3)
    for k in range(12):
        w = rng.integers(0, 256, (13, 13), dtype=np.uint8) if k % 2 else np.full((13, 13), rng.integers(0, 256), np.uint8)
        v_ref, v_gpu = oracle.adv_point(w, weights_blob)
        assert np.float32(v_ref) == oracle.forward_y(w, weights_blob)[1][6, 6]
        assert np.float32(v_gpu) == oracle.gpuorder_forward_y(w, weights_blob)[1][6, 6]


def test_search_climbs(weights_blob):
    rng = np.random.default_rng(4)
    st = rng.integers(0, 256, (16, 13, 13), dtype=np.uint8)
    before = np.array([abs(np.subtract(*oracle.adv_point(w, weights_blob))) for w in st])
    wins, dev, vals, evals = oracle.adv_search(st, weights_blob, 200, seed=7, scale_iters=60)
    assert evals > 16 * 100 and (dev >= 0).all()
    for w, d, (vr, vg) in zip(wins, dev, vals):
        assert oracle.adv_point(w, weights_blob) == (float(vr), float(vg))
        live = (0.5 < vr < 255.5) or (0.5 < vg < 255.5)      # values the store clamps need no margin: they score 0
        assert abs(d - (abs(vr - vg) if live else 0.0)) < 1e-9
    assert dev.max() >= before.max() * 0.5          # (the magnitude climb may trade deviation for scale on single restarts)


@pytest.mark.skipif(not FIX.exists(), reason="fixture not generated")
def test_worst_windows_found_stay_below_half_the_threshold(weights_blob):
    fx = np.load(FIX)
    delta = shipped_delta(weights_blob)
    assert abs(float(fx["shipped_delta"]) - shipped_delta(weights_blob, 6.0)) < 1e-6      # (the search ran under round 4's factor 6)
    devs = []
    for w, d in zip(fx["shipped_windows"], fx["shipped_dev"]):
        v_ref, v_gpu = oracle.adv_point(w, weights_blob)
        assert abs(abs(v_ref - v_gpu) - d) < 1e-9
        devs.append(abs(v_ref - v_gpu))
    assert max(devs) > 4.4e-4, "the search should at least reach what plain sampling of 54 MPix met"
    assert max(devs) < 0.5 * delta, f"an adversarial window reaches {max(devs) / delta:.2f} delta: raise fixup_delta()'s factor"
    for blob, wins, dv in zip(fx["random_blobs"], fx["random_windows"], fx["random_dev"]):
        d2 = shipped_delta(blob)
        for w, d in zip(wins, dv):
            v_ref, v_gpu = oracle.adv_point(w, blob)
            assert abs(abs(v_ref - v_gpu) - d) < 1e-9 and abs(v_ref - v_gpu) < 0.5 * d2


@pytest.mark.skipif(not FIX.exists(), reason="fixture not generated")
def test_selection_rule_on_a_plane_of_adversarial_windows(weights_blob):
    fx = np.load(FIX)
    plane, cy, cx = tile_windows(fx["shipped_windows"])
    delta = shipped_delta(weights_blob)
    g_out, g_pre = oracle.gpuorder_forward_y(plane, weights_blob)
    r_out, r_pre = oracle.forward_y(plane, weights_blob)
    for k, w in enumerate(fx["shipped_windows"]):
        v_ref, v_gpu = oracle.adv_point(w, weights_blob)
        assert r_pre[cy[k], cx[k]] == np.float32(v_ref) and g_pre[cy[k], cx[k]] == np.float32(v_gpu)
    flagged = (np.abs(g_pre - np.rint(g_pre)) <= delta) & (g_pre > 0.5) & (g_pre < 255.5)
    assert np.array_equal(np.where(flagged, r_out, g_out), r_out)
    assert np.abs(g_pre - r_pre).max() < 0.5 * delta


def test_windows_the_gpu_search_found_reproduce_on_the_cpu_model(weights_blob):
    """Round 5 searched on the GPU itself (tests/checks/adversarial_gpu.py: a plane tiled with 10,000 independent windows per
    launch, 35 million evaluations per kernel) and drove the float32 MFMA kernel to 7.9e-4 -- past the 6.4e-4 of round 4's CPU
    search, and past HALF of round 5's threshold.  The CPU model of that kernel reproduces the deviation bit for bit (the search
    and the model agree about the arithmetic); the bytes stay the reference's as long as the deviation stays below the threshold
    itself, and the largest any search has produced must keep a factor 1.5 below it."""
    fx = np.load(FIX_GPU)
    delta = shipped_delta(weights_blob)
    devs = []
    for w, d in zip(fx["mfma_windows"], fx["mfma_dev"]):
        v_ref, v_gpu = oracle.adv_point(w, weights_blob)
        assert abs(abs(v_ref - v_gpu) - float(d)) < 1e-7
        devs.append(abs(v_ref - v_gpu))
    assert max(devs) > 7.5e-4
    assert max(devs) < delta / 1.5, f"an adversarial window reaches {max(devs) / delta:.2f} delta: raise fixup_delta()'s factor"
    # the split-f16 kernel has no CPU model; its windows are checked on the GPU (tests/test_gpu_refbytes.py).  Against ITS threshold:
    assert float(fx["split16_dev"].max()) < delta * (8.0 / 6.0) / 1.5


FIX_RATIO = Path(__file__).resolve().parent / "golden" / "adversarial_windows_ratio.npz"     # tests/checks/fixup_adversarial_ratio.py (round 6)
FIX_GPU_RATIO = Path(__file__).resolve().parent / "golden" / "adversarial_windows_gpu_ratio.npz"     # tests/checks/adversarial_gpu_ratio.py (round 6)
GAIN = 1.73          # the factor the global delta keeps over the worst deviation any search has produced


def k_needed(w, blob, abs_term):
    """The k this window needs for thr = k * 2^-24 * S1 + abs to stay GAIN x above its deviation (oracle/adversarial.c)."""
    from test_refbytes_model import EPS
    v_ref, v_gpu, s1 = oracle.adv_point_local(w, blob)
    return max(GAIN * abs(v_gpu - v_ref) - abs_term, 0.0) / (EPS * s1), (v_ref, v_gpu, s1)


@pytest.mark.skipif(not FIX_RATIO.exists(), reason="fixture not generated")
def test_every_window_found_keeps_the_per_pixel_threshold_1_73_above_its_deviation(weights_blob):
    """Round 6 flags against a PER-PIXEL threshold k * 2^-24 * S1 + abs (capped at delta), so the attack has to be on what that
    threshold must cover: a window may win by a large deviation or by a small local scale.  The committed windows of that search
    (240,000 restarts over four objectives, shipped model; 6,000 for each of 24 random models) reproduce their recorded figure on
    the one-pixel evaluator, and for EVERY window any search has produced -- this one, the deviation searches of rounds 4-5, the
    climb on the GPU itself -- the library's threshold stays 1.73 x above the deviation: the factor the global delta keeps."""
    from test_refbytes_model import ABS_TERM, K_LOCAL
    fx = np.load(FIX_RATIO)
    assert abs(float(fx["gain"]) - GAIN) < 1e-6 and abs(float(fx["abs_term"]) - ABS_TERM) < 1e-9      # searched under the constants in use
    worst = 0.0
    for w, r, v in zip(fx["shipped_windows"], fx["shipped_kappa"], fx["shipped_vals"]):
        k, vals = k_needed(w, weights_blob, ABS_TERM)
        assert abs(k - float(r)) < 1e-3 * max(1.0, k) and np.allclose(vals, v, rtol=1e-6)
        worst = max(worst, k)
    assert worst > 1.4, "the search should reach what it reported (profiles/r06/fixup_adversarial_ratio.txt: 1.480)"
    pools = [(FIX, "shipped_windows"), (FIX_GPU, "mfma_windows")] + ([(FIX_GPU_RATIO, "mfma_windows")] if FIX_GPU_RATIO.exists() else [])
    # ... and the long run of the same search (900,000 restarts, another seed: 906 M point evaluations, 1.540)
    pools += [(p, "shipped_windows") for p in sorted(FIX_RATIO.parent.glob("adversarial_windows_ratio_*.npz"))]
    for fn, key in pools:
        for w in np.load(fn)[key]:
            worst = max(worst, k_needed(w, weights_blob, ABS_TERM)[0])
    assert worst <= K_LOCAL, f"a window needs k = {worst:.3f}, the library uses {K_LOCAL}: raise kFixLocal"
    for blob, wins, rr in zip(fx["random_blobs"], fx["random_windows"], fx["random_kappa"]):
        for w, r in zip(wins, rr):
            k, _ = k_needed(w, blob, ABS_TERM)
            assert abs(k - float(r)) < 1e-3 * max(1.0, k) and k <= K_LOCAL


@pytest.mark.skipif(not FIX_RATIO.exists(), reason="fixture not generated")
def test_local_selection_rule_on_a_plane_of_ratio_windows(weights_blob):
    from test_refbytes_model import local_threshold
    fx = np.load(FIX_RATIO)
    plane, cy, cx = tile_windows(fx["shipped_windows"])
    delta = shipped_delta(weights_blob)
    thr = local_threshold(plane, weights_blob, delta)
    g_out, g_pre = oracle.gpuorder_forward_y(plane, weights_blob)
    r_out, r_pre = oracle.forward_y(plane, weights_blob)
    flagged = (np.abs(g_pre - np.rint(g_pre)) <= thr) & (g_pre > 0.5) & (g_pre < 255.5)
    assert np.array_equal(np.where(flagged, r_out, g_out), r_out)
    # the centre pixels carry the searched values, and no pixel of the plane -- the windows' neighbourhoods are nearly as hard as
    # the windows -- comes closer to its own threshold than 1 / 1.73 (the global delta is capped in: below it the cap binds)
    for k, v in enumerate(fx["shipped_vals"]):
        assert g_pre[cy[k], cx[k]] == np.float32(v[1]) and r_pre[cy[k], cx[k]] == np.float32(v[0])
    live = (g_pre > 0.5) & (g_pre < 255.5)
    assert (np.abs(g_pre.astype(np.float64) - r_pre) / thr)[live].max() <= 1 / GAIN + 1e-9
