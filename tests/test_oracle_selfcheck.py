"""The checker checks itself (oracle/__init__.py::_forward): on large shared hosts a small plane is computed twice and a
disagreement -- met twice in ~10,000 planes on the GPU boxes, profiles/r05/soak_long.txt -- is settled by a third run."""
from unittest import mock

import numpy as np
import pytest

import oracle
import srcnn_cpp_amd as S


def test_a_disagreeing_run_is_outvoted_and_counted():
    blob = S.load_weights()
    src = np.random.default_rng(5).integers(0, 256, (33, 47), dtype=np.uint8)
    good_u8, good_pre = oracle.forward_y(src, blob)
    real, calls = oracle._forward_once, []

    def flaky(fn, s, ps, b, pw):
        u8, pre = real(fn, s, ps, b, pw)
        calls.append(1)
        if len(calls) == 2:                         # the second run returns a wrong band of rows
            u8 = u8.copy()
            u8[10:14] ^= 0x55
        return u8, pre

    before = oracle.anomalies
    with mock.patch("os.cpu_count", return_value=256), mock.patch.object(oracle, "_forward_once", flaky):
        u8, pre = oracle.forward_y(src, blob)
    assert len(calls) == 3 and oracle.anomalies == before + 1
    assert np.array_equal(u8, good_u8) and np.array_equal(pre, good_pre)
    oracle.anomalies = before                       # (the staged disagreement is not one of the run's: tests/conftest.py reports the counter)


def test_three_different_results_are_an_error():
    blob = S.load_weights()
    src = np.random.default_rng(6).integers(0, 256, (20, 20), dtype=np.uint8)
    real, calls = oracle._forward_once, []

    def broken(fn, s, ps, b, pw):
        u8, pre = real(fn, s, ps, b, pw)
        calls.append(1)
        u8 = u8.copy()
        u8[0, 0] = len(calls)
        return u8, pre

    with mock.patch("os.cpu_count", return_value=256), mock.patch.object(oracle, "_forward_once", broken):
        before = oracle.anomalies
        with pytest.raises(RuntimeError):
            oracle.forward_y(src, blob)
        oracle.anomalies = before


def test_small_hosts_and_large_planes_run_once():
    blob = S.load_weights()
    src = np.random.default_rng(7).integers(0, 256, (16, 16), dtype=np.uint8)
    real, calls = oracle._forward_once, []

    def counting(fn, s, ps, b, pw):
        calls.append(1)
        return real(fn, s, ps, b, pw)

    with mock.patch("os.cpu_count", return_value=8), mock.patch.object(oracle, "_forward_once", counting):
        oracle.forward_y(src, blob)
    assert len(calls) == 1


def test_the_timed_entry_point_runs_the_loops_once_on_any_host():
    """bench.py's cpu_baseline leg times oracle.forward_y_once: ONE run of the loops even where forward_y asks twice (a host with
    more than 32 CPUs) -- the checker's double runs must not halve the reported CPU baseline (they did for a while in round 6)."""
    blob = S.load_weights()
    src = np.random.default_rng(8).integers(0, 256, (16, 16), dtype=np.uint8)
    real, calls = oracle._forward_once, []

    def counting(fn, s, ps, b, pw):
        calls.append(1)
        return real(fn, s, ps, b, pw)

    with mock.patch("os.cpu_count", return_value=256), mock.patch.object(oracle, "_forward_once", counting):
        a = oracle.forward_y_once(src, blob)
        assert len(calls) == 1
        b = oracle.forward_y(src, blob)
        assert len(calls) == 3
    assert np.array_equal(a[0], b[0])
