#!/usr/bin/env python3
"""Adversarial search for the flag threshold of SRCNN_MODE_REFBYTES / REFBYTES16 ON THE GPU ITSELF.

Round 4 searched receptive fields for the largest |v_kernel - v_reference| on the CPU, with a bit-exact model of the float32
MFMA kernel's summation order (tests/checks/fixup_adversarial.py, oracle/adversarial.c).  The split-f16 kernel has no such model
(the f16 MFMA's internal summation order is not documented), so its threshold (8/6 of the float32 mode's) rested on content
statistics alone.  This search needs no model: an output pixel depends on its 13x13 luma window only, so a plane TILED with
G x G independent windows evaluates G*G candidates per launch; the kernel's pre-clamp value at every window centre comes from the
GPU (SRCNN_MODE_MFMA and SRCNN_MODE_SPLIT16 with a pre-clamp plane), the reference arithmetic's from oracle.forward_y on the
same plane.  Every window climbs on its own: a few of its pixels are changed per step and the change is kept where the deviation
grew.  Starts: random noise, flat + noise, saturated patterns, and round 4's worst windows (tests/golden/adversarial_windows.npz).

Afterwards the plane of the worst windows found goes through SRCNN_MODE_REFBYTES and REFBYTES16: the bytes must be the
reference's and the monitored deviation is reported against the thresholds in use.

usage: python tests/checks/adversarial_gpu.py [seconds per mode = 120] [G = 100] [seed = 1] [fresh]
"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402

import oracle  # noqa: E402
import srcnn_cpp_amd as S  # noqa: E402

R = 13
C = R // 2


def tile(wins, g):
    return np.ascontiguousarray(wins.reshape(g, g, R, R).transpose(0, 2, 1, 3).reshape(g * R, g * R))


def centres(plane, g):
    return plane[C::R, C::R].reshape(g * g)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    g = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    fresh = len(sys.argv) > 4 and sys.argv[4] == "fresh"         # do not start from this search's own earlier windows
    rng = np.random.default_rng(seed)
    blob = S.load_weights()
    n = g * g
    ctx = S.Context(0)
    ctx.set_weights_blob(blob)
    deltas = {}

    def starts():
        w = rng.integers(0, 256, (n, R, R), dtype=np.uint8)
        k = n // 4
        base = rng.integers(0, 256, (k, 1, 1))
        w[k:2 * k] = np.clip(base + rng.integers(-6, 7, (k, R, R)), 0, 255).astype(np.uint8)          # flat + noise
        w[2 * k:3 * k] = np.where(rng.random((k, R, R)) < rng.random((k, 1, 1)), 255, 0).astype(np.uint8)   # saturated patterns
        gold = np.load(ROOT / "tests" / "golden" / "adversarial_windows.npz")["shipped_windows"]
        if fresh:
            gold = gold[:0]
        w[3 * k:3 * k + len(gold)] = gold                                                                # round 4's worst
        mine = ROOT / "tests" / "golden" / "adversarial_windows_gpu.npz"                                 # ... and this search's own
        if mine.exists() and not fresh:
            prev = np.load(mine)
            prev = np.concatenate([prev["mfma_windows"], prev["split16_windows"]])
            reps = np.repeat(prev, 40, 0)                                                                # 40 climbers start from each
            w[3 * k + len(gold):3 * k + len(gold) + len(reps)] = reps[: n - 3 * k - len(gold)]
        return w

    def evaluate(mode, wins):
        plane = tile(wins, g)
        pre = np.empty(plane.shape, np.float32)
        ctx.set_mode(mode)
        ctx.forward_y(plane, preclamp=pre)
        _, ref = oracle.forward_y(plane, blob)
        return np.abs(centres(pre, g).astype(np.float64) - centres(ref, g)), centres(ref, g)

    results = {}
    for name, mode in (("mfma", S.MODE_MFMA), ("split16", S.MODE_SPLIT16)):
        wins = starts()
        dev, _ = evaluate(mode, wins)
        first = dev.max()
        t0, steps, evals = time.time(), 0, n
        while time.time() - t0 < secs:
            cand = wins.copy()
            k = int(rng.integers(1, 5))                                  # pixels changed per window this step
            idx = rng.integers(0, R * R, (n, k))
            kind = rng.random((n, 1))
            val = np.where(kind < 0.25, rng.integers(0, 256, (n, k)),
                           np.where(kind < 0.75, np.take_along_axis(cand.reshape(n, -1), idx, 1).astype(int) + rng.integers(-3, 4, (n, k)),
                                    rng.choice([0, 255], (n, k))))
            flat = cand.reshape(n, -1)
            np.put_along_axis(flat, idx, np.clip(val, 0, 255).astype(np.uint8), 1)
            d2, _ = evaluate(mode, cand)
            better = d2 > dev
            wins[better] = cand[better]
            dev[better] = d2[better]
            steps += 1
            evals += n
            if steps % 50 == 0:                                          # the weakest tenth restarts from random members of the best tenth
                order = np.argsort(dev)
                weak, strong = order[: n // 10], rng.choice(order[-(n // 10):], n // 10)
                wins[weak] = wins[strong]
                dev[weak] = dev[strong]
            if steps % 200 == 0:
                print(f"  {name}: step {steps}, {evals} evaluations: largest {dev.max():.3e}, 100th largest {np.sort(dev)[-100]:.3e}, "
                      f"distinct windows among the best 100: {len(np.unique(wins[np.argsort(dev)[-100:]].reshape(100, -1), axis=0))}", flush=True)
        best = np.argsort(dev)[-64:]
        results[name] = (wins[best].copy(), dev[best].copy())
        print(f"{name}: {evals} window evaluations in {time.time() - t0:.0f} s ({steps} steps of {n} climbers); largest |v_kernel - v_ref| "
              f"at the start {first:.3e}, found {dev.max():.3e}; 64th largest {np.sort(dev)[-64]:.3e}", flush=True)

    # the worst windows of both searches through the byte-exact modes
    worst = np.concatenate([results["mfma"][0], results["split16"][0]])
    gg = int(np.ceil(np.sqrt(len(worst))))
    pad = np.concatenate([worst, np.repeat(worst[:1], gg * gg - len(worst), 0)])
    plane = tile(pad, gg)
    r_out, _ = oracle.forward_y(plane, blob)
    for name, mode in (("REFBYTES", S.MODE_REFBYTES), ("REFBYTES16", S.MODE_REFBYTES16)):
        with S.Context(0) as c2:                                         # (the counters are per context and accumulate)
            c2.set_weights_blob(blob)
            c2.set_mode(mode)
            out = c2.forward_y(plane)
            st = c2.fixup_stats()
        deltas[name] = st["delta"]
        ok = bool(np.array_equal(out, r_out))
        print(f"{name} on the plane of the {len(worst)} worst windows: bytes equal the reference arithmetic's: {ok}; "
              f"monitored deviation {st['max_dev']:.3e} = {st['max_dev'] / st['delta']:.3f} of the threshold {st['delta']:.3e}; "
              f"launches redone by the net: {st['exact_reruns']}", flush=True)
        assert ok
    for name in ("mfma", "split16"):
        d = results[name][1].max()
        print(f"{name}: worst found {d:.3e} = {d / deltas['REFBYTES']:.3f} of REFBYTES' threshold {deltas['REFBYTES']:.3e}, "
              f"{d / deltas['REFBYTES16']:.3f} of REFBYTES16's {deltas['REFBYTES16']:.3e}")
    np.savez_compressed(ROOT / "gpurun_out" / "adversarial_gpu_windows.npz", mfma_windows=results["mfma"][0], mfma_dev=results["mfma"][1],
                        split16_windows=results["split16"][0], split16_dev=results["split16"][1])
    ctx.close()


if __name__ == "__main__":
    main()
