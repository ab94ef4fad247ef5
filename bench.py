#!/usr/bin/env python3
"""bench.py -- SRCNN Y-channel MPix/s on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (fused Convolution99x11 + Convolution55,
src/srcnn.cpp:609+627) over one synthetic 3840x2160 luma plane per GPU
(BASELINE.json configs[1]: 1920x1080 x2.0), input already resident in HBM.
N GPUs = N ranks, one process per GPU, one frame per rank per step, no
collective on the data path (frames are independent: weak scaling).
`--workload stripe` row-stripes ONE plane over the ranks instead (configs[3]).

Launching.  `python bench.py --gpus N` starts its own N ranks: the parent --
which never touches the GPU -- spawns N fresh child processes with
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT set and forwards rank
0's JSON line.  Under `python -m torch.distributed.run --nproc-per-node N
bench.py --gpus N` (RANK already in the environment) every process is a rank.

Prints ONE JSON line on rank 0, including
  roofline      dominant kernel vs the f32 MFMA peak (HIP events per launch)
  cpu_baseline  the oracle (port of the reference loops) on the host cores, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time
import zlib
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6     # dense f16/bf16: v_mfma_f32_32x32x16_f16 = 32,768 FLOP / 32 clk / SIMD x 1024 SIMD x 2.4 GHz
PEAK_HBM_GBPS = 8000.0            # MI355X_MICROARCH.md: HBM3E ~8 TB/s


# --------------------------------------------------------------------------- CPU baseline leg

def host_cpu_info():
    """(model name, logical CPUs this process may run on, physical cores among them)."""
    allowed = os.sched_getaffinity(0)
    model, cores, cur = "unknown", set(), {}
    try:
        for line in Path("/proc/cpuinfo").read_text().splitlines() + [""]:
            if not line.strip():
                if cur.get("processor") is not None and int(cur["processor"]) in allowed:
                    cores.add((cur.get("physical id", "0"), cur.get("core id", cur["processor"])))
                cur = {}
                continue
            k, _, v = line.partition(":")
            cur[k.strip()] = v.strip()
            if k.strip() == "model name":
                model = v.strip()
    except OSError:
        pass
    return model, len(allowed), (len(cores) or len(allowed))


def _time_oracle(forward, frame, width, height, target_s, threads):
    probe_rows = min(height, max(threads, 16))
    forward(frame[:probe_rows])                                # warm-up: page in, spin up the OpenMP team
    t = time.perf_counter()
    forward(frame[:probe_rows])
    dt = time.perf_counter() - t
    rows = int(min(height, max(probe_rows, probe_rows * target_s / max(dt, 1e-6))))
    reps, pix, t = 0, 0, time.perf_counter()
    while True:                                                # whole slabs until ~target_s of CPU work
        forward(frame[:rows])
        reps += 1
        pix += width * rows
        dt = time.perf_counter() - t
        if dt >= target_s or reps >= 64:
            break
    return pix / dt / 1e6, reps, rows, dt


def cpu_baseline_child(width, height, target_s=12.0):
    """Runs in its own process (OMP_* set by the parent before libgomp loads): time the oracle
    (reference loops, strict IEEE, OpenMP) on a bounded slab of the same workload; rows are
    independent, so a slab of full-width rows has the per-pixel cost of the whole frame."""
    import ctypes
    import numpy as np
    import oracle
    import srcnn_cpp_amd as S
    from srcnn_cpp_amd.synth import synth_luma

    model, logical, physical = host_cpu_info()
    threads = int(os.environ.get("OMP_NUM_THREADS", physical))
    blob = S.load_weights()
    frame = synth_luma(width, height)
    last = {}

    def fwd(f):
        # (ONE run per call: oracle.forward_y asks the loops twice on large hosts -- the checker guarding itself, not the reference's cost)
        last["rows"], last["out"] = f.shape[0], oracle.forward_y_once(f, blob)[0]
    v, reps, rows, dt = _time_oracle(fwd, frame, width, height, target_s, threads)
    # what the reference arithmetic makes of the bench's frame: sha256 of the last slab computed (the whole plane when the slab
    # is the whole plane) -- bench.py checks the SRCNN_MODE_REFBYTES output of the GPU against it
    import hashlib
    ref_sha = {"rows": int(last["rows"]), "sha256": hashlib.sha256(np.ascontiguousarray(last["out"]).tobytes()).hexdigest()}
    out = {"value": round(v, 4), "unit": "MPix/s", "cores": threads, "kind": "port", "reference_output": ref_sha,
           "threads": threads, "physical_cores": physical, "logical_cpus": logical, "cpu_model": model,
           "binding": f"OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND', 'unset')} "
                      f"OMP_PLACES={os.environ.get('OMP_PLACES', 'unset')}",
           "sample": f"{reps} x top {rows} of {height} rows of the {width}x{height} frame, "
                     f"oracle/srcnn_oracle.c (reference loops, -O3 -ffp-contract=off), OpenMP {threads} threads "
                     f"(one per physical core), {dt:.1f} s"}
    # The reference ships its objects built WITHOUT -O (Makefile:21-23,43): the same loops at -O0, bounded to ~5 s.
    try:
        o0 = oracle.build_o0()
        lib0 = ctypes.CDLL(str(o0))
        fwd0 = oracle.bind_forward(lib0)
        v0, reps0, rows0, dt0 = _time_oracle(lambda f: fwd0(f, blob), frame, width, height, 5.0, threads)
        out["as_shipped_flags"] = {
            "value": round(v0, 4), "unit": "MPix/s", "kind": "port at -O0 (the reference's effective flags)",
            "sample": f"{reps0} x top {rows0} rows, {dt0:.1f} s, same threads",
            "survey_figure": "0.10 MPix/s: the real reference source at shipped flags, 576x576, 8 vCPU Xeon 2.1 GHz "
                             "survey container (BASELINE.md section 2) -- quoted, not measured here"}
    except Exception as e:  # the -O0 leg is informative only
        out["as_shipped_flags"] = {"value": None, "error": str(e)[:200]}
    print(json.dumps(out), flush=True)


def cpu_baseline(width, height):
    """Spawn the CPU-baseline child with one OpenMP thread per physical core, bound."""
    _, _, physical = host_cpu_info()
    env = dict(os.environ, OMP_NUM_THREADS=str(physical), OMP_PROC_BIND="close", OMP_PLACES="cores")
    r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--cpu-baseline-only",
                        "--width", str(width), "--height", str(height)],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if r.returncode != 0:
        return {"value": None, "error": r.stderr[-400:]}
    return json.loads(r.stdout.strip().splitlines()[-1])


# --------------------------------------------------------------------------- self-launch

def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n: int, cmd=None) -> int:
    """Parent of a multi-rank run: start n fresh rank processes (this process has made no GPU call and makes
    none), forward rank 0's stdout, return the worst exit code.  WATCHDOG: every child is polled; as soon as one
    exits non-zero (or this process is told to stop) the others -- exactly the processes started here -- are
    killed and the run returns non-zero within seconds, instead of rank 0 waiting in a rendezvous or a collective
    for a peer that no longer exists until some outer time limit decides."""
    import signal
    import threading

    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen(cmd or [sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    reader.start()

    def kill_all(*_):
        for p in procs:
            if p.poll() is None:
                p.kill()                                       # exactly the children this process started
    old_handlers = {sig: signal.signal(sig, lambda *_: (kill_all(), sys.exit(128 + 15))) for sig in (signal.SIGTERM, signal.SIGINT)}
    failed = None
    while any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            rc = p.poll()
            if rc not in (None, 0) and failed is None:
                failed = (r, rc)
                print(f"[bench.py launcher] rank {r} exited with code {rc}: stopping the other ranks", file=sys.stderr, flush=True)
                kill_all()
        time.sleep(0.05)
    reader.join(timeout=5)
    for sig, h in old_handlers.items():
        signal.signal(sig, h)
    if failed is None:                               # (every rank may have exited between two polls)
        bad = [(r, p.returncode) for r, p in enumerate(procs) if p.returncode != 0]
        failed = bad[0] if bad else None
    if failed is None:
        sys.stdout.write((out0[0] if out0 else b"").decode())
        sys.stdout.flush()
        return 0
    return failed[1] if 0 < failed[1] < 256 else 1


# --------------------------------------------------------------------------- the row-striped plane riding on the frames line

# One 7680x4320 plane on ONE MI355X, fused float32 MFMA path, resident in HBM: the denominator of the stripe leg's speed-up.
# profiles/r04/measurements.jsonl (3.744 ms), re-measured in round 5 on three boxes (3.737-3.752 ms: profiles/r05/).
N1_PLANE_MS = {"width": 7680, "height": 4320, "ms": 3.744, "source": "profiles/r04/measurements.jsonl, profiles/r05/fix_apply_ab*.txt (3.737-3.758 on five boxes)"}


def stripe_leg(args, S, torch, dist, ctx, world, rank, timed, make_rccl):
    """BASELINE configs[3] at N = world: the 7680x4320 plane row-striped over the ranks, every rank ONE launch per step on its own
    rows (srcnn_forward_y_rows_halo_dev), in the transports that exist on this node -- `halo`: the 6 boundary rows travel by RCCL
    send/recv every step (host-staged over gloo with --backend gloo: a smoke configuration, marked degraded); `peer`: no per-step
    exchange, the neighbours' stripes are mapped once through HIP IPC and the kernel loads their edge rows where they lie (xGMI
    between GPUs).  Per form: ms per image (first rank's start -> last rank's end over K steps), per-rank ms, speed-up against
    the committed one-GPU time, and the sha256 of the stitched plane against tests/golden/config_checksums.json.  Collective:
    every rank runs it; rank 0 returns the object."""
    import hashlib
    import numpy as np
    from srcnn_cpp_amd import sharding
    from srcnn_cpp_amd.synth import synth_luma

    SW, SH = N1_PLANE_MS["width"], N1_PLANE_MS["height"]
    if min(b - a for a, b in (sharding.stripe_rows(SH, world, k) for k in range(world))) < sharding.HALO_ROWS:
        return {"skipped": f"{world} ranks leave stripes thinner than the 6-row halo"} if rank == 0 else None
    try:
        pin = json.loads((ROOT / "tests" / "golden" / "config_checksums.json").read_text())["c3_7680x4320"]["gpuorder_sha256"][0]
    except Exception:                      # noqa: BLE001 -- the checksum file travels with the tree; say so if it does not
        pin = None
    r0, r1 = sharding.stripe_rows(SH, world, rank)
    rows_np = synth_luma(SW, SH, rows=(r0, r1))
    d_s = torch.from_numpy(rows_np).cuda()
    d_o = torch.zeros_like(d_s)
    rows = [sharding.stripe_rows(SH, world, k) for k in range(world)]
    pad = max(b - a for a, b in rows)

    def measure(step_fn):
        for _ in range(32):                # the same COUNT on every rank (they exchange every step), untimed
            step_fn()
        torch.cuda.synchronize()
        t_a, t_b, _, kern = timed(args.steps, step_fn)
        t = torch.tensor([t_a, t_b, kern], dtype=torch.float64)
        all_t = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(all_t, t)
        buf = torch.zeros((pad, SW), dtype=torch.uint8)
        buf[: r1 - r0] = d_o.cpu()
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        if rank != 0:
            return None
        plane = torch.cat([p[: b - a] for p, (a, b) in zip(parts, rows)], dim=0).numpy()
        sha = hashlib.sha256(np.ascontiguousarray(plane).tobytes()).hexdigest()
        ms = float(max(x[1] for x in all_t) - min(x[0] for x in all_t)) / args.steps * 1e3
        per_rank = [float(x[1] - x[0]) / args.steps * 1e3 for x in all_t]
        return {"ms_per_image": round(ms, 4), "value": round(SW * SH / ms / 1e3, 2), "unit": "MPix/s",
                "per_rank_ms": [round(v, 4) for v in per_rank], "per_rank_kernel_ms": [round(float(x[2]), 4) for x in all_t],
                "per_rank_frac_of_f32_mfma_peak": [round(S.FLOP_PER_PIXEL * SW * (b - a) / (float(x[2]) * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
                                                   for x, (a, b) in zip(all_t, rows)],
                "speedup_vs_one_gpu": round(N1_PLANE_MS["ms"] / ms, 3), "output_sha256": sha,
                "sha256_equals_golden": (sha == pin) if pin else None}

    def agree(ok, why=""):
        """every rank says whether its part of a set-up worked; all go on, or none does"""
        said = [None] * world
        dist.all_gather_object(said, None if ok else f"rank {rank}: {why}")
        bad = [s for s in said if s]
        return (not bad), "; ".join(bad)

    out = {"workload": f"ONE {SW}x{SH} luma plane (BASELINE configs[3]) row-striped over {world} GPU(s), one launch per rank per step, "
                       f"float32 MFMA path, stripes resident in HBM; {args.steps} timed steps behind 32 untimed ones",
           "one_gpu_reference": N1_PLANE_MS, "golden_sha256": pin, "forms": {}}
    degraded = []
    # ---- halo: a 6-row exchange per step ----
    rccl, rccl_world, why = None, None, ""
    if args.backend == "nccl":
        try:
            rccl, rccl_world = make_rccl()
        except Exception as e:             # noqa: BLE001
            why = str(e)[:200]
        ok, said = agree(rccl is not None, why)
        if not ok:
            rccl, rccl_world, why = None, None, said
    try:
        stepper = sharding.StripeStep(d_s, d_o, SH, world, rank, sharding.gpu_launch_rows(ctx), group=rccl, overlap=True,
                                      via_host=rccl is None, launch_rows_halo=sharding.gpu_launch_rows_halo(ctx))
        res = measure(stepper.step)
        if rank == 0:
            transport = "rccl send/recv" if rccl is not None else "host-staged (gloo)"
            res.update({"halo_transport": transport, "rccl_world": rccl_world})
            if rccl is None:
                res["rccl_error"] = why or "--backend gloo"
                degraded.append("halo: the 6-row exchange did not run over RCCL")
            elif rccl_world != world:
                degraded.append(f"halo: RCCL summed over {rccl_world} ranks")
            if res["sha256_equals_golden"] is False:
                degraded.append("halo: stitched plane differs from the golden checksum")
            out["forms"]["halo"] = res
    except Exception as e:                 # noqa: BLE001 -- a failed form must not take the frames line with it
        if rank == 0:
            out["forms"]["halo"] = {"error": str(e)[:300]}
            degraded.append("halo: failed")
    # ---- peer: the neighbours' stripes mapped once (HIP IPC), no exchange per step ----
    d_o.zero_()
    torch.cuda.synchronize()
    peer = None
    try:
        peer = sharding.PeerStripeStep(ctx, rows_np, d_o, SH, world, rank, group=None)
        err = ""
    except Exception as e:                 # noqa: BLE001 (collective inside: every rank raises, or none)
        err = str(e)[:300]
    if peer is not None:
        res = measure(peer.step)
        peer.close()
        if rank == 0:
            res["halo_transport"] = ("none per step: the neighbours' stripes are mapped once (HIP IPC), the kernel loads their 6 edge "
                                     "rows where they lie" + (" -- ALL RANKS ON ONE GPU (--shared-gpu): no xGMI link crossed" if args.shared_gpu else " (xGMI between GPUs)"))
            if res["sha256_equals_golden"] is False:
                degraded.append("peer: stitched plane differs from the golden checksum")
            out["forms"]["peer"] = res
    elif rank == 0:
        out["forms"]["peer"] = {"error": err}
        degraded.append("peer: HIP IPC mapping unavailable")
    if args.shared_gpu:
        degraded.append("--shared-gpu: the ranks share one GPU, no link was crossed and the kernels of the ranks ran one after another")
    if rank != 0:
        return None
    out["degraded"] = bool(degraded)
    out["degraded_why"] = degraded
    return out


# --------------------------------------------------------------------------- one rank

def emit_line(out):
    """The ONE JSON line, as the LAST line of stdout: native libraries (RCCL prints its path) write through C stdio, whose
    buffer for a pipe is flushed at exit -- behind Python's own write -- unless it is flushed first."""
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.write(json.dumps(out) + "\n")
    sys.stdout.flush()
    os.close(1)                                     # anything a library still prints at exit goes nowhere, not behind the line
    os.open(os.devnull, os.O_WRONLY)



def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=1, help="frames per GPU per step")
    ap.add_argument("--path", choices=["fused", "unfused", "host", "pipeline", "surface", "surface-dev"], default="fused",
                    help="fused: one kernel, u8 in/out (default); unfused: layer-1/2 kernel -> 32 f32 planes in "
                         "HBM -> layer-3 kernel; host: srcnn_forward_y on host buffers (PCIe-inclusive); "
                         "surface: the reference call surface on host buffers -- srcnn_conv99x11 then srcnn_conv55 "
                         "(what include/srcnn_amd.hpp's Convolution99x11 / Convolution55 call), 32 f32 planes over PCIe; "
                         "surface-dev: the same two calls with the 32 planes kept in device memory between them "
                         "(srcnn_conv99x11_to_dev + srcnn_conv55_from_dev, the DevicePlane<float> overloads): only the u8 planes cross PCIe")
    ap.add_argument("--mode", choices=["mfma", "exact", "split16", "refbytes", "refbytes16"], default="mfma",
                    help="mfma: float32 MFMA (default, the headline); exact: reference arithmetic on the vector ALU; "
                         "refbytes: the float32 MFMA kernel + exact recomputation of the ~0.15 %% of pixels whose value lies next to a "
                         "truncation boundary -- the reference's bytes (SRCNN_MODE_REFBYTES); "
                         "split16: opt-in f16-MFMA mode with (hi, lo) operand splitting (SURVEY.md 8f rank 4) -- "
                         "never the headline number")
    ap.add_argument("--workload", choices=["frames", "stripe"], default="frames",
                    help="frames: independent planes per rank, no collective, weak scaling (default); "
                         "stripe: ONE width x height plane row-striped over the ranks with a 6-row "
                         "point-to-point halo exchange per step, strong scaling (BASELINE configs[3])")
    ap.add_argument("--stripe-form", choices=["halo", "peer", "bands", "assemble"], default="halo",
                    help="stripe workload, how a rank's step is launched: halo (default) = ONE launch on the stripe where it "
                         "lies, the received 6-row halos in small buffers of their own that alternate from step to step, so the "
                         "exchange of the next step overlaps this step's kernel (srcnn_forward_y_rows_halo_dev); bands = interior "
                         "rows first, then the two 6-row edge bands behind the exchange (rounds 2-3); assemble = exchange into a "
                         "[halo | stripe | halo] buffer, then one launch; peer = NO per-step exchange: every rank maps its neighbours' "
                         "stripes once (HIP IPC) and its one launch reads their 6 edge rows where they lie, over xGMI "
                         "(sharding.PeerStripeStep)")
    ap.add_argument("--no-overlap", action="store_true", help="stripe workload: the same as --stripe-form assemble")
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed device wake-up BEFORE the W warm-up steps: the same step, repeated for this long.  An idle "
                         "MI355X needs ~20 launches (20-30 ms) of load to reach its steady clocks -- the first steps after "
                         "idle run 4-10 %% slower (profiles/r02/clock_ramp.txt) -- and the contract's W = 5 steps are 5 ms.  "
                         "0 disables it.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-e2e", action="store_true", help="skip the PCIe-inclusive secondary figure (`e2e`)")
    ap.add_argument("--sustained-s", type=float, default=6.0,
                    help="N = 1, fused path: after the timed region, keep running the same step for this many seconds and report the "
                         "rate over it (`sustained`): the K timed steps are 20-50 ms of load, this is what the device holds for "
                         "seconds (clocks, temperature).  0 skips it; so does --no-cpu-baseline (a quick run).")
    ap.add_argument("--seam-deferral", choices=["on", "off"], default="on",
                    help="fused float32 path, one plane per step: on (default) = srcnn_set_seam_deferral(1), the K steps are queued back to "
                         "back and the seam blocks of step k ride behind the work items of step k + 1 instead of a launch of their own; "
                         "the last step's are queued (srcnn_flush) INSIDE the timed region, before its closing event.  off = one seam "
                         "launch per step (rounds 1-4).  Same bytes.")
    ap.add_argument("--no-stripe-leg", action="store_true",
                    help="N > 1, frames workload: skip the row-striped 7680x4320 plane (configs[3], ms per image) that rides on the line as `stripe`")
    ap.add_argument("--stripe-timeout-s", type=float, default=240.0,
                    help="N > 1: seconds the row-striped leg behind the frames figure may take before its watchdog prints the line without it (0 = no watchdog)")
    ap.add_argument("--no-lanes", action="store_true", help="skip the two-lane figure (`two_lanes`: the planes of a stream alternately on two "
                                                            "contexts of the one GPU)")
    ap.add_argument("--no-refbytes", action="store_true", help="skip the SRCNN_MODE_REFBYTES figure and its check against the oracle's bytes")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="transport of the stripe halo exchange (nccl == RCCL over xGMI; gloo stages through host "
                         "memory).  Barriers and the max-over-ranks always use a gloo group.")
    ap.add_argument("--halo-fallback", choices=["fail", "host"], default="fail",
                    help="stripe workload with --backend nccl when RCCL cannot start: fail = exit non-zero (default: a number "
                         "measured on the host-staged path is not the configs[3] number); host = stage the halo rows through "
                         "host memory and mark the line \"degraded\": true")
    ap.add_argument("--host", choices=["py", "cxx"], default="py",
                    help="py: one process per GPU, torch.distributed as plumbing (default, what the driver launches); "
                         "cxx: ONE process drives --gpus contexts through the C ABI's several-GPUs entry points "
                         "(srcnn_forward_y_striped_dev: hipMemcpyPeerAsync halo copies, persistent host threads; frames: one "
                         "context per GPU, launches queued from one host thread) -- the second transport of the scaling curve")
    ap.add_argument("--fix-margin", type=float, default=None,
                    help="REFBYTES modes: factor of the flag threshold's weight-proportional term (srcnn_set_fixup_margin; library default 4)")
    ap.add_argument("--no-fix-strict", action="store_true",
                    help="REFBYTES modes: drop the device-side safety net (srcnn_set_fixup_strict(0): no fix_rerun_kernel launch)")
    ap.add_argument("--lib", default=None, help=argparse.SUPPRESS)        # another build of the library (tools/ab.sh, the tuning build)
    ap.add_argument("--plpad", type=int, default=0, help=argparse.SUPPRESS)   # experiment: floats added to the unfused workspace's plane pitch (tuning build)
    ap.add_argument("--fault-rank", type=int, default=-1, help=argparse.SUPPRESS)      # test hook: this rank exits 7 before the rendezvous
    ap.add_argument("--hang-stripe-rank", type=int, default=-1, help=argparse.SUPPRESS)    # test hook: this rank never enters the stripe leg (its watchdog's test)
    ap.add_argument("--shared-gpu", action="store_true",
                    help="smoke-test aid: every rank uses GPU 0 (1-GPU box, use with --backend gloo)")
    return ap.parse_args()


def worker(args):
    if int(os.environ.get("RANK", "0")) == args.fault_rank:    # test hook of the launcher's watchdog: dies before the rendezvous
        sys.exit(7)
    # the host driver of this pool supports dmabuf IPC only: RCCL and HIP-IPC mappings between the ranks need this before the
    # first GPU call of the process -- also under torch.distributed.run, where bench.py's own launcher has not set it
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import srcnn_cpp_amd as S
    if getattr(args, "lib", None):
        S.use_library(args.lib)
    from srcnn_cpp_amd.synth import synth_batch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    rccl = None            # process group for device-to-device halo rows; None: stage through host memory
    rccl_world = None
    def make_rccl():
        """An RCCL group over the ranks, proven by an all-reduce RCCL itself sums: (group, world it summed over); raises."""
        import datetime
        g = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=120))
        one = torch.ones(1, device="cuda")
        dist.all_reduce(one, group=g)
        torch.cuda.synchronize()
        n = int(one.item())                                    # what RCCL itself summed over: must be `world`
        if n != world:
            raise RuntimeError(f"RCCL reduced over {n} ranks, expected {world}")
        return g, n

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Control plane (barrier, max over ranks, checksum gather): gloo, host tensors.  The data path has
        # no collective for frames; for stripes the 6 halo rows go neighbour to neighbour over RCCL.
        dist.init_process_group("gloo")
        degraded = False
        peer_form = args.workload == "stripe" and args.stripe_form == "peer" and not args.no_overlap
        # (frames exchange nothing and the peer form maps the neighbours' stripes once: no RCCL communicator is built for them)
        if args.backend == "nccl" and args.workload == "stripe" and not peer_form:
            try:
                rccl, rccl_world = make_rccl()
            except Exception as e:
                # A scaling number measured with the halo rows staged through host memory is NOT the configs[3] number:
                # fail loudly (the launcher's watchdog stops the other ranks) unless the caller asked for the fallback,
                # and then say so in the line.
                if args.halo_fallback != "host":
                    print(f"[rank {rank}] RCCL group unavailable ({e}); --backend nccl was asked for: giving up "
                          f"(--halo-fallback host stages the halo rows through host memory instead)", file=sys.stderr, flush=True)
                    sys.exit(3)
                print(f"[rank {rank}] RCCL group unavailable ({e}); halo rows staged through host memory (degraded)",
                      file=sys.stderr, flush=True)
                rccl, rccl_world, degraded = None, None, True

    W, H, F = args.width, args.height, args.frames
    ctx = S.Context(local_rank)
    ctx.set_weights_blob(S.load_weights())
    if args.mode == "exact":
        ctx.set_mode(S.MODE_EXACT)
    elif args.mode == "split16":
        ctx.set_mode(S.MODE_SPLIT16)
    elif args.mode == "refbytes":
        ctx.set_mode(S.MODE_REFBYTES)
    elif args.mode == "refbytes16":                                # opt-in: split-f16 kernel + exact fix-up
        ctx.set_mode(S.MODE_REFBYTES16)
    deferral = args.seam_deferral == "on" and args.mode == "mfma" and args.path == "fused"
    if deferral:
        ctx.set_seam_deferral(True)
    if args.fix_margin is not None:
        ctx.set_fixup_margin(args.fix_margin)
    if args.no_fix_strict:
        ctx.set_fixup_strict(False)
    # a real (non-null) stream that both torch's events and the HIP kernels use
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)

    # each rank owns its frames (frame-sharded stream; no data-path collective)
    stripe = args.workload == "stripe"
    r0, r1 = 0, H
    if stripe:
        from srcnn_cpp_amd import sharding
        from srcnn_cpp_amd.synth import synth_luma
        if args.path != "fused" or F != 1:
            raise SystemExit("--workload stripe runs the fused path on one plane")
        r0, r1 = sharding.stripe_rows(H, world, rank)
        frames = synth_luma(W, H)[None, r0:r1].copy()          # this rank's rows of the plane
    else:
        frames = synth_batch(W, H, F, first_frame=rank * F)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    plpad = args.plpad                                               # experiment knob: see srcnn_forward_y_unfused_dev
    d_work = torch.empty((F * 32 * (H * W + plpad),), dtype=torch.float32, device="cuda") if args.path == "unfused" else None

    stepper = None
    if stripe:      # band buffers, send views and the point-to-point op list are built ONCE (sharding.StripeStep)
        form = "assemble" if args.no_overlap else args.stripe_form
        if args.mode not in ("mfma", "refbytes") and form in ("halo", "peer"):
            form = "bands"              # the split-f16 kernels read one buffer only
        if form == "peer":
            stepper = sharding.PeerStripeStep(ctx, frames[0], d_out[0], H, world, rank, group=None)     # control plane: the gloo group
        else:
            stepper = sharding.StripeStep(d_in[0], d_out[0], H, world, rank, sharding.gpu_launch_rows(ctx), group=rccl,
                                          overlap=form != "assemble", via_host=world > 1 and rccl is None,
                                          launch_rows_halo=sharding.gpu_launch_rows_halo(ctx) if form == "halo" else None)
    host_out = np.empty_like(frames[0])
    host_frames = np.empty_like(frames) if args.path == "host" and F > 1 else None
    if args.path == "pipeline":
        if W % 2 or H % 2 or stripe:
            raise SystemExit("--path pipeline upsamples x2: width and height must be even")
        lo = np.stack([frames[:, ::2, ::2]] * 3, axis=-1).copy()          # [F, H/2, W/2, 3] B,G,R
        d_lo = torch.from_numpy(lo).cuda()
        d_hi = torch.zeros((F, H, W, 3), dtype=torch.uint8, device="cuda")
    if args.path in ("surface", "surface-dev"):
        if F != 1 or stripe:
            raise SystemExit("--path surface runs one plane")
        w1, b1, w2, b2, w3, b3 = S.split_weights(S.load_weights())
        if args.path == "surface":
            host_planes = [np.empty((H, W), np.float32) for _ in range(32)]   # the reference's vector<Mat> (:602-607)
        else:
            d_planes = ctx.dev_alloc(32 * H * W * 4)                          # vector<DevicePlane<float>>

    def step():
        if stripe:
            stepper.step()
        elif args.path == "pipeline":
            for k in range(F):
                ctx.process_bgr_dev(d_lo[k].data_ptr(), 3 * (W // 2), W // 2, H // 2, 2.0, d_hi[k].data_ptr(), 3 * W)
        elif args.path == "host":
            if F == 1:
                ctx.forward_y(frames[0], dst=host_out)
            else:                                             # stream of host frames, transfers overlapped
                ctx.forward_y_frames(frames, out=host_frames)
        elif args.path == "surface":                          # src/srcnn.cpp:609 and :627 through the host surface
            ctx.conv99x11(frames[0], host_planes, w1, b1, w2, b2)
            ctx.conv55(host_planes, host_out, w3, b3)
        elif args.path == "surface-dev":                      # the same two call sites, the map stays on the device
            ctx.conv99x11_to_dev(frames[0], d_planes, W, H * W, w1, b1, w2, b2)
            ctx.conv55_from_dev(d_planes, W, H * W, host_out, w3, b3)
        elif args.path == "fused":
            ctx.forward_y_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, F)
        else:
            ctx.forward_y_unfused_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, F,
                                      d_work.data_ptr())

    default_step = step

    def barrier():
        if dist is not None:
            dist.barrier()

    host_us = [0.0]

    def timed(k_steps, step=None):
        """K steps between two HIP events on the kernels' stream; returns (this rank's start, end on the node's
        monotonic clock, kernel ms per step).  The end is taken BEFORE the closing barrier."""
        step = step or default_step
        ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        if dist is not None:
            # ... and a common start: ranks leave a gloo barrier 50-200 us apart, 0.5-1 % of the driver's 20-step window.  The
            # ranks of a node share one monotonic clock, so rank 0 names an instant 2 ms ahead and everybody spins up to it.
            go = torch.tensor([time.perf_counter() + 0.002], dtype=torch.float64)
            dist.broadcast(go, src=0)
            while time.perf_counter() < float(go.item()):
                pass
        t_a = time.perf_counter()
        ev_a.record(stream)
        for _ in range(k_steps):
            step()
        if deferral:
            ctx.flush()                # the last step's deferred seam launch: queued inside the timed region
        host_us[0] = (time.perf_counter() - t_a) / max(1, k_steps) * 1e6     # everything queued: the host's share of a step
        ev_b.record(stream)
        torch.cuda.synchronize()
        t_b = time.perf_counter()
        barrier()
        torch.cuda.synchronize()
        return t_a, t_b, time.perf_counter(), ev_a.elapsed_time(ev_b) / max(1, k_steps)

    # First, the contract exactly as written: W warm-up steps from wherever the device's clocks are, K timed steps.
    # On an idle MI355X those 25 ms sit inside the clock ramp (profiles/r02/clock_ramp.txt); reported as `cold_start`.
    cold = None
    if args.prewarm_ms > 0:
        for _ in range(args.warmup):
            step()
        c_a, c_b, _, c_kern = timed(args.steps)
        cold = {"ms_per_step": round((c_b - c_a) / args.steps * 1e3, 4), "kernel_ms": round(c_kern, 4)}
    # untimed: bring the device from idle to its steady clocks, then the W warm-up steps of the contract
    prewarm_steps = 0
    if args.prewarm_ms > 0 and stripe and world > 1:
        for _ in range(64):                 # ranks exchange halos every step: the same COUNT on every rank, not a duration
            step()
        torch.cuda.synchronize()
        prewarm_steps = 64
    elif args.prewarm_ms > 0:
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
            for _ in range(8):
                step()
            torch.cuda.synchronize()
            prewarm_steps += 8
    for _ in range(args.warmup):
        step()

    # Two HIP events on the stream the kernels run on bracket the K launches; the average launch duration is their
    # distance / K (it includes the few microseconds between launches: conservative).  An event pair around EVERY step
    # puts two barrier packets between consecutive launches and costs 0.2 % of kernel time and 0.8 % of wall time
    # (tools/evt_test.py: 0.9832 / 0.9893 ms against 0.9808 / 0.9815 ms).
    # The K steps are bracketed by barrier + synchronize on both sides.  What is REPORTED is the job time from the first
    # rank's start to the last rank's end, both read BEFORE the closing barrier on the node's monotonic clock
    # (time.perf_counter() is CLOCK_MONOTONIC: one clock for all ranks of a node): a gloo barrier over 8 local ranks takes
    # 0.3-1 ms, 1.5-5 % of the driver's 20-step window, and N = 1 has none -- timing it would bias the curve.
    t_start, t_end, t_after_barrier, kern_ms = timed(args.steps)
    elapsed = t_end - t_start
    per_rank_ms = [elapsed / args.steps * 1e3]
    closing_barrier_ms = (t_after_barrier - t_end) * 1e3
    if dist is not None:
        t = torch.tensor([t_start, t_end], dtype=torch.float64)
        all_t = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(all_t, t)
        per_rank_ms = [float(x[1] - x[0]) / args.steps * 1e3 for x in all_t]
        elapsed = float(max(x[1] for x in all_t) - min(x[0] for x in all_t))       # whole job: first start -> last end
        if cold is not None:
            c = torch.tensor([c_a, c_b], dtype=torch.float64)
            all_c = [torch.zeros_like(c) for _ in range(world)]
            dist.all_gather(all_c, c)
            cold["ms_per_step"] = round(float(max(x[1] for x in all_c) - min(x[0] for x in all_c)) / args.steps * 1e3, 4)

    # Outside the timed region: what the last step produced.  crc32 of every output plane (frames) / of the
    # stitched plane (stripes), so an N-rank run can be compared with a 1-rank run bit for bit.
    if args.path in ("host", "surface", "surface-dev") and F == 1:
        res = host_out[None]
    elif args.path == "host":
        res = host_frames
    elif args.path == "pipeline":
        res = d_hi.cpu().numpy()
    else:
        res = d_out.cpu().numpy()
    crcs = [zlib.crc32(np.ascontiguousarray(res[k]).tobytes()) for k in range(res.shape[0])]
    chk = int(res[0, res.shape[1] // 2, : min(W, 4096)].astype(np.int64).sum())
    if dist is not None:
        if stripe:
            rows = [sharding.stripe_rows(H, world, k) for k in range(world)]
            pad = max(b - a for a, b in rows)
            buf = torch.zeros((pad, W), dtype=torch.uint8)
            buf[: r1 - r0] = torch.from_numpy(res[0])
            parts = [torch.empty_like(buf) for _ in range(world)]
            dist.all_gather(parts, buf)
            plane = torch.cat([p[: b - a] for p, (a, b) in zip(parts, rows)], dim=0).numpy()
            crcs = [zlib.crc32(plane.tobytes())]
        else:
            t = torch.tensor(crcs, dtype=torch.int64)
            parts = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(parts, t)
            crcs = [int(v) for p in parts for v in p]

    # The OTHER half of the metric ("MPix/s AND ms/image at 1/2/4/8"): ms per image at N > 1 is ONE plane row-striped over the
    # ranks (BASELINE configs[3]).  The driver's one command per N is the frames workload above; this leg rides on it, after
    # its timed region, so that the same line carries both curves.  Never `value`.
    # (It runs LAST, behind the assembly of the line below, under a watchdog: transports that have only ever run with the ranks
    # sharing one GPU must not be able to take the measured frames figure with them if they hang on a real node.)
    stripe_obj = None
    out = None

    if rank == 0:
        pix_per_step = W * H if stripe else W * H * F * world
        value = pix_per_step * args.steps / elapsed / 1e6
        flops_per_launch = S.FLOP_PER_PIXEL * W * (r1 - r0 if stripe else H * F)
        achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12
        # HBM bytes per launch and MFMA-pipe utilisation cannot be read live: they come from rocprofv3 --pmc
        # passes of this same command made by the builder (tools/profile_round.sh) and are labelled as such.
        pmc_ref = None
        pmc_stale = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists() and args.mode in ("mfma", "split16") and not stripe:      # (the REFBYTES modes have no PMC record)
            try:
                from srcnn_cpp_amd.build import kernel_sources_fingerprint
                rec = json.loads(pmc.read_text())
                if rec.get("_kernel_sources") != kernel_sources_fingerprint():
                    # counters of a different build of the kernels are not quoted (re-take them: tools/profile_round.sh)
                    pmc_stale = (f"profiles/pmc_traffic.json was taken with kernel sources {rec.get('_kernel_sources')}, this "
                                 f"build is {kernel_sources_fingerprint()}: not quoted")
                    rec = {}
                key = ("split16" if args.mode == "split16" and args.path == "fused" else args.path) + f"_{W}x{H}x{F}"
                if args.mode == "split16" and args.path != "fused":
                    key = "none"
                if rec.get(key) is not None:
                    pmc_ref = {"traffic": rec.get(key), "mfma_busy_frac": rec.get(key + "_mfma_busy_frac"),
                               "source": "profiles/pmc_traffic.json -- rocprofv3 --pmc passes of this command by the "
                                         "builder, NOT measured in this run",
                               "taken_at": rec.get("_taken_at")}
            except Exception:
                pmc_ref = None
        traffic = pmc_ref["traffic"] if pmc_ref else None
        out = {
            "metric": "SRCNN Y-channel Mpixels/sec", "value": round(value, 2), "unit": "MPix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if stripe else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"ONE {W}x{H} luma plane row-striped over {world} GPU(s), 6-row halo exchange, "
                                    if stripe else f"{F} x {W}x{H} luma plane per GPU per step ")
                                   + ("(1920x1080 x2.0, BASELINE configs[1]), " if (W, H, F) == (3840, 2160, 1) and not stripe else "")
                                   + f"{args.path} conv path, {args.mode} "
                                   "arithmetic, " + ("host buffers over PCIe" if args.path in ("host", "surface", "surface-dev")
                                                     else "inputs resident in HBM"),
                       "frames_per_gpu": F, "width": W, "height": H, "path": args.path, "mode": args.mode,
                       "seam_deferral": bool(deferral),
                       "plan": ctx.query_plan(W, r1 - r0 if stripe else H, F), "output_checksum": chk,
                       "output_crc32": crcs},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": traffic,
                         "traffic_source": pmc_ref["source"] if pmc_ref else pmc_stale,
                         "kernel_ms": round(kern_ms, 4), "flop_per_pixel": S.FLOP_PER_PIXEL},
            "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms],
            "host_us_per_step": round(host_us[0], 2),         # rank 0: time to QUEUE one step (launches, halo posts), not to run it
            "timing": {"what": "first rank's start -> last rank's end of the K steps on the node's monotonic clock, both read "
                               "before the closing barrier (the K steps are bracketed by barrier + synchronize on both sides)",
                       "closing_barrier_ms": round(closing_barrier_ms, 4)},
            # the contract exactly as written -- W warm-up + K timed steps from wherever the clocks were -- measured BEFORE the
            # pre-warm; `value` / `ms_per_step` are the steady-state figures behind it
            "cold_start": cold,
            "prewarm": {"ms": args.prewarm_ms, "steps": prewarm_steps,
                        "why": "untimed clock ramp before the W warm-up steps; an idle GPU runs its first ~20 launches slower"},
        }
        if args.path == "host":
            # what crosses PCIe per step: every frame once each way (u8 in, u8 out); both directions run at the same time in the
            # frame pipeline, so the rate each direction must sustain is bytes / time
            gb = W * H * F * world / 1e9
            out["pcie"] = {"bytes_each_way_per_frame": W * H, "h2d_gbps": round(gb / (elapsed / args.steps), 2),
                           "d2h_gbps": round(gb / (elapsed / args.steps), 2),
                           "what": "host frames (pageable numpy memory) through srcnn_forward_y" + ("_frames: pinned staging, "
                                   "H2D + kernel + D2H of neighbouring frames overlapped on two lanes" if F > 1 else ": row bands")}
        if pmc_ref:
            out["pmc_reference"] = pmc_ref
        if world > 1:
            out["degraded"] = bool(degraded)
            out["distributed"] = {"control_plane": "gloo", "rccl_world": rccl_world,
                                  "halo_transport": (("none per step: the neighbours' stripes are mapped once (HIP IPC), the kernel "
                                                      "loads their 6 edge rows where they lie (xGMI between GPUs)" if form == "peer" else
                                                      "rccl send/recv" if rccl is not None else "host-staged (gloo)")
                                                     if stripe else "none (frames are independent)"),
                                  "stripe_form": form if stripe else None,
                                  "halo_overlap": bool(stripe and form != "assemble")}
        if args.mode in ("refbytes", "refbytes16"):
            out["fixup"] = ctx.fixup_stats()                  # accumulated over every launch of the run
        if args.mode == "refbytes16":
            # opt-in mode: the split-f16 kernel is priced against the dense f16 MFMA peak (as --mode split16 below); the
            # fix-up is vector-ALU work, so the line's frac is only the ALGORITHMIC rate of the whole step over that peak
            out["dtype"] = "f16x2-split operands, f32 accumulate + exact f32 recomputation of flagged pixels"
            out["roofline"].update({"peak": PEAK_F16_MFMA_TFLOPS, "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4),
                                    "vs_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                                    "note": "strip kernel on f16 MFMAs + fix_apply on the vector ALU in one figure"})
        if args.mode == "refbytes":
            out["roofline"]["note"] = ("the step is the f32-MFMA strip kernel (its own frac: --mode mfma) plus fix_apply (+ the idle fix_rerun launch) "
                                       "on the vector ALU; frac here is the algorithmic rate of the whole step over the f32-MFMA peak")
        if args.mode == "split16":
            # opt-in mode: priced against the dense f16 MFMA peak with the same ALGORITHMIC flops; the
            # kernel executes 42 MFMA x 32x32x16 per 32 pixels = 43,008 flop/pixel (2-3 f16 products per MAC)
            peak16 = PEAK_F16_MFMA_TFLOPS
            out["dtype"] = "f16x2-split operands, f32 accumulate"
            out["roofline"].update({"peak": peak16, "frac": round(achieved / peak16, 4),
                                    "executed_mfma_flop_per_pixel": 43008,
                                    "executed_frac": round(achieved * 43008 / S.FLOP_PER_PIXEL / peak16, 4),
                                    "vs_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4)})
        if world == 1 and args.path == "fused" and not stripe and args.sustained_s > 0 and not args.no_cpu_baseline:     # (--no-cpu-baseline marks a quick run)
            try:
                # The timed region is K = 20-50 steps, 20-50 ms of load.  What the device sustains over SECONDS: the same step queued
                # back to back in chunks of 200, every chunk between two events on the kernels' stream.
                chunk, per_chunk, t_s = 200, [], time.perf_counter()
                while time.perf_counter() - t_s < args.sustained_s:
                    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    ea.record(stream)
                    for _ in range(chunk):
                        step()
                    if deferral:
                        ctx.flush()
                    eb.record(stream)
                    eb.synchronize()
                    per_chunk.append(ea.elapsed_time(eb) / chunk)
                ms_s = sum(per_chunk) / len(per_chunk)
                out["sustained"] = {"seconds": round(time.perf_counter() - t_s, 2), "steps": chunk * len(per_chunk),
                                    "ms_per_step": round(ms_s, 4), "value": round(W * H * F / ms_s / 1e3, 2), "unit": "MPix/s",
                                    "frac": round(S.FLOP_PER_PIXEL * W * H * F / (ms_s * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                    "slowest_chunk_ms_per_step": round(max(per_chunk), 4), "fastest_chunk_ms_per_step": round(min(per_chunk), 4),
                                    "what": f"the same step back to back for {args.sustained_s:g} s in chunks of {chunk}, HIP events per chunk; never `value`"}
            except Exception as e:             # noqa: BLE001 -- a secondary leg must not take the measured line with it
                out["sustained"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        if world == 1 and args.path == "fused" and args.mode == "mfma" and not stripe and F == 1 and not args.no_lanes:
            ctx2 = None
            try:
                # TWO LANES (round 6), never `value`: the planes of a stream alternately on two contexts / HIP streams of this ONE GPU
                # (srcnn_forward_y_lanes_dev) -- the next plane's kernel fills the compute units that the previous plane's slowest
                # workgroups leave idle.  Nothing to gain at 3840x2160 (a 2 % tail), a quarter of the step at 576x576.
                ctx2 = S.Context(local_rank)
                ctx2.set_weights_blob(S.load_weights())
                stream2 = torch.cuda.Stream()
                ctx2.set_stream(stream2.cuda_stream)
                d_out2 = torch.zeros_like(d_out)
                n_l = max(args.steps, 200)
                srcs = [d_in.data_ptr()] * n_l
                dsts = [(d_out if k % 2 == 0 else d_out2).data_ptr() for k in range(n_l)]
                ctx.flush()
                torch.cuda.synchronize()

                def lanes_pass():
                    ea, eb, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), torch.cuda.Event()
                    ea.record(stream)
                    stream2.wait_event(ea)
                    S.forward_y_lanes_dev([ctx, ctx2], srcs, W, dsts, W, W, H)
                    e2.record(stream2)
                    stream.wait_event(e2)
                    eb.record(stream)
                    eb.synchronize()
                    return ea.elapsed_time(eb) / n_l
                t_l = time.perf_counter()
                while time.perf_counter() - t_l < 0.3:
                    lanes_pass()
                ms_l = min(lanes_pass() for _ in range(3))
                same = bool(torch.equal(d_out, d_out2)) and zlib.crc32(d_out2.cpu().numpy()[0].tobytes()) == crcs[0]
                out["two_lanes"] = {"ms_per_step": round(ms_l, 4), "value": round(W * H / ms_l / 1e3, 2), "unit": "MPix/s",
                                    "frac": round(S.FLOP_PER_PIXEL * W * H / (ms_l * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                    "vs_one_lane": round(ms_l / (elapsed / args.steps * 1e3), 3), "outputs_equal_the_headline_planes": same,
                                    "what": f"{n_l} planes alternately on two contexts / streams of the one GPU (srcnn_forward_y_lanes_dev), "
                                            "HIP events around the pass, best of 3; never `value`"}
            except Exception as e:             # noqa: BLE001 -- a secondary leg must not take the measured line with it
                out["two_lanes"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
            finally:
                if ctx2 is not None:
                    ctx2.close()
                torch.cuda.set_stream(stream)
        if world == 1 and args.path == "fused" and not stripe and not args.no_e2e:
            try:
                # SURVEY 8d's secondary metric, never `value`: the same planes from and to HOST memory, PCIe-inclusive
                # (srcnn_forward_y_frames: pinned staging, uploads / kernels / downloads of neighbouring frames overlapped).
                n_e2e = 16                                         # (fill and drain of the two-lane pipeline are 0.6 ms: 7 % of 8 frames)
                hf = np.ascontiguousarray(np.broadcast_to(frames[0], (n_e2e,) + frames[0].shape))
                ho = np.empty_like(hf)
                ctx.set_stream(0)                                  # the context's own stream
                ctx.forward_y_frames(hf, out=ho)                   # staging buffers, pinned memory
                reps, t_e = 3, time.perf_counter()
                for _ in range(reps):
                    ctx.forward_y_frames(hf, out=ho)
                dt = (time.perf_counter() - t_e) / reps
                out["e2e"] = {"value": round(W * H * n_e2e / dt / 1e6, 2), "unit": "MPix/s", "ms_per_frame": round(dt / n_e2e * 1e3, 4),
                              "what": f"{n_e2e} x {W}x{H} host frames through srcnn_forward_y_frames, pageable caller memory, H2D + "
                                      f"kernel + D2H overlapped on two lanes; mean of {reps} passes",
                              "output_equals_resident": bool(zlib.crc32(ho[n_e2e - 1].tobytes()) == crcs[0])}
            except Exception as e:             # noqa: BLE001 -- a secondary leg must not take the measured line with it
                out["e2e"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(W, H)
            except Exception as e:             # noqa: BLE001 -- a secondary leg must not take the measured line with it
                out["cpu_baseline"] = {"error": f"{type(e).__name__}: {str(e)[:300]}"}
        if world == 1 and args.path == "fused" and args.mode == "mfma" and not stripe and not args.no_refbytes:
            try:
                # The same step in SRCNN_MODE_REFBYTES (the MFMA kernel + exact recomputation of the pixels next to a truncation
                # boundary): its time, what the fix-up did, and -- against the sha256 the cpu_baseline leg's oracle made of the same
                # frame -- whether the bytes ARE the reference arithmetic's.  Never `value`.
                import hashlib
                ctx.set_stream(stream.cuda_stream)

                def wall(n_steps):
                    for _ in range(10):
                        step()
                    ctx.flush()
                    torch.cuda.synchronize()
                    t_r = time.perf_counter()
                    for _ in range(n_steps):
                        step()
                    ctx.flush()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t_r) / n_steps
                # the MFMA mode launched the way REFBYTES launches (a seam launch per step: the fix-up needs every flag of a plane
                # before it starts, so REFBYTES steps cannot defer theirs): the like-for-like denominator beside the headline's
                ctx.set_seam_deferral(False)
                dt_plain = wall(args.steps)
                ctx.set_mode(S.MODE_REFBYTES)
                dt_r = wall(args.steps)
                rb = d_out.cpu().numpy()
                ref = (out.get("cpu_baseline") or {}).get("reference_output") or {}
                rows = int(ref.get("rows", 0))
                equal = None
                if 0 < rows <= H:
                    # a slab of the top `rows` rows of the plane is exact away from its cut: compare all but its last 6 rows' worth
                    # when it is a crop, the whole plane when it is the whole plane
                    if rows == H:
                        equal = hashlib.sha256(np.ascontiguousarray(rb[0]).tobytes()).hexdigest() == ref.get("sha256")
                out["refbytes"] = {"ms_per_step": round(dt_r * 1e3, 4), "value": round(W * H * F / dt_r / 1e6, 2), "unit": "MPix/s",
                                   "vs_mfma_mode": round(dt_r / (elapsed / args.steps), 3),
                                   "vs_mfma_mode_without_seam_deferral": round(dt_r / dt_plain, 3), "mfma_without_seam_deferral_ms": round(dt_plain * 1e3, 4),
                                   "threshold_factor": 4.0, "device_side_net": True, "fixup": ctx.fixup_stats(),
                                   # round 6: every pixel is flagged against ITS OWN threshold min(delta, k * 2^-24 * S1(x) + abs)
                                   "per_pixel_threshold": dict(zip(("k", "largest_deviation_over_own_threshold_this_run"),
                                                                   (round(v, 4) for v in ctx.fixup_local_stats()))),
                                   # |v - r| / thr of the worst window any search has produced (not measured in this run): the searches
                                   # climb on the k that keeps thr 1.73 x above the deviation and ask for 1.480 - 1.540 (CPU, 241 M / 906 M evaluations;
                                   # GPU 1.480), a longer GPU climb 1.525), the library uses 1.6 (abs 2.44e-4): 0.55 of thr -- profiles/r06/fixup_adversarial_ratio.txt; the
                                   # worst of all is a window of LARGE local scale, where the cap delta binds: 7.93e-4 / 1.376e-3 (round 5's GPU search)
                                   "largest_deviation_any_search_found_over_threshold": 0.577,
                                   "equals_reference_arithmetic": equal,
                                   "checked_against": "sha256 of oracle.forward_y on the same frame (cpu_baseline leg)" if equal is not None
                                                      else "not checked (no whole-plane oracle output in this run)"}
                # ... and in SRCNN_MODE_REFBYTES16: the same fix-up and the same device-side net behind the split-f16 strip kernel
                # (opt-in: f16 MFMAs on (hi, lo) operand pairs carry the pass, the float32 reference arithmetic decides every byte
                # next to a truncation boundary) -- the reference's bytes again, checked against the same sha256.  Never `value`.
                ctx.set_mode(S.MODE_REFBYTES16)
                dt_r16 = wall(args.steps)
                rb16 = d_out.cpu().numpy()
                equal16 = None
                if rows == H:
                    equal16 = hashlib.sha256(np.ascontiguousarray(rb16[0]).tobytes()).hexdigest() == ref.get("sha256")
                out["refbytes16"] = {"ms_per_step": round(dt_r16 * 1e3, 4), "value": round(W * H * F / dt_r16 / 1e6, 2), "unit": "MPix/s",
                                     "vs_mfma_mode": round(dt_r16 / (elapsed / args.steps), 3), "dtype": "f16 (hi, lo) pairs + f32 fix-up",
                                     "threshold_factor": round(4.0 * 8.0 / 6.0, 3), "device_side_net": True, "fixup": ctx.fixup_stats(),
                                     "per_pixel_threshold": dict(zip(("k", "largest_deviation_over_own_threshold_this_run"),
                                                                     (round(v, 4) for v in ctx.fixup_local_stats()))),
                                     # the GPU-side climbs ask for k = 2.023 at 1.73 x, the library uses 2.15 (profiles/r06/adversarial_gpu_ratio.txt);
                                     # where the cap binds: 1.13e-3 / 1.835e-3 (profiles/r05/adversarial_gpu.txt)
                                     "largest_deviation_any_search_found_over_threshold": 0.616,
                                     "equals_reference_arithmetic": equal16,
                                     "equals_refbytes_output": bool(np.array_equal(rb, rb16)),
                                     "note": "opt-in mode outside the float32 north star: never the headline"}
                ctx.set_mode(S.MODE_MFMA)
                ctx.set_seam_deferral(deferral)
            except Exception as e:             # noqa: BLE001 -- a secondary leg must not take the measured line with it
                out.setdefault("refbytes", {"error": f"{type(e).__name__}: {str(e)[:300]}"})
                out.setdefault("refbytes16", {"error": f"{type(e).__name__}: {str(e)[:300]}"})

    if (world > 1 and not stripe and args.path == "fused" and args.mode == "mfma" and F == 1 and not args.no_stripe_leg):
        import threading

        def give_up():
            # runs on the timer's thread while the main thread sits in a collective or a synchronize that does not return:
            # no GPU call, no collective here -- print what was measured and leave
            if rank == 0:
                out["stripe"] = {"error": f"the stripe leg did not finish within {args.stripe_timeout_s:g} s and was abandoned "
                                          "(a transport hung); the frames figure above is unaffected",
                                 "degraded": True, "degraded_why": ["stripe leg abandoned by its watchdog"]}
                emit_line(out)
            os._exit(0)

        guard = None
        if args.stripe_timeout_s > 0:
            guard = threading.Timer(args.stripe_timeout_s, give_up)
            guard.daemon = True
            guard.start()
        if rank == args.hang_stripe_rank:       # test hook: the other ranks wait for this one in the leg's first collective
            time.sleep(1e6)
        stripe_obj = stripe_leg(args, S, torch, dist, ctx, world, rank, timed, make_rccl)
        if guard is not None:
            guard.cancel()
        if rank == 0:
            out["stripe"] = stripe_obj
    if rank == 0:
        emit_line(out)

    if stripe and isinstance(stepper, sharding.PeerStripeStep):
        stepper.close()                # the neighbours' mappings go before the allocations do
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def worker_cxx(args):
    """--host cxx: ONE process, --gpus contexts, the C ABI's several-GPUs entry points (include/srcnn_amd.h).
    stripe : srcnn_forward_y_striped_dev -- every context holds its rows of ONE plane, the 6 halo rows per boundary
             travel device to device (hipMemcpyPeerAsync over xGMI) on a second stream, interior rows first;
             persistent host threads inside the library, one per context.
    frames : one context per GPU, one plane each per step, launches queued from this one host thread, no exchange.
    torch is used for device memory, streams and events only."""
    import numpy as np
    import torch
    import srcnn_cpp_amd as S
    if getattr(args, "lib", None):
        S.use_library(args.lib)
    from srcnn_cpp_amd.synth import synth_batch, synth_luma

    n, W, H, F = args.gpus, args.width, args.height, args.frames
    stripe = args.workload == "stripe"
    if args.path != "fused" or args.mode != "mfma" or (stripe and F != 1):
        raise SystemExit("--host cxx runs the fused float32 path")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    devs = [0] * n if args.shared_gpu else list(range(n))
    if max(devs) >= torch.cuda.device_count():
        raise SystemExit(f"--gpus {n} but {torch.cuda.device_count()} device(s) visible")
    blob = S.load_weights()
    ctxs, streams, d_in, d_out = [], [], [], []
    plane = synth_luma(W, H) if stripe else None
    for k, d in enumerate(devs):
        with torch.cuda.device(d):
            c = S.Context(d)
            c.set_weights_blob(blob)
            st = torch.cuda.Stream(device=d)
            c.set_stream(st.cuda_stream)
            ctxs.append(c)
            streams.append(st)
            if stripe:
                r0, r1 = S.stripe_rows(H, n, k)
                src = plane[r0:r1]
            else:
                src = synth_batch(W, H, F, first_frame=k * F)
            d_in.append(torch.from_numpy(np.ascontiguousarray(src)).to(f"cuda:{d}"))
            d_out.append(torch.zeros_like(d_in[-1]))
    ins, outs = [t.data_ptr() for t in d_in], [t.data_ptr() for t in d_out]

    def step():
        if stripe:
            S.forward_y_striped_dev(ctxs, ins, W, outs, W, W, H)
        else:
            for k, c in enumerate(ctxs):
                c.forward_y_dev(ins[k], W, H * W, outs[k], W, H * W, W, H, F)

    def sync():
        for c in ctxs:
            c.synchronize()

    def timed(k_steps):
        evs = []
        sync()
        for k, d in enumerate(devs):
            with torch.cuda.device(d):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(streams[k])
                evs.append((a, b))
        t_a = time.perf_counter()
        for _ in range(k_steps):
            step()
        t_q = time.perf_counter()                               # everything queued: the host's share of the K steps
        for k, d in enumerate(devs):
            with torch.cuda.device(d):
                evs[k][1].record(streams[k])
        sync()
        t_b = time.perf_counter()
        kern = [a.elapsed_time(b) / max(1, k_steps) for a, b in evs]
        return t_b - t_a, (t_q - t_a) / max(1, k_steps) * 1e6, kern

    cold = None
    if args.prewarm_ms > 0:
        for _ in range(args.warmup):
            step()
        c_el, _, c_kern = timed(args.steps)
        cold = {"ms_per_step": round(c_el / args.steps * 1e3, 4), "kernel_ms": round(max(c_kern), 4)}
        t_pw = time.perf_counter()
        while (time.perf_counter() - t_pw) * 1e3 < args.prewarm_ms:
            for _ in range(8):
                step()
            sync()
    for _ in range(args.warmup):
        step()
    elapsed, host_us, kern = timed(args.steps)

    if stripe:
        res = np.concatenate([t.cpu().numpy() for t in d_out], axis=0)
        crcs = [zlib.crc32(res.tobytes())]
    else:
        crcs = [zlib.crc32(np.ascontiguousarray(t[f].cpu().numpy()).tobytes()) for t in d_out for f in range(F)]
    pix_per_step = W * H if stripe else W * H * F * n
    rows0 = (S.stripe_rows(H, n, 0)[1] - S.stripe_rows(H, n, 0)[0]) if stripe else H * F
    achieved = S.FLOP_PER_PIXEL * W * rows0 / (kern[0] * 1e-3) / 1e12
    out = {
        "metric": "SRCNN Y-channel Mpixels/sec", "value": round(pix_per_step * args.steps / elapsed / 1e6, 2), "unit": "MPix/s",
        "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if stripe else "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": (f"ONE {W}x{H} luma plane row-striped over {n} GPU(s), 6-row halo copies device to device, "
                                if stripe else f"{F} x {W}x{H} luma plane per GPU per step, ")
                               + "fused conv path, mfma arithmetic, inputs resident in HBM, ONE host process (C ABI, --host cxx)",
                   "host": "cxx", "frames_per_gpu": F, "width": W, "height": H, "path": "fused", "mode": "mfma",
                   "devices": devs, "output_crc32": crcs},
        "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                     "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                     "kernel_ms": round(kern[0], 4), "flop_per_pixel": S.FLOP_PER_PIXEL,
                     "what": "context 0's launches of one step (its stripe / its frames), HIP events on its stream"},
        "per_rank_ms_per_step": [round(v, 4) for v in kern],
        "host_us_per_step": round(host_us, 2),
        "cold_start": cold,
        "distributed": {"control_plane": "none (one process)",
                        "halo_transport": ({1: "none: the contexts share a device, the kernel reads the neighbours' rows where they lie",
                                            2: "peer access: the kernel loads the neighbours' 6 edge rows over xGMI, no copy",
                                            3: "hipMemcpyPeerAsync WITHOUT peer access: staged through host memory by the runtime"}
                                           .get(max(c.halo_transport() for c in ctxs), "none (one stripe)")
                                           if stripe else "none (frames are independent)"),
                        "stripe_form": "halo" if stripe else None, "halo_overlap": bool(stripe)},
    }
    if stripe and any(c.halo_transport() == 3 for c in ctxs):
        # a link refused peer access: the rows crossed host memory -- not the configs[3] transport; say so like the Python path
        out["degraded"] = True
        out["degraded_why"] = "; ".join(sorted({c._lib.srcnn_last_error(c._h).decode() for c in ctxs if c.halo_transport() == 3}))
    emit_line(out)
    for c in ctxs:
        c.close()


def main():
    args = parse_args()
    if args.cpu_baseline_only:
        cpu_baseline_child(args.width, args.height)
        return 0
    if args.host == "cxx":
        if int(os.environ.get("RANK", "0")) != 0:
            return 0                             # under torch.distributed.run: one process does the work
        worker_cxx(args)
        return 0
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args.gpus)          # before anything in this process touches the GPU
    worker(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
