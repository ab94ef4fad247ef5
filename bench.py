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
    import numpy as np  # noqa: F401
    import oracle
    import srcnn_cpp_amd as S
    from srcnn_cpp_amd.synth import synth_luma

    model, logical, physical = host_cpu_info()
    threads = int(os.environ.get("OMP_NUM_THREADS", physical))
    blob = S.load_weights()
    frame = synth_luma(width, height)
    v, reps, rows, dt = _time_oracle(lambda f: oracle.forward_y(f, blob), frame, width, height, target_s, threads)
    out = {"value": round(v, 4), "unit": "MPix/s", "cores": threads, "kind": "port",
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


def launch_ranks(n: int) -> int:
    """Parent of a multi-rank run: start n fresh rank processes (this process has made no GPU call and
    makes none), forward rank 0's stdout, return the worst exit code."""
    port = os.environ.get("MASTER_PORT") or str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve())] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    deadline = time.time() + 120
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=max(1.0, deadline - time.time())))
        except subprocess.TimeoutExpired:
            p.kill()                                           # exactly the child this process started
            rcs.append(-9)
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [rc for rc in rcs if rc != 0]
    return bad[0] if bad else 0


# --------------------------------------------------------------------------- one rank

def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=1, help="frames per GPU per step")
    ap.add_argument("--path", choices=["fused", "unfused", "host", "pipeline", "surface"], default="fused",
                    help="fused: one kernel, u8 in/out (default); unfused: layer-1/2 kernel -> 32 f32 planes in "
                         "HBM -> layer-3 kernel; host: srcnn_forward_y on host buffers (PCIe-inclusive); "
                         "surface: the reference call surface on host buffers -- srcnn_conv99x11 then srcnn_conv55 "
                         "(what include/srcnn_amd.hpp's Convolution99x11 / Convolution55 call), 32 f32 planes over PCIe")
    ap.add_argument("--mode", choices=["mfma", "exact", "split16"], default="mfma",
                    help="mfma: float32 MFMA (default, the headline); exact: reference arithmetic on the vector ALU; "
                         "split16: opt-in f16-MFMA mode with (hi, lo) operand splitting (SURVEY.md 8f rank 4) -- "
                         "never the headline number")
    ap.add_argument("--workload", choices=["frames", "stripe"], default="frames",
                    help="frames: independent planes per rank, no collective, weak scaling (default); "
                         "stripe: ONE width x height plane row-striped over the ranks with a 6-row "
                         "point-to-point halo exchange per step, strong scaling (BASELINE configs[3])")
    ap.add_argument("--no-overlap", action="store_true",
                    help="stripe workload: wait for the halo before the one launch (default: interior rows first, "
                         "the two 6-row edge bands after the exchange)")
    ap.add_argument("--prewarm-ms", type=float, default=400.0,
                    help="untimed device wake-up BEFORE the W warm-up steps: the same step, repeated for this long.  An idle "
                         "MI355X needs ~20 launches (20-30 ms) of load to reach its steady clocks -- the first steps after "
                         "idle run 4-10 %% slower (profiles/r02/clock_ramp.txt) -- and the contract's W = 5 steps are 5 ms.  "
                         "0 disables it.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="transport of the stripe halo exchange (nccl == RCCL over xGMI; gloo stages through host "
                         "memory).  Barriers and the max-over-ranks always use a gloo group.")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="smoke-test aid: every rank uses GPU 0 (1-GPU box, use with --backend gloo)")
    return ap.parse_args()


def worker(args):
    import numpy as np
    import torch
    import srcnn_cpp_amd as S
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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # Control plane (barrier, max over ranks, checksum gather): gloo, host tensors.  The data path has
        # no collective for frames; for stripes the 6 halo rows go neighbour to neighbour over RCCL.
        dist.init_process_group("gloo")
        if args.backend == "nccl" and args.workload == "stripe":    # frames exchange nothing: no RCCL communicator is built for them
            try:
                rccl = dist.new_group(backend="nccl")
                one = torch.ones(1, device="cuda")
                dist.all_reduce(one, group=rccl)
                torch.cuda.synchronize()
                rccl_world = int(one.item())                   # what RCCL itself summed over: must be `world`
                if rccl_world != world:
                    raise RuntimeError(f"RCCL reduced over {rccl_world} ranks, expected {world}")
            except Exception as e:                              # keep the run alive on the host-staged path
                print(f"[rank {rank}] RCCL group unavailable ({e}); halo rows staged through host memory",
                      file=sys.stderr, flush=True)
                rccl, rccl_world = None, None

    W, H, F = args.width, args.height, args.frames
    ctx = S.Context(local_rank)
    ctx.set_weights_blob(S.load_weights())
    if args.mode == "exact":
        ctx.set_mode(S.MODE_EXACT)
    elif args.mode == "split16":
        ctx.set_mode(S.MODE_SPLIT16)
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
        launch_rows = sharding.gpu_launch_rows(ctx)
    else:
        frames = synth_batch(W, H, F, first_frame=rank * F)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    plpad = int(os.environ.get("SRCNN_DEBUG_PLPAD", "0"))           # experiment knob: see srcnn_forward_y_unfused_dev
    d_work = torch.empty((F * 32 * (H * W + plpad),), dtype=torch.float32, device="cuda") if args.path == "unfused" else None

    host_out = np.empty_like(frames[0])
    host_frames = np.empty_like(frames) if args.path == "host" and F > 1 else None
    if args.path == "pipeline":
        if W % 2 or H % 2 or stripe:
            raise SystemExit("--path pipeline upsamples x2: width and height must be even")
        lo = np.stack([frames[:, ::2, ::2]] * 3, axis=-1).copy()          # [F, H/2, W/2, 3] B,G,R
        d_lo = torch.from_numpy(lo).cuda()
        d_hi = torch.zeros((F, H, W, 3), dtype=torch.uint8, device="cuda")
    if args.path == "surface":
        if F != 1 or stripe:
            raise SystemExit("--path surface runs one plane")
        w1, b1, w2, b2, w3, b3 = S.split_weights(S.load_weights())
        host_planes = [np.empty((H, W), np.float32) for _ in range(32)]   # the reference's vector<Mat> (:602-607)

    def step():
        if stripe:
            sharding.forward_striped_launch(d_in[0], d_out[0], H, world, rank, launch_rows, group=rccl,
                                            overlap=not args.no_overlap, via_host=world > 1 and rccl is None)
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
        elif args.path == "fused":
            ctx.forward_y_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, F)
        else:
            ctx.forward_y_unfused_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, F,
                                      d_work.data_ptr())

    def barrier():
        if dist is not None:
            dist.barrier()

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
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()

    # Two HIP events on the stream the kernels run on bracket the K launches; the average launch duration is their
    # distance / K (it includes the few microseconds between launches: conservative).  An event pair around EVERY step
    # puts two barrier packets between consecutive launches and costs 0.2 % of kernel time and 0.8 % of wall time
    # (tools/evt_test.py: 0.9832 / 0.9893 ms against 0.9808 / 0.9815 ms).
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_a.record(stream)
    for _ in range(args.steps):
        step()
    ev_b.record(stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kern_ms = ev_a.elapsed_time(ev_b) / max(1, args.steps)
    per_rank_ms = [elapsed / args.steps * 1e3]
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        all_t = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(all_t, t)
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in all_t]
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # Outside the timed region: what the last step produced.  crc32 of every output plane (frames) / of the
    # stitched plane (stripes), so an N-rank run can be compared with a 1-rank run bit for bit.
    if args.path in ("host", "surface") and F == 1:
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

    if rank == 0:
        pix_per_step = W * H if stripe else W * H * F * world
        value = pix_per_step * args.steps / elapsed / 1e6
        flops_per_launch = S.FLOP_PER_PIXEL * W * (r1 - r0 if stripe else H * F)
        achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12
        # HBM bytes per launch and MFMA-pipe utilisation cannot be read live: they come from rocprofv3 --pmc
        # passes of this same command made by the builder (tools/profile_round.sh) and are labelled as such.
        pmc_ref = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists() and args.mode in ("mfma", "split16") and not stripe:
            try:
                rec = json.loads(pmc.read_text())
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
                                   "arithmetic, " + ("host buffers over PCIe" if args.path in ("host", "surface")
                                                     else "inputs resident in HBM"),
                       "frames_per_gpu": F, "width": W, "height": H, "path": args.path, "mode": args.mode,
                       "plan": ctx.query_plan(W, r1 - r0 if stripe else H, F), "output_checksum": chk,
                       "output_crc32": crcs},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": traffic,
                         "traffic_source": pmc_ref["source"] if pmc_ref else None,
                         "kernel_ms": round(kern_ms, 4), "flop_per_pixel": S.FLOP_PER_PIXEL},
            "per_rank_ms_per_step": [round(v, 4) for v in per_rank_ms],
            "prewarm": {"ms": args.prewarm_ms, "steps": prewarm_steps,
                        "why": "untimed clock ramp before the W warm-up steps; an idle GPU runs its first ~20 launches slower"},
        }
        if pmc_ref:
            out["pmc_reference"] = pmc_ref
        if world > 1:
            out["distributed"] = {"control_plane": "gloo", "rccl_world": rccl_world,
                                  "halo_transport": (("rccl send/recv" if rccl is not None else "host-staged (gloo)")
                                                     if stripe else "none (frames are independent)"),
                                  "halo_overlap": bool(stripe and not args.no_overlap)}
        if args.mode == "split16":
            # opt-in mode: priced against the dense f16 MFMA peak with the same ALGORITHMIC flops; the
            # kernel executes 42 MFMA x 32x32x16 per 32 pixels = 43,008 flop/pixel (2-3 f16 products per MAC)
            peak16 = PEAK_F16_MFMA_TFLOPS
            out["dtype"] = "f16x2-split operands, f32 accumulate"
            out["roofline"].update({"peak": peak16, "frac": round(achieved / peak16, 4),
                                    "executed_mfma_flop_per_pixel": 43008,
                                    "executed_frac": round(achieved * 43008 / S.FLOP_PER_PIXEL / peak16, 4),
                                    "vs_f32_mfma_peak": round(achieved / PEAK_F32_MFMA_TFLOPS, 4)})
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(W, H)
        print(json.dumps(out), flush=True)

    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.cpu_baseline_only:
        cpu_baseline_child(args.width, args.height)
        return 0
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args.gpus)          # before anything in this process touches the GPU
    worker(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
