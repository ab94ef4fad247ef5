#!/usr/bin/env python3
"""bench.py -- SRCNN Y-channel MPix/s on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (fused Convolution99x11 + Convolution55,
src/srcnn.cpp:609+627) over one synthetic 3840x2160 luma plane per GPU
(BASELINE.json configs[1]: 1920x1080 x2.0), input already resident in HBM.
N GPUs = N ranks, one frame per rank per step, no collective on the data path
(frames are independent: weak scaling).

Prints ONE JSON line on rank 0, including
  roofline      dominant kernel vs the f32 MFMA peak (HIP events per launch)
  cpu_baseline  the oracle (port of the reference loops) on the host cores, N=1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: 256 CU x 256 FLOP/clk x 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2516.6     # dense f16/bf16: v_mfma_f32_32x32x16_f16 = 32,768 FLOP / 32 clk / SIMD x 1024 SIMD x 2.4 GHz


def cpu_baseline(width, height, target_s=12.0):
    """Time the oracle (reference loops, strict IEEE, OpenMP) on a bounded slab
    of the same workload; rows are independent so a slab of full-width rows has
    the per-pixel cost of the whole frame."""
    import numpy as np
    import oracle
    import srcnn_cpp_amd as S
    from srcnn_cpp_amd.synth import synth_luma

    cores = len(os.sched_getaffinity(0))
    oracle.set_threads(cores)
    blob = S.load_weights()
    frame = synth_luma(width, height)
    probe_rows = min(height, max(cores, 16))
    oracle.forward_y(frame[:probe_rows], blob)                 # warm-up: page in, spin up the OpenMP team
    t = time.perf_counter()
    oracle.forward_y(frame[:probe_rows], blob)
    dt = time.perf_counter() - t
    rows = int(min(height, max(probe_rows, probe_rows * target_s / max(dt, 1e-6))))
    reps, pix, t = 0, 0, time.perf_counter()
    while True:                                                # whole slabs until ~target_s of CPU work
        oracle.forward_y(frame[:rows], blob)
        reps += 1
        pix += width * rows
        dt = time.perf_counter() - t
        if dt >= target_s or reps >= 64:
            break
    return {"value": round(pix / dt / 1e6, 4), "unit": "MPix/s", "cores": cores, "kind": "port",
            "sample": f"{reps} x top {rows} of {height} rows of the {width}x{height} frame, "
                      f"oracle/srcnn_oracle.c (reference loops, -O3 -ffp-contract=off), OpenMP {cores} threads, "
                      f"{dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--frames", type=int, default=1, help="frames per GPU per step")
    ap.add_argument("--path", choices=["fused", "unfused", "host", "pipeline"], default="fused",
                    help="fused: one kernel, u8 in/out (default); unfused: layer-1/2 kernel -> 32 f32 planes in "
                         "HBM -> layer-3 kernel; host: srcnn_forward_y on pageable host buffers (PCIe-inclusive)")
    ap.add_argument("--mode", choices=["mfma", "exact", "split16"], default="mfma",
                    help="mfma: float32 MFMA (default, the headline); exact: reference arithmetic on the vector ALU; "
                         "split16: opt-in f16-MFMA mode with (hi, lo) operand splitting (SURVEY.md 8f rank 4) -- "
                         "never the headline number")
    ap.add_argument("--workload", choices=["frames", "stripe"], default="frames",
                    help="frames: independent planes per rank, no collective, weak scaling (default); "
                         "stripe: ONE width x height plane row-striped over the ranks with a 6-row "
                         "point-to-point halo exchange per step, strong scaling (BASELINE configs[3])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="torch.distributed backend for the barrier / max-over-ranks (nccl == RCCL)")
    ap.add_argument("--shared-gpu", action="store_true",
                    help="smoke-test aid: every rank uses GPU 0 (1-GPU box, use with --backend gloo)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import srcnn_cpp_amd as S
    from srcnn_cpp_amd.synth import synth_batch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.shared_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

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
    if stripe:
        from srcnn_cpp_amd import sharding
        from srcnn_cpp_amd.synth import synth_luma
        if args.path != "fused" or F != 1:
            raise SystemExit("--workload stripe runs the fused path on one plane")
        r0, r1 = sharding.stripe_rows(H, world, rank)
        frames = synth_luma(W, H)[None, r0:r1].copy()          # this rank's rows of the plane
        compute_rows = sharding.gpu_compute_rows(ctx)
    else:
        frames = synth_batch(W, H, F, first_frame=rank * F)
    d_in = torch.from_numpy(frames).cuda()
    d_out = torch.zeros_like(d_in)
    d_work = torch.empty((F, 32, H, W), dtype=torch.float32, device="cuda") if args.path == "unfused" else None

    host_out = np.empty_like(frames[0])
    host_frames = np.empty_like(frames) if args.path == "host" and F > 1 else None
    if args.path == "pipeline":
        if W % 2 or H % 2 or stripe:
            raise SystemExit("--path pipeline upsamples x2: width and height must be even")
        lo = np.stack([frames[:, ::2, ::2]] * 3, axis=-1).copy()          # [F, H/2, W/2, 3] B,G,R
        d_lo = torch.from_numpy(lo).cuda()
        d_hi = torch.zeros((F, H, W, 3), dtype=torch.uint8, device="cuda")

    def step():
        if stripe:
            if world > 1 and args.backend == "gloo":            # smoke-test aid: halo over host memory
                ext, s0 = sharding.exchange_halo(d_in[0].cpu(), H, world, rank)
                ext = ext.cuda()
            else:                                               # RCCL send/recv over xGMI
                ext, s0 = sharding.exchange_halo(d_in[0], H, world, rank)
            ctx.forward_y_rows_dev(ext.data_ptr(), ext.stride(0), s0, d_out.data_ptr(), W, r0, W, H, r0, r1)
        elif args.path == "pipeline":
            for k in range(F):
                ctx.process_bgr_dev(d_lo[k].data_ptr(), 3 * (W // 2), W // 2, H // 2, 2.0, d_hi[k].data_ptr(), 3 * W)
        elif args.path == "host":
            if F == 1:
                ctx.forward_y(frames[0], dst=host_out)
            else:                                             # stream of host frames, transfers overlapped
                ctx.forward_y_frames(frames, out=host_frames)
        elif args.path == "fused":
            ctx.forward_y_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, F)
        else:
            ctx.forward_y_unfused_dev(d_in.data_ptr(), W, H * W, d_out.data_ptr(), W, H * W, W, H, F,
                                      d_work.data_ptr())

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for a, b in ev:                      # HIP events on the stream the kernels run on
        a.record(stream)
        step()
        b.record(stream)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0

    kern_ms = sum(a.elapsed_time(b) for a, b in ev) / max(1, args.steps)
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # sanity: the result of the last step is the real thing (bitwise == frame 0 recomputed)
    chk = int(d_out[0, d_out.shape[1] // 2, : min(W, 4096)].to(torch.int64).sum().item())

    if rank == 0:
        pix_per_step = W * H if stripe else W * H * F * world
        value = pix_per_step * args.steps / elapsed / 1e6
        flops_per_launch = S.FLOP_PER_PIXEL * W * (r1 - r0 if stripe else H * F)
        achieved = flops_per_launch / (kern_ms * 1e-3) / 1e12
        # HBM bytes per launch and MFMA-pipe utilisation come from rocprofv3 PMC passes of this same
        # command (separate --pmc runs, profiles/r01/README.md); they cannot be read live.
        traffic = mfma_busy = None
        pmc = ROOT / "profiles" / "pmc_traffic.json"
        if pmc.exists() and args.mode in ("mfma", "split16") and not stripe:
            try:
                rec = json.loads(pmc.read_text())
                key = ("split16" if args.mode == "split16" and args.path == "fused" else args.path) + f"_{W}x{H}x{F}"
                if args.mode == "split16" and args.path != "fused":
                    key = "none"
                traffic = rec.get(key)
                mfma_busy = rec.get(key + "_mfma_busy_frac")
            except Exception:
                traffic = mfma_busy = None
        out = {
            "metric": "SRCNN Y-channel Mpixels/sec", "value": round(value, 2), "unit": "MPix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
            "scaling": "strong" if stripe else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"ONE {W}x{H} luma plane row-striped over {world} GPU(s), 6-row halo exchange, "
                                    if stripe else f"{F} x {W}x{H} luma plane per GPU per step ")
                                   + ("(1920x1080 x2.0, BASELINE configs[1]), " if (W, H, F) == (3840, 2160, 1) and not stripe else "")
                                   + f"{args.path} conv path, {args.mode} "
                                   "arithmetic, " + ("host buffers over PCIe" if args.path == "host"
                                                     else "inputs resident in HBM"),
                       "frames_per_gpu": F, "width": W, "height": H, "path": args.path, "mode": args.mode,
                       "plan": ctx.query_plan(W, r1 - r0 if stripe else H, F), "output_checksum": chk},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
                         "traffic": traffic,
                         "hbm_gbps": round(traffic / (kern_ms * 1e-3) / 1e9, 1) if traffic else None,
                         "mfma_busy_frac_pmc": mfma_busy, "kernel_ms": round(kern_ms, 4),
                         "flop_per_pixel": S.FLOP_PER_PIXEL},
        }
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


if __name__ == "__main__":
    main()
