#!/usr/bin/env python3
"""What ONE rank does at N = 1 / 2 / 4 / 8, timed on the one GPU of the box -- the N > 1 half of BASELINE's metric without
an 8-GPU node.

For every N and every rank k of a row-striped 7680x4320 plane (BASELINE configs[3]) this runs exactly that rank's step
ALONE on the GPU: its rows where a rank holds them, the 6 halo rows either side pre-placed where the exchange would put
them, the launches of each step form, HIP events around K pre-warmed steps on the stream the kernels run on:

  halo      ONE launch on the stripe where it lies, halo rows in two small buffers (srcnn_forward_y_rows_halo_dev);
            what the one-process host does with peer access: the kernel loads the neighbours' edge rows where they lie.
            `halo+copy` adds two 46 KB copy KERNELS on a second stream and the event hand-over per step, four buffer sets
            in turn -- the shape of sharding.StripeStep's RCCL exchange (its kernels need a compute unit too) and of the
            one-process host on a link without peer access
  bands     rounds 2-3: interior rows first, then the two 6-row edge bands from [6 halo | 12 own] buffers (three strip
            launches and their seam launches)
  assemble  one launch on a [halo | stripe | halo] copy of the rows

and prints per-rank ms, the fraction of the f32 MFMA peak that rank's kernel work reaches, and the PROJECTED speed-up
t(N = 1) / max_k t_k(N): what an N-GPU node delivers if the exchange itself hides as designed (the xGMI copy of 46 KB is
~10 us and overlaps the previous step's kernel; it cannot be measured here).  The frames workload (configs[2], configs[4])
is one independent launch sequence per rank, so its projection is N x the one-rank rate; printed for completeness.

usage: python tools/stripe_projection.py [--width 7680 --height 4320 --steps 40] > profiles/r04/stripe_projection.txt
"""
import argparse
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import srcnn_cpp_amd as S  # noqa: E402
from srcnn_cpp_amd.synth import synth_luma  # noqa: E402

PEAK = 157.3e12
HALO = 6
SETS = 4        # halo buffer sets used in turn (sharding.HALO_SETS, srcnn_ctx::kHaloSets)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=7680)
    ap.add_argument("--height", type=int, default=4320)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--mode", default="mfma", choices=["mfma", "refbytes"])
    ap.add_argument("--ns", default="1,2,4,8")
    ap.add_argument("--seam-deferral", choices=["on", "off"], default="on",
                    help="on (default, round 5): srcnn_set_seam_deferral(1) -- the seam blocks of step k ride behind the work items of step "
                         "k + 1, the last step's are queued (srcnn_flush) inside the timed window; off: a seam launch per step (round 4)")
    ap.add_argument("--lib", default=None, help="another build of the library (the tuning build, for the planner's experiment knobs)")
    ap.add_argument("--diag", action="store_true", help="N = 8, rank 1 only: where the cost of the copy + hand-over lies")
    args = ap.parse_args()
    W, H, K = args.width, args.height, args.steps
    if args.lib:
        S.use_library(args.lib)
    plane = synth_luma(W, H)
    ctx = S.Context(0)
    ctx.set_weights_blob(S.load_weights())
    if args.mode == "refbytes":
        ctx.set_mode(S.MODE_REFBYTES)
    deferral = args.seam_deferral == "on" and args.mode == "mfma"
    print(f"# seam deferral: {'on' if deferral else 'off'}")
    stream, side = torch.cuda.Stream(), torch.cuda.Stream()
    ctx.set_stream(stream.cuda_stream)
    d_plane = torch.from_numpy(plane).cuda()
    whole = torch.zeros_like(d_plane)
    torch.cuda.synchronize()
    ctx.forward_y_dev(d_plane.data_ptr(), W, W * H, whole.data_ptr(), W, W * H, W, H, 1)
    ctx.synchronize()
    whole = whole.cpu().numpy()
    if deferral:
        ctx.set_seam_deferral(True)

    def timed(step, check=None):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:           # clock ramp (profiles/r02/clock_ramp.txt)
            for _ in range(8):
                step()
            torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(3):
            a.record(stream)
            for _ in range(K):
                step()
            if deferral:
                ctx.flush()
            b.record(stream)
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) / K)
        if check is not None:
            check()
        return best

    if args.diag:
        n, k = 8, 1
        r0, r1 = S.stripe_rows(H, n, k)
        rows = r1 - r0
        own = d_plane[r0:r1]
        out = torch.zeros((rows, W), dtype=torch.uint8, device="cuda")
        tops = [d_plane[r0 - HALO:r0].clone() for _ in range(2)]
        bots = [d_plane[r1:r1 + HALO].clone() for _ in range(2)]
        nb_top, nb_bot = d_plane[r0 - HALO:r0], d_plane[r1:r1 + HALO]

        def launch(par):
            ctx.forward_y_rows_halo_dev(own.data_ptr(), W, r0, rows, tops[par].data_ptr(), bots[par].data_ptr(), W,
                                        out.data_ptr(), W, r0, W, H, r0, r1)
        st = {"i": 0, "free": [None, None]}

        def v0():
            launch(0)

        def v1():                      # copies on the kernels' own stream, no events
            with torch.cuda.stream(stream):
                tops[0].copy_(nb_top, non_blocking=True)
                bots[0].copy_(nb_bot, non_blocking=True)
            launch(0)

        def v1b():                     # ONE copy kernel for both halos on the kernels' own stream
            with torch.cuda.stream(stream):
                tops[0].copy_(nb_top, non_blocking=True)
            launch(0)

        def v2(copies=True, wait_free=True):
            par = st["i"] & 1
            st["i"] += 1
            if wait_free and st["free"][par] is not None:
                side.wait_event(st["free"][par])
            with torch.cuda.stream(side):
                if copies:
                    tops[par].copy_(nb_top, non_blocking=True)
                    bots[par].copy_(nb_bot, non_blocking=True)
                ready = side.record_event()
            stream.wait_event(ready)
            launch(par)
            if wait_free:
                st["free"][par] = stream.record_event()

        def v4():                      # only an event recorded on the main stream per step
            launch(0)
            stream.record_event()
        for name, fn in (("launch only", v0), ("2 copies, same stream", v1), ("1 copy, same stream", v1b),
                         ("2 streams: copies + both events", v2), ("2 streams: events only", lambda: v2(False)),
                         ("2 streams: copies, no free-event", lambda: v2(True, False)),
                         ("2 streams: ready-event only, no copies", lambda: v2(False, False)),
                         ("record one event per step", v4)):
            print(f"diag N=8 rank 1: {name:<42} {timed(fn):.4f} ms")
        ctx.close()
        return
    print(f"# tools/stripe_projection.py: {W}x{H} plane, mode {args.mode}, {K} steps x 3 (best), one MI355X; "
          f"plan for the whole plane: {ctx.query_plan(W, H)}")
    print("# per-rank step, ms (fraction of the 157.3 TFLOP/s f32 MFMA peak for that rank's rows)")
    print(f"# {'N':>2} {'rank':>4} {'rows':>5}  {'halo':>14} {'halo+copy':>14} {'bands':>14} {'assemble':>14}")
    t1 = {}
    table = {}
    for n in [int(x) for x in args.ns.split(",")]:
        worst = {}
        for k in range(n):
            r0, r1 = S.stripe_rows(H, n, k)
            rows = r1 - r0
            has_top, has_bot = k > 0, k < n - 1
            own = d_plane[r0:r1]                                   # the rank's rows, where they lie
            out = torch.zeros((rows, W), dtype=torch.uint8, device="cuda")
            tops = [d_plane[r0 - HALO:r0].clone() if has_top else None for _ in range(SETS)]
            bots = [d_plane[r1:r1 + HALO].clone() if has_bot else None for _ in range(SETS)]
            nb_top = d_plane[r0 - HALO:r0] if has_top else None    # where the neighbours' edge rows lie
            nb_bot = d_plane[r1:r1 + HALO] if has_bot else None
            flops = S.FLOP_PER_PIXEL * W * rows

            def check():
                assert np.array_equal(out.cpu().numpy(), whole[r0:r1]), (n, k)

            def halo_step(par=0):
                ctx.forward_y_rows_halo_dev(own.data_ptr(), W, r0, rows, tops[par].data_ptr() if has_top else 0,
                                            bots[par].data_ptr() if has_bot else 0, W, out.data_ptr(), W, r0, W, H, r0, r1)
            res = {}
            if n == 1:
                def one():
                    ctx.forward_y_rows_dev(own.data_ptr(), W, 0, out.data_ptr(), W, 0, W, H, 0, H)
                res = {f: timed(one, check) for f in ("halo", "halo+copy", "bands", "assemble")}
            else:
                res["halo"] = timed(halo_step, check)
                # the product's step: copies on a second stream into the set of this step, event hand-over, one launch
                state = {"i": 0, "free": [None] * SETS}

                def halo_copy_step():
                    par = state["i"] % SETS
                    state["i"] += 1
                    if state["free"][par] is not None:
                        side.wait_event(state["free"][par])
                    with torch.cuda.stream(side):
                        if has_top:
                            tops[par].copy_(nb_top, non_blocking=True)
                        if has_bot:
                            bots[par].copy_(nb_bot, non_blocking=True)
                        ready = side.record_event()
                    stream.wait_event(ready)
                    halo_step(par)
                    state["free"][par] = stream.record_event()
                res["halo+copy"] = timed(halo_copy_step, check)
                # bands: [6 halo | 12 own] / [12 own | 6 halo] buffers pre-placed
                i0, i1 = (r0 + HALO if has_top else r0), (r1 - HALO if has_bot else r1)
                top_buf = d_plane[r0 - HALO:r0 + 2 * HALO].clone() if has_top else None
                bot_buf = d_plane[r1 - 2 * HALO:r1 + HALO].clone() if has_bot else None

                def bands_step():
                    ctx.forward_y_rows_dev(own.data_ptr(), W, r0, out.data_ptr(), W, r0, W, H, i0, i1)
                    if has_top:
                        ctx.forward_y_rows_dev(top_buf.data_ptr(), W, r0 - HALO, out.data_ptr(), W, r0, W, H, r0, i0)
                    if has_bot:
                        ctx.forward_y_rows_dev(bot_buf.data_ptr(), W, r1 - 2 * HALO, out.data_ptr(), W, r0, W, H, i1, r1)
                res["bands"] = timed(bands_step, check)
                s0, s1 = max(0, r0 - HALO), min(H, r1 + HALO)
                ext = d_plane[s0:s1].clone()

                def assemble_step():
                    ext[r0 - s0:r1 - s0].copy_(own, non_blocking=True)         # on torch's current stream = the context's
                    ctx.forward_y_rows_dev(ext.data_ptr(), W, s0, out.data_ptr(), W, r0, W, H, r0, r1)
                with torch.cuda.stream(stream):
                    res["assemble"] = timed(assemble_step, check)
            for f, t in res.items():
                worst[f] = max(worst.get(f, 0.0), t)
            print(f"  {n:>2} {k:>4} {rows:>5}  " + " ".join(f"{res[f]:7.4f} ({flops / (res[f] * 1e-3) / PEAK:5.3f})"
                                                               for f in ("halo", "halo+copy", "bands", "assemble")))
        table[n] = worst
        if n == 1:
            t1 = dict(worst)
    print("# projected scaling of the row-striped plane: t(N=1) / max_k t_k(N)   (ideal = N)")
    print(f"# {'N':>2}  {'halo':>8} {'halo+copy':>10} {'bands':>8} {'assemble':>9}   MPix/s (halo+copy)")
    for n, w in table.items():
        print(f"  {n:>2}  " + " ".join(f"{t1[f] / w[f]:8.2f}" if f != "halo+copy" else f"{t1[f] / w[f]:10.2f}"
                                        for f in ("halo", "halo+copy", "bands", "assemble"))
              + f"   {W * H / (w['halo+copy'] * 1e-3) / 1e6:10.0f}")
    # frames: independent planes per rank (configs[4] shape), no exchange -- every rank runs what one rank runs
    fw, fh = 5760, 3240
    fr = torch.from_numpy(synth_luma(fw, fh)).cuda()
    fo = torch.zeros_like(fr)

    def frame_step():
        ctx.forward_y_dev(fr.data_ptr(), fw, fw * fh, fo.data_ptr(), fw, fw * fh, fw, fh, 1)
    tf = timed(frame_step)
    print(f"# frames workload (configs[4], {fw}x{fh} per rank per step, no exchange): {tf:.4f} ms per frame = "
          f"{fw * fh / (tf * 1e-3) / 1e6:.0f} MPix/s per rank ({S.FLOP_PER_PIXEL * fw * fh / (tf * 1e-3) / PEAK:.3f}); projected "
          + ", ".join(f"N={n}: {n * fw * fh / (tf * 1e-3) / 1e6:.0f}" for n in (1, 2, 4, 8)) + " MPix/s (N x: ranks share nothing)")
    ctx.close()


if __name__ == "__main__":
    main()
