"""Multi-GPU decomposition of the SRCNN Y-channel path (one process per GPU).

The reference has no distributed code; the path shards in
two ways, both derived from its data dependences (src/srcnn.cpp:254-325,
189-243: every output pixel depends on a 13x13 window of ONE input plane and on
the constant weights):

* independent planes (configs[2], configs[4]: batches / streams of frames):
  contiguous frame ranges per rank, NO data-path collective --
  ``frame_range()``;
* one large plane (configs[3]): row stripes.  Each rank owns rows
  [r0, r1) of the input; the fused kernel needs 6 more input rows on each
  interior side (4 for the 9x9 layer + 2 for the 5x5 layer), so neighbours
  swap exactly 6 luma rows (6*W bytes per boundary) point-to-point --
  ``exchange_halo()`` -- and then run ``srcnn_forward_y_rows_dev`` on the
  halo-extended stripe.  Exchanging 6 INPUT rows (46 KB at W=7680) instead of
  2 rows of the 32-channel map (1.97 MB) keeps the layer-2 -> layer-3 hand-off
  inside the fused kernel; image-edge rows are replicated as in the reference,
  stripe-edge rows are real neighbour data.

torch.distributed is plumbing only (NCCL == RCCL on ROCm for GPU tensors, gloo
for the CPU tests).  The compute call is injected (``compute_rows``) so the
CPU tests can drive this logic with the oracle; the product passes
``Context.forward_y_rows_dev`` -- there is no CPU fallback in here.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Tuple

HALO_ROWS = 6   # 4 (9x9 layer) + 0 (1x1) + 2 (5x5 layer)
HALO_SETS = 4   # halo tensors of the one-launch stripe step, used in turn (StripeStep)


def split_range(n: int, parts: int, index: int) -> Tuple[int, int]:
    """Contiguous, balanced [begin, end) of `n` items for part `index` of `parts`
    (the first n % parts parts get one extra item)."""
    if parts <= 0 or not (0 <= index < parts) or n < 0:
        raise ValueError("bad split")
    base, extra = divmod(n, parts)
    begin = index * base + min(index, extra)
    return begin, begin + base + (1 if index < extra else 0)


def frame_range(n_frames: int, world: int, rank: int) -> Tuple[int, int]:
    """Frames [begin, end) of a batch/stream handled by `rank` (no collective)."""
    return split_range(n_frames, world, rank)


def stripe_rows(height: int, world: int, rank: int) -> Tuple[int, int]:
    """Output rows [r0, r1) of a single plane handled by `rank`."""
    return split_range(height, world, rank)


def halo_extent(height: int, r0: int, r1: int) -> Tuple[int, int]:
    """Input rows [s0, s1) the fused kernel needs to produce output rows [r0, r1)."""
    return max(0, r0 - HALO_ROWS), min(height, r1 + HALO_ROWS)


def exchange_halo(stripe, height: int, world: int, rank: int, group=None):
    """Swap HALO_ROWS input rows with the neighbouring ranks.

    stripe : uint8 tensor [r1-r0, W] holding this rank's rows of the plane
             (CPU tensor with gloo, GPU tensor with NCCL/RCCL).
    Returns (ext, s0): tensor with rows [s0, s1) = stripe plus the neighbours'
    halo rows; ranks owning fewer than HALO_ROWS rows are rejected (a halo would
    span more than one neighbour).
    """
    import torch
    import torch.distributed as dist

    r0, r1 = stripe_rows(height, world, rank)
    if stripe.shape[0] != r1 - r0:
        raise ValueError(f"rank {rank}: stripe has {stripe.shape[0]} rows, owns [{r0},{r1})")
    if world > 1 and min(stripe_rows(height, world, k)[1] - stripe_rows(height, world, k)[0]
                         for k in range(world)) < HALO_ROWS:
        raise ValueError("stripes thinner than the halo: use fewer ranks for this plane")
    s0, s1 = halo_extent(height, r0, r1)
    top_n, bot_n = r0 - s0, s1 - r1
    width = stripe.shape[1]
    ext = torch.empty((s1 - s0, width), dtype=stripe.dtype, device=stripe.device)
    ext[top_n:top_n + (r1 - r0)] = stripe
    ops = []
    keep = []
    if top_n:      # rank-1 exists: receive its last rows, send it my first rows
        recv_top = ext[:top_n]
        send_top = stripe[:HALO_ROWS].contiguous()
        keep += [send_top]
        ops += [dist.P2POp(dist.irecv, recv_top, rank - 1, group),
                dist.P2POp(dist.isend, send_top, rank - 1, group)]
    if bot_n:      # rank+1 exists
        recv_bot = ext[top_n + (r1 - r0):]
        send_bot = stripe[-HALO_ROWS:].contiguous()
        keep += [send_bot]
        ops += [dist.P2POp(dist.isend, send_bot, rank + 1, group),
                dist.P2POp(dist.irecv, recv_bot, rank + 1, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    return ext, s0


def exchange_halo_via_host(stripe, height: int, world: int, rank: int, group=None):
    """``exchange_halo`` for a DEVICE stripe over a host-memory transport (gloo): only the 6-row halos are
    staged through the host, the extended stripe is assembled on the device."""
    import torch
    import torch.distributed as dist

    r0, r1 = stripe_rows(height, world, rank)
    if world > 1 and min(stripe_rows(height, world, k)[1] - stripe_rows(height, world, k)[0]
                         for k in range(world)) < HALO_ROWS:
        raise ValueError("stripes thinner than the halo: use fewer ranks for this plane")
    s0, s1 = halo_extent(height, r0, r1)
    top_n, bot_n = r0 - s0, s1 - r1
    ext = torch.empty((s1 - s0, stripe.shape[1]), dtype=stripe.dtype, device=stripe.device)
    ext[top_n:top_n + (r1 - r0)] = stripe
    ops, keep = [], []
    recv_top = recv_bot = None
    if top_n:
        recv_top = torch.empty((top_n, stripe.shape[1]), dtype=stripe.dtype)
        keep.append(stripe[:HALO_ROWS].cpu().contiguous())
        ops += [dist.P2POp(dist.irecv, recv_top, rank - 1, group), dist.P2POp(dist.isend, keep[-1], rank - 1, group)]
    if bot_n:
        recv_bot = torch.empty((bot_n, stripe.shape[1]), dtype=stripe.dtype)
        keep.append(stripe[-HALO_ROWS:].cpu().contiguous())
        ops += [dist.P2POp(dist.isend, keep[-1], rank + 1, group), dist.P2POp(dist.irecv, recv_bot, rank + 1, group)]
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    if top_n:
        ext[:top_n] = recv_top.to(stripe.device)
    if bot_n:
        ext[top_n + (r1 - r0):] = recv_bot.to(stripe.device)
    return ext, s0


def forward_striped(stripe, height: int, world: int, rank: int,
                    compute_rows: Callable, group=None):
    """Row-striped forward pass of ONE plane: halo exchange + one kernel launch.

    compute_rows(ext, s0, out, r0, height, r0, r1) must fill `out`
    (uint8 [r1-r0, W]) with output rows [r0, r1) given input rows starting at
    image row s0 -- ``Context.forward_y_rows_dev`` semantics.
    Returns this rank's output stripe.
    """
    import torch

    r0, r1 = stripe_rows(height, world, rank)
    ext, s0 = exchange_halo(stripe, height, world, rank, group)
    out = torch.empty((r1 - r0, stripe.shape[1]), dtype=torch.uint8, device=stripe.device)
    compute_rows(ext, s0, out, r0, height, r0, r1)
    return out


def gpu_compute_rows(ctx) -> Callable:
    """compute_rows for GPU tensors through the C ABI (srcnn_forward_y_rows_dev).

    The halo rows arrive on torch's current stream (``req.wait()`` of an NCCL/RCCL work
    object orders that stream, not the host), so the context must launch on the same
    stream: ``ctx.set_stream(torch.cuda.current_stream().cuda_stream)`` with a non-default
    stream current, as ``bench.py`` does."""
    def run(ext, s0, out, dst_row0, height, r0, r1):
        if not ext.is_cuda:
            raise RuntimeError("the HIP path needs device tensors (no CPU fallback)")
        w = ext.shape[1]
        ctx.forward_y_rows_dev(ext.data_ptr(), ext.stride(0), s0, out.data_ptr(), out.stride(0),
                               dst_row0, w, height, r0, r1)
        ctx.synchronize()
    return run


def gpu_launch_rows(ctx) -> Callable:
    """Asynchronous launch_rows(src, src_row0, out, dst_row0, height, row_begin, row_end) for GPU tensors
    through the C ABI (srcnn_forward_y_rows_dev) on the context's stream -- no host synchronisation.
    The context must launch on torch's current stream (see ``gpu_compute_rows``)."""
    def run(src, src_row0, out, dst_row0, height, row_begin, row_end):
        if not (src.is_cuda and out.is_cuda):
            raise RuntimeError("the HIP path needs device tensors (no CPU fallback)")
        ctx.forward_y_rows_dev(src.data_ptr(), src.stride(0), src_row0, out.data_ptr(), out.stride(0),
                               dst_row0, src.shape[1], height, row_begin, row_end)
    return run


def gpu_launch_rows_halo(ctx) -> Callable:
    """launch_rows_halo(src, src_row0, top, bot, out, dst_row0, height, row_begin, row_end): the stripe where it lies, its 6
    halo rows either side in tensors of their own (None at an image edge) -- srcnn_forward_y_rows_halo_dev, ONE launch."""
    def run(src, src_row0, top, bot, out, dst_row0, height, row_begin, row_end):
        if not (src.is_cuda and out.is_cuda):
            raise RuntimeError("the HIP path needs device tensors (no CPU fallback)")
        halo = top if top is not None else bot
        ctx.forward_y_rows_halo_dev(src.data_ptr(), src.stride(0), src_row0, src.shape[0],
                                    top.data_ptr() if top is not None else 0, bot.data_ptr() if bot is not None else 0,
                                    halo.stride(0) if halo is not None else src.shape[1],
                                    out.data_ptr(), out.stride(0), dst_row0, src.shape[1], height, row_begin, row_end)
    return run


def band_plan(height: int, world: int, rank: int):
    """Split of this rank's output rows [r0, r1) for the overlapped stripe step:
    (interior [i0, i1), top band or None, bottom band or None).  The interior rows need only the rank's
    own input rows; a band is the HALO_ROWS output rows next to a neighbour, which need its halo rows.
    Returns None when the stripe is too thin to split (then: exchange first, one launch)."""
    r0, r1 = stripe_rows(height, world, rank)
    has_top, has_bot = rank > 0, rank < world - 1
    i0 = r0 + HALO_ROWS if has_top else r0
    i1 = r1 - HALO_ROWS if has_bot else r1
    # a band reads 2 * HALO_ROWS own rows below (above) its HALO_ROWS output rows
    if (has_top or has_bot) and (r1 - r0 < 3 * HALO_ROWS or i1 - i0 < 1):
        return None
    return (i0, i1), ((r0, r0 + HALO_ROWS) if has_top else None), ((r1 - HALO_ROWS, r1) if has_bot else None)


class StripeStep:
    """The row-striped step of ONE plane on this rank, with everything that does not change from step to step
    built ONCE: band buffers ([6 halo rows | 12 own rows], [12 own rows | 6 halo rows]), host staging of the
    halo rows (``via_host``), the send views and the point-to-point op list.  ``step()`` then only posts the
    exchange, launches, copies 2 x 12 rows on the device and waits -- no allocation, no ``.contiguous()`` copy,
    no tensor construction per step (at 8 GPUs a 7680x4320 step is 0.47 ms of kernel per rank).

    stripe / out : uint8 tensors [r1-r0, W], this rank's rows of the input / output plane; the SAME tensors
                   every step.  For a new plane refill ``stripe`` in place on the CURRENT stream and call ``refilled()``:
                   in the one-launch form the exchange runs on a side stream, which must be ordered behind the refill
                   (the other forms post on the current stream and need nothing).  ``stripe`` must be contiguous: its
                   first / last 6 rows are sent as they lie.
    launch_rows_halo : given, the step is ONE launch on the stripe where it lies with the received halo rows in two small
                   tensors of their own (``gpu_launch_rows_halo``; the float32 MFMA kernel picks the buffer a row lives in).
                   The halo tensors come in HALO_SETS sets used in turn, and on the GPU the exchange is posted on a side
                   stream: the exchange of a step runs while the kernels of the steps before it still read the other sets.
                   This is the default form of ``bench.py --workload stripe``: the band form below pays ~55 us for its two
                   6-row band launches to hide a ~15 us exchange (profiles/r04/stripe_projection.txt).
    overlap      : (without launch_rows_halo) post the 6-row exchange, launch the INTERIOR rows (which need no halo)
                   while it is in flight, then the two 6-row edge bands; otherwise (or for stripes thinner than 18 rows)
                   exchange into a persistent [halo | stripe | halo] buffer and launch once.  Same bytes either
                   way: any partition of the rows computes the same plane (tests/test_sharding_gloo.py).
    via_host     : stage the halo rows through host memory (gloo group; smoke tests on a shared GPU).
    """

    def __init__(self, stripe, out, height: int, world: int, rank: int, launch_rows: Callable, group=None,
                 overlap: bool = True, via_host: bool = False, launch_rows_halo: Optional[Callable] = None):
        import torch
        import torch.distributed as dist

        self.stripe, self.out, self.height, self.world, self.rank = stripe, out, height, world, rank
        self.launch_rows, self.group, self.via_host = launch_rows, group, via_host
        self.launch_rows_halo, self.n_steps = launch_rows_halo, 0
        self._refill = None            # event behind the stripe's last write, consumed by the next exchange (refilled())
        self.r0, self.r1 = stripe_rows(height, world, rank)
        if stripe.shape[0] != self.r1 - self.r0 or out.shape != stripe.shape:
            raise ValueError(f"rank {rank}: stripe has {stripe.shape[0]} rows, owns [{self.r0},{self.r1})")
        if world > 1 and not stripe.is_contiguous():
            raise ValueError("stripe must be contiguous (its edge rows are sent in place)")
        if world > 1 and min(b - a for a, b in (stripe_rows(height, world, k) for k in range(world))) < HALO_ROWS:
            raise ValueError("stripes thinner than the halo: use fewer ranks for this plane")
        self.plan = band_plan(height, world, rank) if (overlap and world > 1) else None
        self.ops, self.top_buf, self.bot_buf, self.ext = [], None, None, None
        self.stage = []                # (device view, host tensor): sends copied out before / receives copied in after
        if world == 1:
            return
        dev, width, rows = stripe.device, stripe.shape[1], self.r1 - self.r0
        has_top, has_bot = rank > 0, rank < world - 1
        self.s0, self.s1 = halo_extent(height, self.r0, self.r1)
        if launch_rows_halo is not None:
            # HALO_SETS sets of halo tensors and of everything that refers to them
            self.sets = []
            for _ in range(HALO_SETS):
                top = torch.empty((HALO_ROWS, width), dtype=stripe.dtype, device=dev) if has_top else None
                bot = torch.empty((HALO_ROWS, width), dtype=stripe.dtype, device=dev) if has_bot else None
                send_stage, recv_stage, ops = [], [], []

                def endpoint(view, sending, send_stage=send_stage, recv_stage=recv_stage):
                    if not via_host:
                        return view
                    host = torch.empty(view.shape, dtype=view.dtype)
                    (send_stage if sending else recv_stage).append((view, host))
                    return host
                if has_top:
                    ops += [dist.P2POp(dist.irecv, endpoint(top, False), rank - 1, group),
                            dist.P2POp(dist.isend, endpoint(stripe[:HALO_ROWS], True), rank - 1, group)]
                if has_bot:
                    ops += [dist.P2POp(dist.isend, endpoint(stripe[rows - HALO_ROWS:], True), rank + 1, group),
                            dist.P2POp(dist.irecv, endpoint(bot, False), rank + 1, group)]
                self.sets.append(dict(top=top, bot=bot, ops=ops, send_stage=send_stage, recv_stage=recv_stage, free=None))
            self.side = torch.cuda.Stream(device=dev) if (stripe.is_cuda and not via_host) else None
            return
        if self.plan is None:          # one launch on [halo | stripe | halo]
            self.ext = torch.empty((self.s1 - self.s0, width), dtype=stripe.dtype, device=dev)
            recv_top = self.ext[:HALO_ROWS] if has_top else None
            recv_bot = self.ext[self.ext.shape[0] - HALO_ROWS:] if has_bot else None
        else:
            if has_top:
                self.top_buf = torch.empty((3 * HALO_ROWS, width), dtype=stripe.dtype, device=dev)
            if has_bot:
                self.bot_buf = torch.empty((3 * HALO_ROWS, width), dtype=stripe.dtype, device=dev)
            recv_top = self.top_buf[:HALO_ROWS] if has_top else None
            recv_bot = self.bot_buf[2 * HALO_ROWS:] if has_bot else None
        send_top, send_bot = stripe[:HALO_ROWS], stripe[rows - HALO_ROWS:]
        self._send_stage, self._recv_stage = [], []

        def endpoint(view, sending):
            if not via_host:
                return view
            host = torch.empty(view.shape, dtype=view.dtype)
            (self._send_stage if sending else self._recv_stage).append((view, host))
            return host
        if has_top:      # rank-1: receive its last rows, send it my first rows
            self.ops += [dist.P2POp(dist.irecv, endpoint(recv_top, False), rank - 1, group),
                         dist.P2POp(dist.isend, endpoint(send_top, True), rank - 1, group)]
        if has_bot:
            self.ops += [dist.P2POp(dist.isend, endpoint(send_bot, True), rank + 1, group),
                         dist.P2POp(dist.irecv, endpoint(recv_bot, False), rank + 1, group)]

    def _post(self):
        import torch.distributed as dist
        for view, host in self._send_stage:
            host.copy_(view)
        return dist.batch_isend_irecv(self.ops)

    def _land(self, reqs):
        for req in reqs:
            req.wait()
        for view, host in self._recv_stage:
            view.copy_(host)

    def step(self):
        """One step, asynchronous on the caller's stream (the context must launch on torch's current stream)."""
        stripe, out, h, r0, r1 = self.stripe, self.out, self.height, self.r0, self.r1
        if self.world == 1:
            self.launch_rows(stripe, 0, out, 0, h, 0, h)
            return out
        if self.launch_rows_halo is not None:
            return self._step_halo()
        if self.plan is None:
            reqs = self._post()
            top_n = r0 - self.s0
            self.ext[top_n:top_n + (r1 - r0)] = stripe
            self._land(reqs)
            self.launch_rows(self.ext, self.s0, out, r0, h, r0, r1)
            return out
        (i0, i1), top, bot = self.plan
        reqs = self._post()
        self.launch_rows(stripe, r0, out, r0, h, i0, i1)          # overlaps the exchange
        if top:
            self.top_buf[HALO_ROWS:] = stripe[:2 * HALO_ROWS]
        if bot:
            self.bot_buf[:2 * HALO_ROWS] = stripe[stripe.shape[0] - 2 * HALO_ROWS:]
        self._land(reqs)
        if top:
            self.launch_rows(self.top_buf, r0 - HALO_ROWS, out, r0, h, top[0], top[1])
        if bot:
            self.launch_rows(self.bot_buf, r1 - 2 * HALO_ROWS, out, r0, h, bot[0], bot[1])
        return out

    def refilled(self):
        """Call after writing a NEW plane's rows into ``stripe`` on the current stream: the next step's exchange, which
        sends the stripe's edge rows from a side stream, waits for those writes (once; steps on an unchanged stripe stay
        unserialised).  The kernels of earlier steps that still read the stripe are ordered by the stream itself."""
        import torch
        if self.stripe.is_cuda:
            self._refill = torch.cuda.current_stream().record_event()

    def _step_halo(self):
        """ONE launch: exchange into this step's halo set, then the whole stripe (see the class docstring)."""
        import torch
        import torch.distributed as dist

        st = self.sets[self.n_steps % HALO_SETS]
        self.n_steps += 1
        for view, host in st["send_stage"]:
            host.copy_(view)
        if self.side is not None:
            if self._refill is not None:           # the edge rows about to be sent were written on the current stream
                self.side.wait_event(self._refill)
                self._refill = None
            # the exchange waits only for the launch that last read THIS set (HALO_SETS steps ago), not for the previous steps:
            # its kernels need a free compute unit and find one in the tail of an earlier step's launch (same-box timing,
            # tools/stripe_projection.py --diag: with two sets the hand-over cost 34 us per step, most of it this wait)
            if st["free"] is not None:
                self.side.wait_event(st["free"])
            with torch.cuda.stream(self.side):
                reqs = dist.batch_isend_irecv(st["ops"])
        else:
            reqs = dist.batch_isend_irecv(st["ops"])
        for req in reqs:
            req.wait()                 # NCCL / RCCL: orders the CURRENT stream behind the exchange, not the host
        for view, host in st["recv_stage"]:
            view.copy_(host)
        self.launch_rows_halo(self.stripe, self.r0, st["top"], st["bot"], self.out, self.r0, self.height, self.r0, self.r1)
        if self.side is not None:
            st["free"] = torch.cuda.current_stream().record_event()
        return self.out

    __call__ = step


class PeerStripeStep:
    """The row-striped step of ONE plane on this rank WITHOUT a per-step exchange: every rank maps its neighbours' stripes into
    its own address space once (HIP IPC handles of the allocations, exchanged over the control-plane group) and its one launch
    reads their 6 edge rows where they lie -- between GPUs that is 46 KB of loads over xGMI in the prologue of the workgroups at
    the stripe's edges (``srcnn_forward_y_rows_halo_dev`` takes the mapped addresses as its halo pointers).  What the one-process
    host does with peer access (``srcnn_forward_y_striped_dev``), for one process per GPU.

    STREAMS OF PLANES.  The stripe lives in TWO allocations of its own used in turn (``Context.dev_alloc``: an IPC handle names a
    whole allocation; both are exported and mapped once).  ``upload(rows)`` writes plane g into allocation g % 2 -- the one no
    neighbour's step on plane g - 1 reads -- through a second context, so the copy does not queue behind the kernels; ``step()``
    runs the latest plane.  What has to be ordered ACROSS ranks is settled neighbour to neighbour, not by a barrier: a rank tells
    its two neighbours "plane g is uploaded" after the copy and waits for theirs before the first step on plane g (their edge
    rows are read by its launch); before it overwrites allocation g % 2 it tells them "my steps on plane g - 2 are done" and waits
    for theirs (their launches read its edge rows).  One-word gloo messages with tags; a plane stepped repeatedly (the bench's
    resident input) exchanges nothing after its first step.  "Done" is an event on the stream the KERNELS run on: the stream
    given to ``ctx.set_stream`` (upload(g) then overlaps the steps on plane g - 1), or, when the context runs on its own stream,
    a wait for the context (correct, no overlap) -- the caller's current torch stream plays no part.
    """

    def __init__(self, ctx, stripe_rows_np, out, height: int, world: int, rank: int, group=None):
        import torch
        import torch.distributed as dist

        self.ctx, self.out, self.height, self.world, self.rank, self.group = ctx, out, height, world, rank, group
        self.r0, self.r1 = stripe_rows(height, world, rank)
        self.width = int(stripe_rows_np.shape[1])
        if stripe_rows_np.shape[0] != self.r1 - self.r0:
            raise ValueError(f"rank {rank}: stripe has {stripe_rows_np.shape[0]} rows, owns [{self.r0},{self.r1})")
        if world > 1 and min(b - a for a, b in (stripe_rows(height, world, k) for k in range(world))) < HALO_ROWS:
            raise ValueError("stripes thinner than the halo: use fewer ranks for this plane")
        nbytes = (self.r1 - self.r0) * self.width
        self.d_stripes = [ctx.dev_alloc(nbytes), ctx.dev_alloc(nbytes)]
        self.d_stripe = self.d_stripes[0]                  # (the allocation of the plane being stepped; kept for older callers)
        self.top, self.bot = [0, 0], [0, 0]
        self._mapped, self._sends, self._copy_ctx = [], [], None
        self.nbrs = [n for n in (rank - 1, rank + 1) if 0 <= n < world]
        self.gen, self._ready_seen = 0, -1                 # planes uploaded so far; the last plane whose neighbours are known ready
        self._done_ev = [None, None]                       # behind this rank's last step on the plane in allocation b
        self._done_sent = -1
        self._torch, self._dist = torch, dist
        if world > 1:
            # Collective-safe: a rank whose export / mapping fails (no IPC between these two devices, a runtime without dmabuf
            # IPC) must not leave the others waiting in a collective -- every rank reports, then all succeed or all raise.
            err, handle = None, None
            try:
                handle = [ctx.ipc_export(d) for d in self.d_stripes]
            except Exception as e:          # noqa: BLE001 -- reported to every rank below
                err = f"rank {rank}: export: {e}"
            handles = [None] * world
            dist.all_gather_object(handles, (handle, err), group=group)
            if all(h[1] is None for h in handles):
                try:
                    for b in range(2):
                        if rank > 0:
                            a0, a1 = stripe_rows(height, world, rank - 1)
                            base = ctx.ipc_open(handles[rank - 1][0][b])
                            self._mapped.append(base)
                            self.top[b] = base + (a1 - a0 - HALO_ROWS) * self.width      # the upper neighbour's last 6 rows
                        if rank < world - 1:
                            base = ctx.ipc_open(handles[rank + 1][0][b])
                            self._mapped.append(base)
                            self.bot[b] = base                                            # the lower neighbour's first 6 rows
                except Exception as e:      # noqa: BLE001
                    err = f"rank {rank}: open: {e}"
            errs = [None] * world
            dist.all_gather_object(errs, err, group=group)
            bad = [e for e in errs if e] + [h[1] for h in handles if h[1]]
            if bad:
                self.close(collective=False)
                raise RuntimeError("HIP IPC mapping of the neighbours' stripes failed: " + "; ".join(sorted(set(bad))))
        self.upload(stripe_rows_np)

    # ---- neighbour-to-neighbour ordering (gloo, one word per message) ----
    def _tell(self, kind: int, g: int):
        for n in self.nbrs:
            self._sends.append(self._dist.isend(self._torch.tensor([g], dtype=self._torch.int64), dst=n, group=self.group, tag=2 * g + kind))
        self._sends = [w for w in self._sends if not w.is_completed()]

    def _hear(self, kind: int, g: int):
        for n in self.nbrs:
            t = self._torch.zeros(1, dtype=self._torch.int64)
            self._dist.recv(t, src=n, group=self.group, tag=2 * g + kind)
            if int(t.item()) != g:
                raise RuntimeError(f"rank {self.rank}: neighbour {n} answered plane {int(t.item())}, expected {g}")

    def upload(self, stripe_rows_np):
        """This rank's rows of the NEXT plane, into the allocation no step in flight reads.  No barrier."""
        import srcnn_cpp_amd as S
        g = self.gen
        b = g % 2
        if self.world > 1 and g >= 2:
            # allocation b held plane g - 2: this rank's steps on it must have run, and the neighbours' (they read its edge rows)
            if self._done_ev[b] == "ctx":
                self.ctx.synchronize()
            elif self._done_ev[b] is not None:
                self._done_ev[b].synchronize()
            if self._done_sent < g - 2:
                self._tell(1, g - 2)
                self._done_sent = g - 2
            self._hear(1, g - 2)
        if self._copy_ctx is None:          # a context of its own for the copies: dev_upload waits for ITS stream only
            self._copy_ctx = S.Context(self.ctx.device) if hasattr(self.ctx, "device") else self.ctx
        self._copy_ctx.dev_upload(self.d_stripes[b], stripe_rows_np)
        if self.world > 1:
            self._tell(0, g)
        self.gen = g + 1
        self.d_stripe = self.d_stripes[b]

    def step(self):
        """One step on the latest plane: ONE launch, asynchronous on the context's stream."""
        g = self.gen - 1
        b = g % 2
        if self.world > 1 and self._ready_seen < g:
            self._hear(0, g)                 # the neighbours' rows of plane g are in place (first step on this plane only)
            self._ready_seen = g
        out, w = self.out, self.width
        self.ctx.forward_y_rows_halo_dev(self.d_stripes[b], w, self.r0, self.r1 - self.r0, self.top[b], self.bot[b], w,
                                         out.data_ptr(), out.stride(0), self.r0, w, self.height, self.r0, self.r1)
        if self.world > 1:
            # "my steps on this plane are done" must cover the KERNEL, which runs on the context's stream -- not on whatever stream
            # torch calls current (advisor, round 5).  A caller that gave the context a framework stream (ctx.set_stream) gets an
            # event on exactly that stream; with the context's own stream, which torch cannot name, upload() waits for the context.
            sp = getattr(self.ctx, "stream_ptr", 0)
            self._done_ev[b] = self._torch.cuda.ExternalStream(sp).record_event() if sp else "ctx"
        return out

    __call__ = step

    def close(self, collective: bool = True):
        import torch.distributed as dist
        self.ctx.synchronize()
        for wk in self._sends:
            wk.wait()
        self._sends = []
        collective = collective and self.world > 1
        if collective:
            dist.barrier(group=self.group)  # nobody reads this rank's rows any more
        for base in self._mapped:
            self.ctx.ipc_close(base)
        self._mapped = []
        if collective:
            dist.barrier(group=self.group)  # every mapping of this rank's allocations is gone before they are freed
        for d in self.d_stripes:
            if d:
                self.ctx.dev_free(d)
        self.d_stripes = [0, 0]
        self.d_stripe = 0
        if self._copy_ctx is not None and self._copy_ctx is not self.ctx:
            self._copy_ctx.close()
        self._copy_ctx = None


def forward_striped_launch(stripe, out, height: int, world: int, rank: int, launch_rows: Callable,
                           group=None, overlap: bool = True, via_host: bool = False, launch_rows_halo: Optional[Callable] = None):
    """One row-striped step of ONE plane (a ``StripeStep`` built and run once; callers that step repeatedly keep
    the ``StripeStep``).  Returns ``out``."""
    return StripeStep(stripe, out, height, world, rank, launch_rows, group=group, overlap=overlap, via_host=via_host,
                      launch_rows_halo=launch_rows_halo).step()


def gather_stripes(out, height: int, world: int, rank: int, dst: int = 0, group=None):
    """Collect the output stripes on `dst` (verification / file output only; the
    data path itself needs no collective).  Returns the full plane on dst, None elsewhere."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return out
    sizes = [stripe_rows(height, world, k) for k in range(world)]
    pad = max(b - a for a, b in sizes)
    buf = torch.zeros((pad, out.shape[1]), dtype=out.dtype, device=out.device)
    buf[:out.shape[0]] = out
    gathered: Optional[List] = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, gathered, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([g[:b - a] for g, (a, b) in zip(gathered, sizes)], dim=0)
