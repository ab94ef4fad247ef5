// srcnn_multi.cpp -- several GPUs driven from ONE host process: a row-striped plane whose stripes read their
// neighbours' 6 edge rows over xGMI (or copies of them), frame ranges over contexts; one persistent host thread per context.
#include "srcnn_ctx.h"

using namespace srcnn;
using namespace srcnn::host;

extern "C" {

/* ------------------------- several GPUs from one host process ---------------- */

int srcnn_stripe_rows(int height, int n_parts, int index, int *row_begin, int *row_end)
{
    if (height < 0 || n_parts <= 0 || index < 0 || index >= n_parts || !row_begin || !row_end) return SRCNN_ERR_INVALID;
    const int base = height / n_parts, extra = height % n_parts;
    *row_begin = index * base + std::min(index, extra);
    *row_end = *row_begin + base + (index < extra ? 1 : 0);
    return SRCNN_OK;
}

}  // extern "C"

namespace {

constexpr int kHalo = kHaloRows;

int check_ctx_set(srcnn_ctx *const *ctxs, int n_ctx)
{
    if (!ctxs || n_ctx <= 0) return SRCNN_ERR_INVALID;
    for (int k = 0; k < n_ctx; ++k) {
        if (!ctxs[k]) return SRCNN_ERR_INVALID;
        if (!has_model(ctxs[k])) return fail(ctxs[k], SRCNN_ERR_STATE, "%s", kNoModel);
        for (int j = 0; j < k; ++j)
            if (ctxs[j] == ctxs[k]) return fail(ctxs[k], SRCNN_ERR_INVALID, "the same context appears twice");
    }
    return SRCNN_OK;
}

// rows of a stripe held by another context (possibly on another device) -> this context's buffer, on `st`
hipError_t copy_rows_between(srcnn_ctx *to, uint8_t *dst, size_t dst_stride, const srcnn_ctx *from, const uint8_t *src,
                             size_t src_stride, int width, int rows, hipStream_t st)
{
    if (from->device == to->device)
        return launch_copy_rows(dst, (long)dst_stride, src, (long)src_stride, width, rows, st);
    if (dst_stride == (size_t)width && src_stride == (size_t)width)
        return hipMemcpyPeerAsync(dst, to->device, src, from->device, (size_t)width * rows, st);
    for (int r = 0; r < rows; ++r) {
        const hipError_t e = hipMemcpyPeerAsync(dst + (size_t)r * dst_stride, to->device, src + (size_t)r * src_stride,
                                                from->device, (size_t)width, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// The streams and events of the striped step (created once per context) and the links to the neighbouring devices: peer access
// is asked for ONCE PER DEVICE and the answer kept -- a refused link still works (hipMemcpyPeerAsync then stages through host
// memory) but is not the xGMI path BASELINE configs[3] names, so the context says so (srcnn_halo_transport(),
// srcnn_last_error()).  The neighbours are looked at on EVERY call: a context may serve in another set later.
int stripe_setup(srcnn_ctx *const *ctxs, int n_ctx, int k)
{
    srcnn_ctx *c = ctxs[k];
    if (!c->halo_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->halo_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipEventCreateWithFlags(&c->halo_ready, hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->bands_done, hipEventDisableTiming));
        for (int i = 0; i < srcnn_ctx::kHaloSets; ++i) HIP_TRY(c, hipEventCreateWithFlags(&c->halo_free[i], hipEventDisableTiming));
    }
    int transport = 1;
    static const char *env_staged = SRCNN_DEBUG_ENV("SRCNN_DEBUG_HALO_STAGED");      // test knob: take the no-peer-access path
    if (env_staged && std::atoi(env_staged)) transport = 3;
    for (int n : {k - 1, k + 1}) {
        if (n < 0 || n >= n_ctx || ctxs[n]->device == c->device) continue;
        const int dev = ctxs[n]->device;
        if ((int)c->peer_state.size() <= dev) c->peer_state.resize((size_t)dev + 1, 0);
        if (c->peer_state[(size_t)dev] == 0) {
            int can = 0;
            hipError_t e = hipDeviceCanAccessPeer(&can, c->device, dev);
            if (e == hipSuccess && can) {
                e = hipDeviceEnablePeerAccess(dev, 0);
                if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
            }
            if (e == hipSuccess && can) {
                c->peer_state[(size_t)dev] = 2;
            } else {
                (void)hipGetLastError();
                c->peer_state[(size_t)dev] = 3;
                (void)fail(c, SRCNN_OK, "row stripes: device %d has no peer access to device %d (%s): halo rows are staged through host "
                                        "memory, not read over xGMI", c->device, dev,
                           e == hipSuccess ? "hipDeviceCanAccessPeer says no" : hipGetErrorString(e));
            }
        }
        transport = std::max(transport, c->peer_state[(size_t)dev]);
    }
    c->halo_transport = transport;
    return SRCNN_OK;
}

// One context's part of the striped step.  Runs on its own host thread (one thread per device).
//
// float32 MFMA kernel (SRCNN_MODE_MFMA / REFBYTES): ONE launch per stripe through srcnn_forward_y_rows_halo_dev -- the kernel
// picks the buffer a Y row lives in with a scalar select.  With peer access (xGMI) or neighbours on the same device the halo
// "buffers" ARE the neighbours' stripes: the 60 workgroups at a stripe edge load 46 KB of the neighbour's edge rows in their
// prologue, straight over the link -- no copy, no second stream, no event, nothing on the critical path but the launch.
// A link that refuses peer access gets copies (staged through the host by the runtime) into halo buffers of this device on
// a second stream, kHaloSets sets in turn so that the copies of a step overlap the kernels of the steps before it.
// The earlier form -- interior rows first, then two 6-row edge bands from [6 halo | 12 own] buffers: three strip launches,
// up to three seam launches, four row copies -- paid 30-55 us for the band launches to hide a 15 us copy
// (profiles/r04/stripe_projection.txt); it remains for the kernels that read one buffer only (split-f16 modes).
int striped_step(srcnn_ctx *const *ctxs, int n_ctx, int k, const uint8_t *const *d_stripes, size_t stripe_stride,
                 uint8_t *const *d_out, size_t out_stride, int width, int height)
{
    srcnn_ctx *c = ctxs[k];
    BIND(c);
    int rc, r0, r1, a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    srcnn_stripe_rows(height, n_ctx, k, &r0, &r1);
    const bool has_top = k > 0, has_bot = k < n_ctx - 1;
    if (has_top) srcnn_stripe_rows(height, n_ctx, k - 1, &a0, &a1);
    if (has_bot) srcnn_stripe_rows(height, n_ctx, k + 1, &b0, &b1);
    if (!has_top && !has_bot)
        return srcnn_forward_y_rows_dev(c, d_stripes[k], stripe_stride, 0, d_out[k], out_stride, 0, width, height, 0, height);
    if ((rc = stripe_setup(ctxs, n_ctx, k))) return rc;
    if (c->mode == SRCNN_MODE_MFMA || c->mode == SRCNN_MODE_REFBYTES) {
        const uint8_t *nb_top = has_top ? d_stripes[k - 1] + (size_t)(a1 - a0 - kHalo) * stripe_stride : nullptr;
        const uint8_t *nb_bot = has_bot ? d_stripes[k + 1] : nullptr;
        if (c->halo_transport != 3)       // the neighbours' rows where they lie (same device, or peer-mapped over xGMI)
            return srcnn_forward_y_rows_halo_dev(c, d_stripes[k], stripe_stride, r0, r1 - r0, nb_top, nb_bot, stripe_stride,
                                                 d_out[k], out_stride, r0, width, height, r0, r1);
        const int set = (int)(c->stripe_steps++ % srcnn_ctx::kHaloSets);
        const size_t halo_bytes = (size_t)kHalo * width;
        if ((rc = reserve(c, c->halo_top[set], halo_bytes))) return rc;
        if ((rc = reserve(c, c->halo_bot[set], halo_bytes))) return rc;
        uint8_t *top = static_cast<uint8_t *>(c->halo_top[set].p), *bot = static_cast<uint8_t *>(c->halo_bot[set].p);
        if (c->halo_free_set[set]) HIP_TRY(c, hipStreamWaitEvent(c->halo_stream, c->halo_free[set], 0));
        if (has_top) HIP_TRY(c, copy_rows_between(c, top, width, ctxs[k - 1], nb_top, stripe_stride, width, kHalo, c->halo_stream));
        if (has_bot) HIP_TRY(c, copy_rows_between(c, bot, width, ctxs[k + 1], nb_bot, stripe_stride, width, kHalo, c->halo_stream));
        HIP_TRY(c, hipEventRecord(c->halo_ready, c->halo_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->halo_ready, 0));
        if ((rc = srcnn_forward_y_rows_halo_dev(c, d_stripes[k], stripe_stride, r0, r1 - r0, has_top ? top : nullptr,
                                                has_bot ? bot : nullptr, (size_t)width, d_out[k], out_stride, r0, width, height, r0, r1)))
            return rc;
        HIP_TRY(c, hipEventRecord(c->halo_free[set], c->stream));
        c->halo_free_set[set] = true;
        return SRCNN_OK;
    }
    const size_t band_bytes = (size_t)3 * kHalo * width;
    if ((rc = reserve(c, c->band_top, band_bytes))) return rc;
    if ((rc = reserve(c, c->band_bot, band_bytes))) return rc;
    uint8_t *top = static_cast<uint8_t *>(c->band_top.p), *bot = static_cast<uint8_t *>(c->band_bot.p);
    const int rows = r1 - r0;
    // the band inputs of the previous step may still be read by its band launches
    if (c->bands_pending) HIP_TRY(c, hipStreamWaitEvent(c->halo_stream, c->bands_done, 0));
    if (rows < 3 * kHalo) {
        // stripe too thin to split: assemble [halo | stripe | halo] and launch once
        const int s0 = has_top ? r0 - kHalo : r0, s1 = has_bot ? r1 + kHalo : r1;
        if ((rc = reserve(c, c->stripe_ext, (size_t)(s1 - s0) * width))) return rc;
        uint8_t *ext = static_cast<uint8_t *>(c->stripe_ext.p);
        if (has_top)
            HIP_TRY(c, copy_rows_between(c, ext, width, ctxs[k - 1], d_stripes[k - 1] + (size_t)(a1 - a0 - kHalo) * stripe_stride,
                                         stripe_stride, width, kHalo, c->halo_stream));
        HIP_TRY(c, launch_copy_rows(ext + (size_t)(r0 - s0) * width, width, d_stripes[k], (long)stripe_stride, width, rows, c->halo_stream));
        if (has_bot)
            HIP_TRY(c, copy_rows_between(c, ext + (size_t)(r1 - s0) * width, width, ctxs[k + 1], d_stripes[k + 1], stripe_stride,
                                         width, kHalo, c->halo_stream));
        HIP_TRY(c, hipEventRecord(c->halo_ready, c->halo_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->halo_ready, 0));
        rc = srcnn_forward_y_rows_dev(c, ext, width, s0, d_out[k], out_stride, r0, width, height, r0, r1);
        if (rc) return rc;
        HIP_TRY(c, hipEventRecord(c->bands_done, c->stream));
        c->bands_pending = true;
        return SRCNN_OK;
    }
    // halo stream: [6 rows of the upper neighbour | my first 12 rows] and [my last 12 rows | 6 rows of the lower one]
    if (has_top) {
        HIP_TRY(c, copy_rows_between(c, top, width, ctxs[k - 1], d_stripes[k - 1] + (size_t)(a1 - a0 - kHalo) * stripe_stride,
                                     stripe_stride, width, kHalo, c->halo_stream));
        HIP_TRY(c, launch_copy_rows(top + (size_t)kHalo * width, width, d_stripes[k], (long)stripe_stride, width, 2 * kHalo, c->halo_stream));
    }
    if (has_bot) {
        HIP_TRY(c, launch_copy_rows(bot, width, d_stripes[k] + (size_t)(rows - 2 * kHalo) * stripe_stride, (long)stripe_stride, width,
                                    2 * kHalo, c->halo_stream));
        HIP_TRY(c, copy_rows_between(c, bot + (size_t)2 * kHalo * width, width, ctxs[k + 1], d_stripes[k + 1], stripe_stride,
                                     width, kHalo, c->halo_stream));
    }
    HIP_TRY(c, hipEventRecord(c->halo_ready, c->halo_stream));
    // main stream: the interior rows need no halo and run while the copies are in flight
    const int i0 = has_top ? r0 + kHalo : r0, i1 = has_bot ? r1 - kHalo : r1;
    if ((rc = srcnn_forward_y_rows_dev(c, d_stripes[k], stripe_stride, r0, d_out[k], out_stride, r0, width, height, i0, i1)))
        return rc;
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->halo_ready, 0));
    if (has_top && (rc = srcnn_forward_y_rows_dev(c, top, width, r0 - kHalo, d_out[k], out_stride, r0, width, height, r0, i0)))
        return rc;
    if (has_bot && (rc = srcnn_forward_y_rows_dev(c, bot, width, r1 - 2 * kHalo, d_out[k], out_stride, r0, width, height, i1, r1)))
        return rc;
    HIP_TRY(c, hipEventRecord(c->bands_done, c->stream));
    c->bands_pending = true;
    return SRCNN_OK;
}

// fn(k) for every context of the set, context k > 0 on the k-th persistent worker thread of ctxs[0]'s pool
template <typename Fn>
int run_per_context(srcnn_ctx *const *ctxs, int n_ctx, Fn fn)
{
    if (n_ctx == 1) return fn(0);
    if (!ctxs[0]->pool) ctxs[0]->pool.reset(new (std::nothrow) WorkerPool());
    if (!ctxs[0]->pool) return fail(ctxs[0], SRCNN_ERR_NOMEM, "worker pool");
    return ctxs[0]->pool->run(n_ctx, fn);
}
// ... for tasks that wait for each other: `on_incomplete` when some of them could not be started
template <typename Fn, typename Abort>
int run_per_context(srcnn_ctx *const *ctxs, int n_ctx, Fn fn, Abort on_incomplete)
{
    if (n_ctx == 1) return fn(0);
    if (!ctxs[0]->pool) ctxs[0]->pool.reset(new (std::nothrow) WorkerPool());
    if (!ctxs[0]->pool) return fail(ctxs[0], SRCNN_ERR_NOMEM, "worker pool");
    return ctxs[0]->pool->run(n_ctx, fn, SRCNN_ERR_NOMEM, on_incomplete);
}

}  // namespace

extern "C" {

int srcnn_forward_y_striped_dev(srcnn_ctx *const *ctxs, int n_ctx, const uint8_t *const *d_stripes, size_t stripe_stride,
                                uint8_t *const *d_out, size_t out_stride, int width, int height)
{
    int rc = check_ctx_set(ctxs, n_ctx);
    if (rc) return rc;
    if (!d_stripes || !d_out || width <= 0 || height <= 0 || stripe_stride < (size_t)width || out_stride < (size_t)width)
        return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_striped_dev: bad arguments");
    if (n_ctx > 1 && height / n_ctx < kHalo)
        return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_striped_dev: %d rows over %d contexts leaves stripes thinner than the "
                                                "%d-row halo", height, n_ctx, kHalo);
    for (int k = 0; k < n_ctx; ++k) {
        if (!d_stripes[k] || !d_out[k]) return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_striped_dev: null stripe %d", k);
        if (ctxs[k]->mode == SRCNN_MODE_EXACT) return fail(ctxs[k], SRCNN_ERR_STATE, "row stripes are MFMA-mode only");
    }
    return run_per_context(ctxs, n_ctx, [&](int k) {
        return striped_step(ctxs, n_ctx, k, d_stripes, stripe_stride, d_out, out_stride, width, height);
    });
}

int srcnn_forward_y_striped(srcnn_ctx *const *ctxs, int n_ctx, const uint8_t *src, size_t src_stride, uint8_t *dst,
                            size_t dst_stride, int width, int height)
{
    int rc = check_ctx_set(ctxs, n_ctx);
    if (rc) return rc;
    if (bad_plane(src, src_stride, width, height) || bad_plane(dst, dst_stride, width, height))
        return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_striped: bad plane geometry");
    if (n_ctx > 1 && height / n_ctx < kHalo)
        return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_striped: stripes thinner than the %d-row halo", kHalo);
    std::vector<const uint8_t *> d_in((size_t)n_ctx);
    std::vector<uint8_t *> d_res((size_t)n_ctx);
    // phase 1: every device receives ITS rows only (the halo rows then travel device to device)
    rc = run_per_context(ctxs, n_ctx, [&](int k) -> int {
        srcnn_ctx *c = ctxs[k];
        BIND(c);
        int r, r0, r1;
        srcnn_stripe_rows(height, n_ctx, k, &r0, &r1);
        const size_t n = (size_t)(r1 - r0) * width;
        if ((r = reserve(c, c->in_u8, n))) return r;
        if ((r = reserve(c, c->out_u8, n))) return r;
        HIP_TRY(c, hipMemcpy2DAsync(c->in_u8.p, width, src + (size_t)r0 * src_stride, src_stride, width, r1 - r0,
                                    hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        d_in[(size_t)k] = static_cast<const uint8_t *>(c->in_u8.p);
        d_res[(size_t)k] = static_cast<uint8_t *>(c->out_u8.p);
        return SRCNN_OK;
    });
    if (rc) return rc;
    // phase 2: halo copies + interior rows + edge bands, then each device returns its rows
    return run_per_context(ctxs, n_ctx, [&](int k) -> int {
        srcnn_ctx *c = ctxs[k];
        BIND(c);
        int r, r0, r1;
        srcnn_stripe_rows(height, n_ctx, k, &r0, &r1);
        if ((r = striped_step(ctxs, n_ctx, k, d_in.data(), width, d_res.data(), width, width, height))) return r;
        HIP_TRY(c, hipMemcpy2DAsync(dst + (size_t)r0 * dst_stride, dst_stride, c->out_u8.p, width, width, r1 - r0,
                                    hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return SRCNN_OK;
    });
}

/* A STREAM of planes, each row-striped over the contexts: a pipeline, not n_planes one-shot calls.  Context k keeps two stripe
 * buffers; while the kernels of plane p run, its rows of plane p + 1 go up on a copy stream of their own and its rows of plane
 * p - 1 come back on another.  What has to be ordered ACROSS contexts goes through events, never through the host: the launch
 * of plane p on context k waits for the uploads of plane p on k - 1, k, k + 1 (it reads their edge rows where they lie), and
 * the upload of plane p + 2 into the buffer plane p used waits for the kernels of plane p on k - 1, k, k + 1 (theirs read its
 * edge rows).  One host thread per context; a thread tells a stream to wait for another context's event only once that event
 * has been RECORDED (the contexts' generation counters), and spins for nothing else. */
int srcnn_forward_y_striped_frames(srcnn_ctx *const *ctxs, int n_ctx, const uint8_t *const *src, size_t src_stride,
                                   uint8_t *const *dst, size_t dst_stride, int width, int height, int n_planes)
{
    int rc = check_ctx_set(ctxs, n_ctx);
    if (rc) return rc;
    if (!src || !dst || n_planes <= 0 || width <= 0 || height <= 0 || src_stride < (size_t)width || dst_stride < (size_t)width)
        return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_striped_frames: bad arguments");
    for (int p = 0; p < n_planes; ++p)
        if (!src[p] || !dst[p]) return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_striped_frames: null plane %d", p);
    if (n_ctx > 1 && height / n_ctx < kHalo)
        return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_striped_frames: stripes thinner than the %d-row halo", kHalo);
    // set-up on every context: lanes, events, links; the pipeline needs the one-launch form (float32 MFMA kernel, neighbours'
    // rows readable where they lie)
    bool one_launch = n_ctx > 1;
    rc = run_per_context(ctxs, n_ctx, [&](int k) -> int {
        srcnn_ctx *c = ctxs[k];
        BIND(c);
        int r, r0, r1;
        srcnn_stripe_rows(height, n_ctx, k, &r0, &r1);
        if ((r = ensure_lanes(c, (size_t)(r1 - r0) * width))) return r;
        for (int b = 0; b < 2; ++b)
            for (hipEvent_t *e : {&c->sf_up[b], &c->sf_k[b], &c->sf_down[b]})
                if (!*e) HIP_TRY(c, hipEventCreateWithFlags(e, hipEventDisableTiming));
        c->sf_gen_up.store(0);
        c->sf_gen_k.store(0);
        c->sf_abort.store(0);
        if (n_ctx > 1 && (r = stripe_setup(ctxs, n_ctx, k))) return r;
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return SRCNN_OK;
    });
    if (rc) return rc;
    for (int k = 0; k < n_ctx; ++k)
        if (ctxs[k]->halo_transport == 3 || (ctxs[k]->mode != SRCNN_MODE_MFMA && ctxs[k]->mode != SRCNN_MODE_REFBYTES)) one_launch = false;
    if (!one_launch) {          // one context, a link without peer access, the split-f16 modes: plane by plane
        for (int p = 0; p < n_planes; ++p)
            if ((rc = srcnn_forward_y_striped(ctxs, n_ctx, src[p], src_stride, dst[p], dst_stride, width, height))) return rc;
        return SRCNN_OK;
    }
    auto aborted = [&] {
        for (int k = 0; k < n_ctx; ++k)
            if (ctxs[k]->sf_abort.load(std::memory_order_acquire)) return true;
        return false;
    };
    return run_per_context(ctxs, n_ctx, [&](int k) -> int {
        srcnn_ctx *c = ctxs[k];
        // EVERY exit of this task that is not a success tells the others to stop waiting for this context's events (advisor,
        // round 5: a device that cannot be bound returned before the flag was ever set, and the neighbours spun for ever)
        struct AbortUnlessOk {
            srcnn_ctx *c;
            bool ok = false;
            ~AbortUnlessOk() { if (!ok) c->sf_abort.store(1, std::memory_order_release); }
        } leave{c};
        BIND(c);
        int r0, r1, a0 = 0, a1 = 0;
        srcnn_stripe_rows(height, n_ctx, k, &r0, &r1);
        if (k > 0) srcnn_stripe_rows(height, n_ctx, k - 1, &a0, &a1);
        const int rows = r1 - r0;
        const size_t n = (size_t)rows * width;
        hipStream_t up = c->lane_stream[0], down = c->lane_stream[1];
        const int nb[3] = {k - 1, k, k + 1};
        // spin until context j has RECORDED its event of plane `want - 1`; false = some context gave up
        auto recorded = [&](const std::atomic<int> &gen, int want) {
            while (gen.load(std::memory_order_acquire) < want) {
                if (aborted()) return false;
                std::this_thread::yield();
            }
            return true;
        };
        auto body = [&]() -> int {
            int r;
            for (int p = 0; p <= n_planes; ++p) {
                const int b = p & 1;
                if (p < n_planes) {
                    if (p >= 2) {
                        // buffer b held plane p - 2: its readers (this context's kernel and the neighbours') must be done, and
                        // the pinned staging must have left for the device
                        for (int j : nb) {
                            if (j < 0 || j >= n_ctx) continue;
                            if (!recorded(ctxs[j]->sf_gen_k, p - 1)) return SRCNN_ERR_STATE;
                            HIP_TRY(c, hipStreamWaitEvent(up, ctxs[j]->sf_k[b], 0));
                        }
                        HIP_TRY(c, hipEventSynchronize(c->sf_up[b]));
                    }
                    copy_rows_mt<uint8_t>(static_cast<uint8_t *>(c->pin_in[b]), (size_t)width, src[p] + (size_t)r0 * src_stride, src_stride, width, rows);
                    HIP_TRY(c, hipMemcpyAsync(c->lane_in[b].p, c->pin_in[b], n, hipMemcpyHostToDevice, up));
                    HIP_TRY(c, hipEventRecord(c->sf_up[b], up));
                    c->sf_gen_up.store(p + 1, std::memory_order_release);
                    // the launch: behind the uploads of plane p on the three contexts whose rows it reads, and behind the download of
                    // plane p - 2 from the output buffer it writes
                    for (int j : nb) {
                        if (j < 0 || j >= n_ctx) continue;
                        if (!recorded(ctxs[j]->sf_gen_up, p + 1)) return SRCNN_ERR_STATE;
                        HIP_TRY(c, hipStreamWaitEvent(c->stream, ctxs[j]->sf_up[b], 0));
                    }
                    if (p >= 2) HIP_TRY(c, hipStreamWaitEvent(c->stream, c->sf_down[b], 0));
                    const uint8_t *top = k > 0 ? static_cast<const uint8_t *>(ctxs[k - 1]->lane_in[b].p) + (size_t)(a1 - a0 - kHalo) * width : nullptr;
                    const uint8_t *bot = k < n_ctx - 1 ? static_cast<const uint8_t *>(ctxs[k + 1]->lane_in[b].p) : nullptr;
                    if ((r = srcnn_forward_y_rows_halo_dev(c, static_cast<const uint8_t *>(c->lane_in[b].p), (size_t)width, r0, rows, top, bot, (size_t)width,
                                                           static_cast<uint8_t *>(c->lane_out[b].p), (size_t)width, r0, width, height, r0, r1)))
                        return r;
                    HIP_TRY(c, hipEventRecord(c->sf_k[b], c->stream));
                    c->sf_gen_k.store(p + 1, std::memory_order_release);
                    HIP_TRY(c, hipStreamWaitEvent(down, c->sf_k[b], 0));
                    HIP_TRY(c, hipMemcpyAsync(c->pin_out[b], c->lane_out[b].p, n, hipMemcpyDeviceToHost, down));
                    HIP_TRY(c, hipEventRecord(c->sf_down[b], down));
                }
                if (p >= 1) {       // hand plane p - 1 to the caller while plane p computes
                    const int q = (p - 1) & 1;
                    HIP_TRY(c, hipEventSynchronize(c->sf_down[q]));
                    copy_rows_mt<uint8_t>(dst[p - 1] + (size_t)r0 * dst_stride, dst_stride, static_cast<const uint8_t *>(c->pin_out[q]), (size_t)width, width, rows);
                }
            }
            return SRCNN_OK;
        };
        const int r = body();
        leave.ok = r == SRCNN_OK;
        if (r) c->sf_abort.store(1, std::memory_order_release);       // (at once: the waits below may take a while)
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamSynchronize(up);
        (void)hipStreamSynchronize(down);
        return r;
    }, [&] {      // a task that could not even be posted: nobody will ever record its events
        for (int k = 0; k < n_ctx; ++k) ctxs[k]->sf_abort.store(1, std::memory_order_release);
    });
}

/* Device-resident planes of a STREAM over n_ctx contexts used as LANES: plane f runs on ctxs[f % n_ctx], on that context's
 * stream, as one fused launch whose seam blocks ride behind the lane's next plane (seam deferral inside the call; the last
 * plane of every lane is flushed before the call returns).  Nothing waits: the call returns with everything queued.
 * TWO CONTEXTS ON ONE GPU are two lanes of that GPU: while the slowest compute units of one plane's launch finish, the other
 * lane's next kernel is already being dispatched onto the ones that are free -- the idle tail and the launch boundary of a
 * launch, 2 % of a 3840x2160 step but 25 % of a 576x576 one, are filled (profiles/r06/two_lane_probe.txt: 576x576 0.60 ->
 * 0.75 of the f32 MFMA peak, 1280x720 0.80 -> 0.83, 1920x1080 0.861 -> 0.872, 3840x2160 unchanged).  Contexts on different
 * GPUs are frame sharding for planes that already live there (d_src[f] / d_dst[f] on the device of ctxs[f % n_ctx]). */
int srcnn_forward_y_lanes_dev(srcnn_ctx *const *ctxs, int n_ctx, const uint8_t *const *d_src, size_t src_stride,
                              uint8_t *const *d_dst, size_t dst_stride, int width, int height, int n_planes)
{
    int rc = check_ctx_set(ctxs, n_ctx);
    if (rc) return rc;
    if (!d_src || !d_dst || n_planes <= 0) return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_lanes_dev: bad arguments");
    for (int f = 0; f < n_planes; ++f)
        if (!d_src[f] || !d_dst[f]) return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_lanes_dev: null plane %d", f);
    // one host thread queues everything: a launch is a few microseconds of host time, and the order of the lanes' submissions
    // is what lets their kernels interleave on one GPU
    std::vector<char> was(n_ctx);
    for (int k = 0; k < n_ctx; ++k) {
        was[(size_t)k] = ctxs[k]->defer_seams ? 1 : 0;
        ctxs[k]->defer_seams = true;
    }
    for (int f = 0; f < n_planes && !rc; ++f)
        rc = srcnn_forward_y_dev(ctxs[f % n_ctx], d_src[f], src_stride, 0, d_dst[f], dst_stride, 0, width, height, 1, nullptr);
    for (int k = 0; k < n_ctx; ++k) {
        ctxs[k]->defer_seams = was[(size_t)k] != 0;
        const int r = srcnn_flush(ctxs[k]);       // every lane's last plane is complete on its stream once the call has returned
        if (!rc) rc = r;
    }
    return rc;
}

int srcnn_halo_transport(const srcnn_ctx *c) { return c ? c->halo_transport : SRCNN_ERR_INVALID; }

int srcnn_forward_y_frames_multi(srcnn_ctx *const *ctxs, int n_ctx, const uint8_t *const *src, size_t src_stride,
                                 uint8_t *const *dst, size_t dst_stride, int width, int height, int n_frames)
{
    int rc = check_ctx_set(ctxs, n_ctx);
    if (rc) return rc;
    if (!src || !dst || n_frames <= 0 || width <= 0 || height <= 0 || src_stride < (size_t)width ||
        dst_stride < (size_t)width)
        return fail(ctxs[0], SRCNN_ERR_INVALID, "forward_y_frames_multi: bad arguments");
    // contiguous frame ranges, one host thread per context, no collective: frames are independent
    return run_per_context(ctxs, n_ctx, [&](int k) -> int {
        int f0, f1;
        srcnn_stripe_rows(n_frames, n_ctx, k, &f0, &f1);
        if (f1 == f0) return SRCNN_OK;
        return srcnn_forward_y_frames(ctxs[k], src + f0, src_stride, dst + f0, dst_stride, width, height, f1 - f0);
    });
}

}  // extern "C"
