// srcnn_api.cpp -- the C-ABI layer (include/srcnn_amd.h) over the HIP kernels: context life cycle, modes, streams, device
// memory for hosts that hold no HIP headers, the REFBYTES monitor.  The rest of the layer: see srcnn_ctx.h.
//
// Host-side mirror of the reference's call surface: srcnn_conv99 / conv11 /
// conv55 / conv99x11 take the same planes and weight tables as the reference's
// Convolution99 / Convolution11 / Convolution55 / Convolution99x11
// (src/srcnn.cpp:60-73); srcnn_forward_y is what the pipeline driver does with
// them at src/srcnn.cpp:602-627.  Host-buffer entry points stage through
// context-owned device buffers; *_dev entry points take device pointers.
#include "srcnn_ctx.h"

using namespace srcnn;
using namespace srcnn::host;

namespace srcnn {
namespace host {

int fail(srcnn_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(c->err, sizeof(c->err), fmt, ap);
        va_end(ap);
    }
    return code;
}

// Context-owned buffers may be in use by work queued on ANY stream the context was given
// (srcnn_set_stream), so growing one waits for the whole device, not just the current stream.
int reserve(srcnn_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return SRCNN_OK;
    if (b.p) {
        // a deferred seam launch may still have to READ this buffer (its seam scratch, the set a non-deferring launch shares
        // with it): queue it before the buffer goes -- the synchronize below then waits for it
        if (int rc = flush_seams(c)) return rc;
        HIP_TRY(c, hipDeviceSynchronize());
        HIP_TRY(c, hipFree(b.p));
        b.p = nullptr;
        b.cap = 0;
    }
    HIP_TRY(c, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return SRCNN_OK;
}

void release(DevBuf &b)
{
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

}  // namespace host
}  // namespace srcnn

// The interlock probe's verdict for a device (0 = the hardware interlocks the fast row body's dependencies), run once per device
// and process.
static long probe_verdict(int device, int n_devices, hipStream_t stream)
{
    static std::mutex probe_mutex;
    static std::vector<long> probe_result;          // per device: -2 = not run yet
    std::lock_guard<std::mutex> lk(probe_mutex);
    if ((int)probe_result.size() < n_devices) probe_result.resize((size_t)n_devices, -2);
    if (device < 0 || device >= (int)probe_result.size()) return -1;
    if (probe_result[(size_t)device] == -2) probe_result[(size_t)device] = srcnn::interlock_probe_mismatches(device, stream);
    return probe_result[(size_t)device];
}

extern "C" {

int srcnn_abi_version(void) { return 1; }

int srcnn_create(srcnn_ctx **out, int device)
{
    if (!out) return SRCNN_ERR_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return SRCNN_ERR_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SRCNN_ERR_NODEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SRCNN_ERR_NODEVICE;   // gfx950 code objects only
    srcnn_ctx *c = new (std::nothrow) srcnn_ctx();
    if (!c) return SRCNN_ERR_NOMEM;
    c->device = device;
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    DeviceScope dev_scope_(c);           // like every entry point: the caller's current device is put back on return
    if (dev_scope_.rc != SRCNN_OK || hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SRCNN_ERR_HIP;
    }
    c->stream = c->own_stream;
    // The fast strip kernels rely on the hardware interlocking three inline-asm MFMA <-> vector-ALU dependencies (srcnn_probe.hip):
    // checked once per device and process; a device that does not gets the hazard-safe kernels -- same bytes, ~3 % slower.
    long bad = probe_verdict(device, n, c->own_stream);
    const char *force = SRCNN_DEBUG_ENV("SRCNN_DEBUG_FORCE_SAFE");      // test knob: behave as if the probe had failed
    if (force && std::atoi(force)) bad = 1;
    if (bad != 0) {
        c->safe_hazards = true;
        (void)fail(c, SRCNN_OK, "the MFMA interlock probe %s on device %d: this context launches the hazard-safe strip kernels",
                   bad < 0 ? "could not run" : "found differing results", device);
    }
    *out = c;
    return SRCNN_OK;
}

int srcnn_kernel_variant(const srcnn_ctx *c) { return c ? (c->safe_hazards ? 1 : 0) : SRCNN_ERR_INVALID; }

int srcnn_set_kernel_variant(srcnn_ctx *c, int variant)
{
    BIND(c);
    if (variant != 0 && variant != 1) return fail(c, SRCNN_ERR_INVALID, "set_kernel_variant: 0 (what the interlock probe allows) or 1 (hazard-safe)");
    if (variant == 1) {
        c->safe_hazards = true;
        return SRCNN_OK;
    }
    int n = 0;
    HIP_TRY(c, hipGetDeviceCount(&n));
    c->safe_hazards = probe_verdict(c->device, n, c->own_stream) != 0;      // the fast form only where the probe found it safe
    return SRCNN_OK;
}

void srcnn_destroy(srcnn_ctx *c)
{
    if (!c) return;
    c->pool.reset();                       // parks no more workers: joins them
    DeviceScope dev_scope_(c);
    (void)flush_seams(c);
    (void)hipStreamSynchronize(c->stream);
    (void)hipDeviceSynchronize();          // work on any stream the context was given may still use its buffers
    for (DevBuf *b : {&c->wfrag, &c->wraw, &c->in_u8, &c->out_u8, &c->pre_f32, &c->planes, &c->plane1, &c->kern, &c->sink,
                      &c->bgr_in, &c->bgr_out, &c->ycc_lo, &c->ycc_hi, &c->y_sr, &c->tables, &c->wfrag16,
                      &c->band_top, &c->band_bot, &c->stripe_ext})
        release(*b);
    for (int k = 0; k < srcnn_ctx::kHaloSets; ++k) {
        release(c->halo_top[k]);
        release(c->halo_bot[k]);
        if (c->halo_free[k]) (void)hipEventDestroy(c->halo_free[k]);
    }
    for (int k = 0; k < 2; ++k)
        if (c->pin_plane[k]) (void)hipHostFree(c->pin_plane[k]);
    if (c->halo_ready) (void)hipEventDestroy(c->halo_ready);
    if (c->bands_done) (void)hipEventDestroy(c->bands_done);
    if (c->halo_stream) (void)hipStreamDestroy(c->halo_stream);
    for (auto &sc : c->seam_scratch) {
        release(sc.buf);
        release(sc.cbuf);
        release(sc.buf2);
        release(sc.cbuf2);
        release(sc.flag);
        release(sc.fix_lists);
        release(sc.fix_counters);
    }
    release(c->fix_totals);
    for (auto &t : c->item_tables) {
        release(t.dev);
        release(t.dev_seams);
        release(t.dev_winmap);
    }
    for (int k = 0; k < 2; ++k) {
        for (hipEvent_t e : {c->sf_up[k], c->sf_k[k], c->sf_down[k]})
            if (e) (void)hipEventDestroy(e);
        release(c->lane_in[k]);
        release(c->lane_out[k]);
        if (c->pin_in[k]) (void)hipHostFree(c->pin_in[k]);
        if (c->pin_out[k]) (void)hipHostFree(c->pin_out[k]);
        if (c->lane_stream[k]) (void)hipStreamDestroy(c->lane_stream[k]);
    }
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}

const char *srcnn_last_error(const srcnn_ctx *c) { return c ? c->err : "null context"; }

int srcnn_set_mode(srcnn_ctx *c, int mode)
{
    if (!c || (mode != SRCNN_MODE_MFMA && mode != SRCNN_MODE_EXACT && mode != SRCNN_MODE_SPLIT16 && mode != SRCNN_MODE_REFBYTES &&
               mode != SRCNN_MODE_REFBYTES16))
        return SRCNN_ERR_INVALID;
    BIND(c);
    c->mode = mode;
    return SRCNN_OK;
}

int srcnn_get_mode(const srcnn_ctx *c) { return c ? c->mode : SRCNN_ERR_INVALID; }

int srcnn_set_stream(srcnn_ctx *c, void *hip_stream)
{
    BIND(c);                               // (deferred seam work belongs to the stream it was deferred on: queued there first)
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
    return SRCNN_OK;
}

int srcnn_set_seam_deferral(srcnn_ctx *c, int on)
{
    BIND(c);
    c->defer_seams = on != 0;
    return SRCNN_OK;
}

int srcnn_flush(srcnn_ctx *c)
{
    BIND(c);
    return SRCNN_OK;
}

int srcnn_synchronize(srcnn_ctx *c)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SRCNN_OK;
}

int srcnn_fixup_stats(srcnn_ctx *c, unsigned long long out[4], float *delta, float *max_dev)
{
    BIND(c);
    if (!out) return fail(c, SRCNN_ERR_INVALID, "fixup_stats: null output");
    unsigned long long t[FIX_TOTALS] = {0, 0, 0, 0, 0, 0};
    if (c->fix_totals.p) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpy(t, c->fix_totals.p, sizeof(t), hipMemcpyDeviceToHost));
    }
    out[0] = t[FIX_N_SCAT];
    out[1] = t[FIX_N_DENSE];
    out[2] = t[FIX_N_CHANGED];
    out[3] = t[FIX_N_RERUN];
    if (delta) *delta = c->mode == SRCNN_MODE_REFBYTES16 ? c->fix_delta * (8.f / 6.f) : c->fix_delta;
    if (max_dev) {
        const unsigned bits = (unsigned)t[FIX_MAX_DEV];
        std::memcpy(max_dev, &bits, sizeof(float));
    }
    return SRCNN_OK;
}

int srcnn_set_fixup_local(srcnn_ctx *c, float k_local)
{
    BIND(c);
    if (!(k_local >= 0.f && k_local <= 64.f)) return fail(c, SRCNN_ERR_INVALID, "set_fixup_local: the factor must lie in [0, 64]");
    // one knob for both byte-exact modes: REFBYTES16 keeps its own factor's ratio to REFBYTES' (2.15 / 1.6)
    c->fix_local = k_local;
    c->fix_local16 = k_local * (kFixLocal16 / kFixLocal);
    return SRCNN_OK;
}

int srcnn_fixup_local_stats(srcnn_ctx *c, float *k, float *max_ratio)
{
    BIND(c);
    if (k) *k = (c->mode == SRCNN_MODE_REFBYTES16 ? c->fix_local16 : c->fix_local) * c->fix_margin;
    if (max_ratio) {
        unsigned long long t[FIX_TOTALS] = {0, 0, 0, 0, 0, 0};
        if (c->fix_totals.p) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipMemcpy(t, c->fix_totals.p, sizeof(t), hipMemcpyDeviceToHost));
        }
        const unsigned bits = (unsigned)t[FIX_MAX_RATIO];
        std::memcpy(max_ratio, &bits, sizeof(float));
    }
    return SRCNN_OK;
}

int srcnn_set_fixup_margin(srcnn_ctx *c, float factor)
{
    BIND(c);
    if (!(factor >= 0.25f && factor <= 64.f)) return fail(c, SRCNN_ERR_INVALID, "set_fixup_margin: factor must lie in [0.25, 64]");
    c->fix_margin = factor;
    if (c->has_l12 || c->has_l3) {
        const float *hr = c->host_raw.data();
        c->fix_delta = fixup_delta(hr + 64, hr, hr + 5280, hr + 5248, hr + 7329, factor);
    }
    return SRCNN_OK;
}

int srcnn_set_fixup_strict(srcnn_ctx *c, int on)
{
    if (!c) return SRCNN_ERR_INVALID;
    c->fix_strict = on != 0;
    return SRCNN_OK;
}

/* Device memory for callers that hold no HIP headers (include/srcnn_amd.hpp's DevicePlane): plain allocations on the
 * context's GPU, freed by the caller; copies are synchronous and ordered behind the context's stream. */
int srcnn_dev_alloc(srcnn_ctx *c, size_t bytes, void **out)
{
    BIND(c);
    if (!out || bytes == 0) return fail(c, SRCNN_ERR_INVALID, "dev_alloc: bad arguments");
    *out = nullptr;
    HIP_TRY(c, hipMalloc(out, bytes));
    return SRCNN_OK;
}

int srcnn_dev_free(srcnn_ctx *c, void *p)
{
    BIND(c);
    if (!p) return SRCNN_OK;
    HIP_TRY(c, hipDeviceSynchronize());                   // work queued on ANY stream the context was given may still use it
    HIP_TRY(c, hipFree(p));
    return SRCNN_OK;
}

int srcnn_dev_download(srcnn_ctx *c, void *dst, const void *d_src, size_t bytes)
{
    BIND(c);
    if (!dst || !d_src) return fail(c, SRCNN_ERR_INVALID, "dev_download: null pointer");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return SRCNN_OK;
}

int srcnn_dev_upload(srcnn_ctx *c, void *d_dst, const void *src, size_t bytes)
{
    BIND(c);
    if (!d_dst || !src) return fail(c, SRCNN_ERR_INVALID, "dev_upload: null pointer");
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return SRCNN_OK;
}

/* Device memory shared between PROCESSES on one node (one process per GPU): the owner exports a handle of an allocation made
 * with srcnn_dev_alloc, a neighbour opens it and gets a device address valid in ITS process -- on another GPU the accesses go
 * over xGMI.  This is how the ranks of a row-striped plane read each other's 6 edge rows where they lie instead of exchanging
 * them every step (srcnn_forward_y_rows_halo_dev takes such addresses as its halo pointers). */
int srcnn_ipc_export(srcnn_ctx *c, void *d_ptr, unsigned char handle[64])
{
    BIND(c);
    if (!d_ptr || !handle) return fail(c, SRCNN_ERR_INVALID, "ipc_export: null pointer");
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "the ABI carries the handle as 64 bytes");
    hipIpcMemHandle_t h;
    HIP_TRY(c, hipIpcGetMemHandle(&h, d_ptr));
    std::memcpy(handle, &h, sizeof(h));
    return SRCNN_OK;
}

int srcnn_ipc_open(srcnn_ctx *c, const unsigned char handle[64], void **d_ptr)
{
    BIND(c);
    if (!d_ptr || !handle) return fail(c, SRCNN_ERR_INVALID, "ipc_open: null pointer");
    *d_ptr = nullptr;
    hipIpcMemHandle_t h;
    std::memcpy(&h, handle, sizeof(h));
    HIP_TRY(c, hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return SRCNN_OK;
}

int srcnn_ipc_close(srcnn_ctx *c, void *d_ptr)
{
    BIND(c);
    if (!d_ptr) return SRCNN_OK;
    HIP_TRY(c, hipDeviceSynchronize());                   // queued work may still read through the mapping
    HIP_TRY(c, hipIpcCloseMemHandle(d_ptr));
    return SRCNN_OK;
}

#ifdef SRCNN_TUNING_BUILD
/* ---- test / diagnostics hooks: tuning build only, not part of the ABI ---- */

/* Undocumented diagnostics hook (not part of the ABI): copy the scratch buffer to the host. */
int srcnn_debug_read_sink(srcnn_ctx *c, void *dst, size_t bytes)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!dst || bytes > c->sink.cap) return SRCNN_ERR_INVALID;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(dst, c->sink.p, bytes, hipMemcpyDeviceToHost));
    return SRCNN_OK;
}

/* Undocumented test hooks (not part of the ABI, need no device): the host-side table builders. */
int srcnn_debug_pack_fragments(const float *blob8129, float *frag /*[NFRAG*64]*/, uint8_t *frag16 /*S16_TABLE_BYTES*/,
                               int *frag_floats, int *frag16_bytes)
{
    if (frag_floats) *frag_floats = NFRAG * 64;
    if (frag16_bytes) *frag16_bytes = (int)S16_TABLE_BYTES;
    if (!blob8129) return SRCNN_ERR_INVALID;
    const float *b1 = blob8129, *w1 = blob8129 + 64, *b2 = blob8129 + 5248, *w2 = blob8129 + 5280, *w3 = blob8129 + 7329;
    if (frag) pack_fragments(w1, b1, w2, b2, w3, frag);
    if (frag16) pack_fragments16(w1, b1, w2, b2, w3, frag16);
    return split16_range_ok(w1, b1, w2, b2, w3) ? 1 : 0;
}

/* Undocumented test hook (needs no device): the persistent worker threads of the several-GPUs entry points.  `rounds` calls of
 * run() over n "contexts", growing from 1 to n; every task must run exactly once per call, on its own thread for k > 0, and the
 * first non-zero code must come back.  Returns 0 when all of that held. */
int srcnn_debug_worker_pool(int n, int rounds)
{
    if (n < 1 || n > 64 || rounds < 1) return SRCNN_ERR_INVALID;
    WorkerPool pool;
    std::vector<long> hits((size_t)n, 0);
    for (int r = 0; r < rounds; ++r) {
        const int m = 1 + r % n;
        const int fail_at = (r % 7 == 3 && m > 1) ? m - 1 : -1;
        const int rc = pool.run(m, [&](int k) -> int {
            ++hits[(size_t)k];                       // each k is touched by one thread per call: no race
            return k == fail_at ? -100 - k : 0;
        });
        if (rc != (fail_at >= 0 ? -100 - fail_at : 0)) return -1;
    }
    for (int k = 0; k < n; ++k) {
        long want = 0;
        for (int r = 0; r < rounds; ++r) want += (1 + r % n) > k;
        if (hits[(size_t)k] != want) return -2;
    }
    return pool.size() == n - 1 ? 0 : -3;
}

int srcnn_debug_cubic_table(int n_src, int n_dst, int *ofs, short *coef)
{
    if (n_src <= 0 || n_dst <= 0 || !ofs || !coef) return SRCNN_ERR_INVALID;
    cubic_table(n_src, n_dst, ofs, coef);
    return SRCNN_OK;
}
#endif  /* SRCNN_TUNING_BUILD */

}  // extern "C"
