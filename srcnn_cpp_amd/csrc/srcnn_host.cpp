// srcnn_host.cpp -- the host-buffer entry points of include/srcnn_amd.h: staging through context-owned device buffers, the band
// pipeline of one large plane and the two-lane pipeline of a frame stream, the reference call surface on host planes
// (src/srcnn.cpp:60-73) and with the 32-plane map kept on the device, and the steps either side of the path (colour conversion,
// bicubic resize; src/srcnn.cpp:505-659).
#include "srcnn_ctx.h"

using namespace srcnn;
using namespace srcnn::host;

namespace srcnn {
namespace host {

// Keys cubic (A = -0.75) coefficient table of one axis in OpenCV's 11-bit fixed point:
// ofs[d] = floor((d + 0.5) * n_src / n_dst - 0.5), coef[d][0..3] = round(2048 * w_k(frac)).
// Float arithmetic in exactly this order (no contraction): cv::resize INTER_CUBIC, 8-bit path.
#pragma clang fp contract(off)
void cubic_table(int n_src, int n_dst, int *ofs, short *coef)
{
    const double scale = 1.0 / ((double)n_dst / n_src);
    const float A = -0.75f;
    for (int d = 0; d < n_dst; ++d) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        const int sx = (int)std::floor(fx);
        fx -= sx;
        float cf[4];
        cf[0] = ((A * (fx + 1) - 5 * A) * (fx + 1) + 8 * A) * (fx + 1) - 4 * A;
        cf[1] = ((A + 2) * fx - (A + 3)) * fx * fx + 1;
        cf[2] = ((A + 2) * (1 - fx) - (A + 3)) * (1 - fx) * (1 - fx) + 1;
        cf[3] = 1.f - cf[0] - cf[1] - cf[2];
        ofs[d] = sx;
        for (int k = 0; k < 4; ++k) {
            const long q = std::lrintf(cf[k] * 2048.f);
            coef[4 * d + k] = (short)std::min(32767L, std::max(-32768L, q));
        }
    }
}

// Cubic coefficient tables of a (sw x sh) -> (dw x dh) resize on the device: built on the host and uploaded once for a
// stream of equally sized frames.  Layout: int xofs[dw], yofs[dh]; short alpha[4 dw], beta[4 dh].
struct ResizeTables {
    const int *xofs, *yofs;
    const short *alpha, *beta;
};
int ensure_tables(srcnn_ctx *c, int sw, int sh, int dw, int dh, ResizeTables *t)
{
    const size_t ints = (size_t)dw + dh, shorts = 4 * ((size_t)dw + dh);
    const size_t bytes = ints * 4 + shorts * 2;
    if (!(c->tables.p && c->tab_sw == sw && c->tab_sh == sh && c->tab_dw == dw && c->tab_dh == dh)) {
        std::vector<unsigned char> host(bytes);
        int *xofs = reinterpret_cast<int *>(host.data()), *yofs = xofs + dw;
        short *alpha = reinterpret_cast<short *>(yofs + dh), *beta = alpha + 4 * (size_t)dw;
        cubic_table(sw, dw, xofs, alpha);
        cubic_table(sh, dh, yofs, beta);
        int rc;
        if ((rc = reserve(c, c->tables, bytes))) return rc;
        HIP_TRY(c, hipStreamSynchronize(c->stream));      // an earlier launch may still read the old tables
        HIP_TRY(c, hipMemcpy(c->tables.p, host.data(), bytes, hipMemcpyHostToDevice));
        c->tab_sw = sw; c->tab_sh = sh; c->tab_dw = dw; c->tab_dh = dh;
    }
    t->xofs = static_cast<const int *>(c->tables.p);
    t->yofs = t->xofs + dw;
    t->alpha = reinterpret_cast<const short *>(t->yofs + dh);
    t->beta = t->alpha + 4 * (size_t)dw;
    return SRCNN_OK;
}

// Device-side cubic resize of n_planes planes.
int resize_planes_dev(srcnn_ctx *c, const uint8_t *src, long sstride, long spitch, int sw, int sh, uint8_t *dst,
                      long dstride, long dpitch, int dw, int dh, int n_planes)
{
    ResizeTables t;
    int rc;
    if ((rc = ensure_tables(c, sw, sh, dw, dh, &t))) return rc;
    HIP_TRY(c, launch_resize_cubic(src, sstride, spitch, sw, sh, dst, dstride, dpitch, dw, dh, n_planes, t.xofs, t.alpha,
                                   t.yofs, t.beta, c->stream));
    return SRCNN_OK;
}

// The timed region of the reference's pipeline driver (src/srcnn.cpp:505-659) on device memory.
int process_bgr_dev(srcnn_ctx *c, const uint8_t *d_bgr, size_t stride, int w, int h, float scale, uint8_t *d_out,
                    size_t out_stride)
{
    const int ow = (int)((float)w * scale), oh = (int)((float)h * scale);    // src/srcnn.cpp:573-575
    if (ow <= 0 || oh <= 0) return fail(c, SRCNN_ERR_INVALID, "scale too small");   // :485-495
    const size_t lo = (size_t)w * h, hi = (size_t)ow * oh;
    int rc;
    if ((rc = reserve(c, c->ycc_lo, 3 * lo))) return rc;
    if ((rc = reserve(c, c->ycc_hi, 3 * hi))) return rc;
    if ((rc = reserve(c, c->y_sr, hi))) return rc;
    uint8_t *ycc_lo = static_cast<uint8_t *>(c->ycc_lo.p), *ycc_hi = static_cast<uint8_t *>(c->ycc_hi.p);
    uint8_t *y_sr = static_cast<uint8_t *>(c->y_sr.p);
    // Two launches around the conv path instead of three (and 54 MB instead of 93 MB at 1080p -> 4K): the colour conversion
    // happens while the resize stages its source tile, the resized Cr / Cb go straight into the final BGR.  Same integer
    // arithmetic per value.  SRCNN_DEBUG_PIPE3=1: the three separate kernels (A/B; also the fallback for geometries
    // outside the tiled resize's limits).
    static const char *env_pipe3 = SRCNN_DEBUG_ENV("SRCNN_DEBUG_PIPE3");
    if (!(env_pipe3 && std::atoi(env_pipe3)) && fused_pipeline_ok(w, h, ow, oh, ycc_hi, (long)ow, d_out, (long)out_stride)) {
        ResizeTables t;
        if ((rc = ensure_tables(c, w, h, ow, oh, &t))) return rc;
        HIP_TRY(c, launch_bgr_to_y_resized(d_bgr, (long)stride, w, h, ycc_hi, ow, ow, oh, t.xofs, t.alpha, t.yofs, t.beta,
                                           c->stream));                                            // :509, :540, :568-575 (Y)
        if ((rc = srcnn_forward_y_dev(c, ycc_hi, ow, hi, y_sr, ow, hi, ow, oh, 1, nullptr))) return rc;           // :609, :627
        HIP_TRY(c, launch_resize_merge(d_bgr, (long)stride, w, h, y_sr, ow, d_out, (long)out_stride, ow, oh, t.xofs, t.alpha,
                                       t.yofs, t.beta, c->stream));                                // :576-583 (Cr, Cb), :638-657
        return SRCNN_OK;
    }
    HIP_TRY(c, launch_bgr2ycrcb(d_bgr, (long)stride, w, h, ycc_lo, w, (long)lo, c->stream));      // :509, :540
    if ((rc = resize_planes_dev(c, ycc_lo, w, (long)lo, w, h, ycc_hi, ow, (long)hi, ow, oh, 3))) return rc;  // :568-583
    if ((rc = srcnn_forward_y_dev(c, ycc_hi, ow, hi, y_sr, ow, hi, ow, oh, 1, nullptr))) return rc;           // :609, :627
    HIP_TRY(c, launch_ycrcb2bgr(y_sr, ow, ycc_hi + hi, ow, (long)hi, ow, oh, d_out, (long)out_stride,
                                c->stream));                                                      // :638-657
    return SRCNN_OK;
}

// The reference surface moves 32 separately allocated float planes per call (std::vector<cv::Mat>,
// src/srcnn.cpp:602-607): 128 B/pixel over PCIe, 1.06 GB at 3840x2160.  Pageable-memory copies are staged by the
// runtime one after the other; here each plane crosses PCIe into / out of one of two PINNED slots while a few host
// threads copy the previous plane between its slot and the caller's memory.
int reserve_pin_planes(srcnn_ctx *c, size_t bytes)
{
    if (c->pin_plane_cap >= bytes) return SRCNN_OK;
    for (int k = 0; k < 2; ++k) {
        if (c->pin_plane[k]) (void)hipHostFree(c->pin_plane[k]);
        c->pin_plane[k] = nullptr;
    }
    c->pin_plane_cap = 0;
    for (int k = 0; k < 2; ++k) HIP_TRY(c, hipHostMalloc(&c->pin_plane[k], bytes, hipHostMallocDefault));
    c->pin_plane_cap = bytes;
    return SRCNN_OK;
}

// device planes (packed, plane k at d_planes + k * pitch) -> the caller's n_planes host planes
int planes_to_host(srcnn_ctx *c, const float *d_planes, size_t pitch, float *const *dst, size_t dst_stride, int width,
                   int height, int n_planes)
{
    const size_t bytes = (size_t)width * height * sizeof(float);
    int rc;
    if ((rc = reserve_pin_planes(c, bytes))) return rc;
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k) HIP_TRY(c, hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    hipError_t e = hipSuccess;
    for (int k = 0; k <= n_planes && e == hipSuccess; ++k) {
        if (k < n_planes) {
            e = hipMemcpyAsync(c->pin_plane[k & 1], d_planes + pitch * k, bytes, hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipEventRecord(ev[k & 1], c->stream);
        }
        if (k > 0 && e == hipSuccess) {        // plane k-1 has landed in its slot: hand it over while plane k is in flight
            e = hipEventSynchronize(ev[(k - 1) & 1]);
            if (e == hipSuccess)
                copy_rows_mt(dst[k - 1], dst_stride, static_cast<const float *>(c->pin_plane[(k - 1) & 1]), (size_t)width, width, height);
        }
    }
    for (int k = 0; k < 2; ++k) (void)hipEventDestroy(ev[k]);
    if (e != hipSuccess) return fail(c, SRCNN_ERR_HIP, "planes_to_host: %s", hipGetErrorString(e));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SRCNN_OK;
}

// the caller's n_planes host planes -> device planes (packed); asynchronous tail on the context's stream
int planes_from_host(srcnn_ctx *c, const float *const *src, size_t src_stride, float *d_planes, size_t pitch, int width,
                     int height, int n_planes)
{
    const size_t bytes = (size_t)width * height * sizeof(float);
    int rc;
    if ((rc = reserve_pin_planes(c, bytes))) return rc;
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int k = 0; k < 2; ++k) HIP_TRY(c, hipEventCreateWithFlags(&ev[k], hipEventDisableTiming));
    hipError_t e = hipSuccess;
    for (int k = 0; k < n_planes && e == hipSuccess; ++k) {
        if (k >= 2) e = hipEventSynchronize(ev[k & 1]);         // the slot's previous upload has left it
        if (e != hipSuccess) break;
        copy_rows_mt(static_cast<float *>(c->pin_plane[k & 1]), (size_t)width, src[k], src_stride, width, height);
        e = hipMemcpyAsync(d_planes + pitch * k, c->pin_plane[k & 1], bytes, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipEventRecord(ev[k & 1], c->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // the pinned slots are free again
    for (int k = 0; k < 2; ++k) (void)hipEventDestroy(ev[k]);
    if (e != hipSuccess) return fail(c, SRCNN_ERR_HIP, "planes_from_host: %s", hipGetErrorString(e));
    return SRCNN_OK;
}

}  // namespace host
}  // namespace srcnn

extern "C" {

/* ------------------------- host-buffer entry points ------------------------- */

}  // extern "C"

namespace srcnn {
namespace host {

int ensure_lanes(srcnn_ctx *c, size_t n)
{
    int rc;
    for (int k = 0; k < 2; ++k) {
        if (!c->lane_stream[k]) HIP_TRY(c, hipStreamCreateWithFlags(&c->lane_stream[k], hipStreamNonBlocking));
        if ((rc = reserve(c, c->lane_in[k], n))) return rc;
        if ((rc = reserve(c, c->lane_out[k], n))) return rc;
    }
    if (c->pin_cap < n) {       // pinned staging: copies from/to pageable memory would serialise the lanes
        for (int k = 0; k < 2; ++k) {
            if (c->pin_in[k]) (void)hipHostFree(c->pin_in[k]);
            if (c->pin_out[k]) (void)hipHostFree(c->pin_out[k]);
            c->pin_in[k] = c->pin_out[k] = nullptr;
        }
        c->pin_cap = 0;
        for (int k = 0; k < 2; ++k) {
            HIP_TRY(c, hipHostMalloc(&c->pin_in[k], n, hipHostMallocDefault));
            HIP_TRY(c, hipHostMalloc(&c->pin_out[k], n, hipHostMallocDefault));
        }
        c->pin_cap = n;
    }
    return SRCNN_OK;
}

}  // namespace host
}  // namespace srcnn

extern "C" {

/* A stream of host frames (BASELINE configs[4] shape): two lanes, each with its own HIP stream and
 * device buffers, alternate, so frame i+1's upload and frame i-1's download run while frame i's
 * kernel computes -- the PCIe transfers hide behind the MFMA-bound kernel. */
int srcnn_forward_y_frames(srcnn_ctx *c, const uint8_t *const *src, size_t src_stride, uint8_t *const *dst,
                           size_t dst_stride, int width, int height, int n_frames)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    if (!src || !dst || n_frames <= 0 || width <= 0 || height <= 0 || src_stride < (size_t)width ||
        dst_stride < (size_t)width)
        return fail(c, SRCNN_ERR_INVALID, "forward_y_frames: bad arguments");
    for (int i = 0; i < n_frames; ++i)
        if (!src[i] || !dst[i]) return fail(c, SRCNN_ERR_INVALID, "forward_y_frames: null frame %d", i);
    if (c->mode == SRCNN_MODE_EXACT) {          // verification mode: no pipelining
        for (int i = 0; i < n_frames; ++i)
            if ((rc = srcnn_forward_y(c, src[i], src_stride, dst[i], dst_stride, width, height, nullptr, 0)))
                return rc;
        return SRCNN_OK;
    }
    const size_t n = (size_t)width * height;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if ((rc = ensure_lanes(c, n))) return rc;
    hipStream_t caller = c->stream;
    // (between the caller's pageable memory and the pinned staging on a few host threads: one thread moves 16.6 MB per
    // 3840x2160 frame -- in and out -- in about the time the kernel takes, and the stream becomes host-bound)
    auto rows_copy = [&](uint8_t *d, size_t ds, const uint8_t *sp, size_t ss) { copy_rows_mt<uint8_t>(d, ds, sp, ss, width, height); };
    auto finish = [&](int i) -> hipError_t {           // wait for frame i's lane, hand the plane to the caller
        const int k = i & 1;
        hipError_t e = hipStreamSynchronize(c->lane_stream[k]);
        if (e == hipSuccess) rows_copy(dst[i], dst_stride, static_cast<const uint8_t *>(c->pin_out[k]), width);
        return e;
    };
    for (int i = 0; i < n_frames && rc == SRCNN_OK; ++i) {
        const int k = i & 1;
        // lane k is idle here: frame i-2 was finished in the previous iteration
        rows_copy(static_cast<uint8_t *>(c->pin_in[k]), width, src[i], src_stride);     // overlaps kernel i-1
        hipError_t e = hipMemcpyAsync(c->lane_in[k].p, c->pin_in[k], n, hipMemcpyHostToDevice, c->lane_stream[k]);
        if (e == hipSuccess) {
            c->stream = c->lane_stream[k];
            rc = srcnn_forward_y_dev(c, static_cast<uint8_t *>(c->lane_in[k].p), width, n,
                                     static_cast<uint8_t *>(c->lane_out[k].p), width, n, width, height, 1, nullptr);
            c->stream = caller;
        }
        if (e == hipSuccess && rc == SRCNN_OK)
            e = hipMemcpyAsync(c->pin_out[k], c->lane_out[k].p, n, hipMemcpyDeviceToHost, c->lane_stream[k]);
        if (e == hipSuccess && rc == SRCNN_OK && i > 0) e = finish(i - 1);                // overlaps kernel i
        if (e != hipSuccess && rc == SRCNN_OK)
            rc = fail(c, SRCNN_ERR_HIP, "forward_y_frames: %s", hipGetErrorString(e));
    }
    if (rc == SRCNN_OK) {
        hipError_t e = finish(n_frames - 1);
        if (e != hipSuccess) rc = fail(c, SRCNN_ERR_HIP, "forward_y_frames: %s", hipGetErrorString(e));
    }
    for (int k = 0; k < 2; ++k) (void)hipStreamSynchronize(c->lane_stream[k]);
    return rc;
}

int srcnn_forward_y(srcnn_ctx *c, const uint8_t *src, size_t src_stride, uint8_t *dst, size_t dst_stride,
                    int width, int height, float *preclamp, size_t preclamp_stride)
{
    BIND(c);
    int rc = SRCNN_OK;
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    if (bad_plane(src, src_stride, width, height) || bad_plane(dst, dst_stride, width, height) ||
        (preclamp && preclamp_stride < (size_t)width))
        return fail(c, SRCNN_ERR_INVALID, "forward_y: bad plane geometry");
    const size_t n = (size_t)width * height;
    if ((rc = reserve(c, c->in_u8, n))) return rc;
    if ((rc = reserve(c, c->out_u8, n))) return rc;
    if (preclamp && (rc = reserve(c, c->pre_f32, n * 4))) return rc;
    uint8_t *d_in = static_cast<uint8_t *>(c->in_u8.p), *d_out = static_cast<uint8_t *>(c->out_u8.p);
    float *d_pre = preclamp ? static_cast<float *>(c->pre_f32.p) : nullptr;
    // A large plane goes through in row bands: band i's rows are uploaded while band i-1 computes, and band i-1's
    // result comes back while band i computes, so only the first upload and the last download are exposed
    // (copies from / to pageable memory block this thread, not the other streams).  Any partition of the rows
    // computes the same plane (srcnn_forward_y_rows_dev).  EXACT mode and small planes: one upload, one launch.
    static const char *env_bands = SRCNN_DEBUG_ENV("SRCNN_DEBUG_BANDS");
    // bands of >= 1024 rows: shorter ones lose more in their launches than the overlap wins (measured: 3840x2160 1.35 ms
    // in one piece, 1.27 in two bands, 1.28 in four, 1.40 in eight; 7680x4320 5.20 -> 4.44 in four)
    int n_bands = env_bands ? std::atoi(env_bands) : ((long)width * height >= (4L << 20) ? std::min(8, height / 1024) : 1);
    if (c->mode == SRCNN_MODE_EXACT || preclamp || n_bands < 1) n_bands = 1;
    n_bands = std::min(n_bands, std::max(1, height / 64));
    if (n_bands == 1) {
        HIP_TRY(c, hipMemcpy2DAsync(d_in, width, src, src_stride, width, height, hipMemcpyHostToDevice, c->stream));
        rc = srcnn_forward_y_dev(c, d_in, width, n, d_out, width, n, width, height, 1, d_pre);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpy2DAsync(dst, dst_stride, d_out, width, width, height, hipMemcpyDeviceToHost, c->stream));
        if (preclamp)
            HIP_TRY(c, hipMemcpy2DAsync(preclamp, preclamp_stride * 4, d_pre, (size_t)width * 4, (size_t)width * 4, height,
                                        hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        return SRCNN_OK;
    }
    for (int k = 0; k < 2; ++k)
        if (!c->lane_stream[k]) HIP_TRY(c, hipStreamCreateWithFlags(&c->lane_stream[k], hipStreamNonBlocking));
    hipStream_t s_up = c->lane_stream[0], s_down = c->lane_stream[1];
    std::vector<hipEvent_t> up((size_t)n_bands, nullptr), done((size_t)n_bands, nullptr);
    auto cleanup = [&] {
        for (auto e : up) if (e) (void)hipEventDestroy(e);
        for (auto e : done) if (e) (void)hipEventDestroy(e);
    };
    hipError_t e = hipStreamSynchronize(c->stream);            // earlier work on the context's buffers
    int uploaded = 0;
    for (int i = 0; i < n_bands && e == hipSuccess && rc == SRCNN_OK; ++i) {
        int r0, r1;
        srcnn_stripe_rows(height, n_bands, i, &r0, &r1);
        const int need = std::min(height, r1 + 6);             // the band reads 6 rows beyond its own
        if (need > uploaded) {
            e = hipMemcpy2DAsync(d_in + (size_t)uploaded * width, width, src + (size_t)uploaded * src_stride, src_stride,
                                 width, need - uploaded, hipMemcpyHostToDevice, s_up);
            uploaded = need;
        }
        if (e == hipSuccess) e = hipEventCreateWithFlags(&up[(size_t)i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(up[(size_t)i], s_up);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, up[(size_t)i], 0);
        if (e != hipSuccess) break;
        rc = srcnn_forward_y_rows_dev(c, d_in, width, 0, d_out, width, 0, width, height, r0, r1);
        if (rc) break;
        e = hipEventCreateWithFlags(&done[(size_t)i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventRecord(done[(size_t)i], c->stream);
    }
    for (int i = 0; i < n_bands && e == hipSuccess && rc == SRCNN_OK; ++i) {
        int r0, r1;
        srcnn_stripe_rows(height, n_bands, i, &r0, &r1);
        e = hipStreamWaitEvent(s_down, done[(size_t)i], 0);
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(dst + (size_t)r0 * dst_stride, dst_stride, d_out + (size_t)r0 * width, width, width, r1 - r0,
                                 hipMemcpyDeviceToHost, s_down);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s_down);
    (void)hipStreamSynchronize(s_up);
    (void)hipStreamSynchronize(c->stream);
    cleanup();
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, SRCNN_ERR_HIP, "forward_y: %s", hipGetErrorString(e));
    return SRCNN_OK;
}


int srcnn_conv99x11(srcnn_ctx *c, const uint8_t *src, size_t src_stride, float *const *dst, size_t dst_stride,
                    int width, int height, const float *kernel99, const float *bias99, const float *kernel11,
                    const float *bias11)
{
    BIND(c);
    int rc = SRCNN_OK;
    if (bad_plane(src, src_stride, width, height) || !dst || dst_stride < (size_t)width || !kernel99 ||
        !bias99 || !kernel11 || !bias11)
        return fail(c, SRCNN_ERR_INVALID, "conv99x11: bad arguments");
    for (int k = 0; k < 32; ++k)
        if (!dst[k]) return fail(c, SRCNN_ERR_INVALID, "conv99x11: null output plane %d", k);
    if ((rc = use_layers12(c, kernel99, bias99, kernel11, bias11))) return rc;
    const size_t n = (size_t)width * height;
    if ((rc = reserve(c, c->in_u8, n))) return rc;
    if ((rc = reserve(c, c->planes, n * 32 * 4))) return rc;
    HIP_TRY(c, hipMemcpy2DAsync(c->in_u8.p, width, src, src_stride, width, height, hipMemcpyHostToDevice,
                                c->stream));
    rc = srcnn_conv99x11_dev(c, static_cast<uint8_t *>(c->in_u8.p), width, static_cast<float *>(c->planes.p),
                             width, n, width, height);
    if (rc) return rc;
    return planes_to_host(c, static_cast<const float *>(c->planes.p), n, dst, dst_stride, width, height, 32);
}

int srcnn_conv55(srcnn_ctx *c, const float *const *src, size_t src_stride, uint8_t *dst, size_t dst_stride,
                 int width, int height, const float *kernel, float bias)
{
    BIND(c);
    int rc = SRCNN_OK;
    if (!src || src_stride < (size_t)width || bad_plane(dst, dst_stride, width, height) || !kernel)
        return fail(c, SRCNN_ERR_INVALID, "conv55: bad arguments");
    for (int k = 0; k < 32; ++k)
        if (!src[k]) return fail(c, SRCNN_ERR_INVALID, "conv55: null input plane %d", k);
    if ((rc = use_layer3(c, kernel, bias))) return rc;      // src/srcnn.cpp:627
    const size_t n = (size_t)width * height;
    if ((rc = reserve(c, c->planes, n * 32 * 4))) return rc;
    if ((rc = reserve(c, c->out_u8, n))) return rc;
    if ((rc = planes_from_host(c, src, src_stride, static_cast<float *>(c->planes.p), n, width, height, 32))) return rc;
    rc = srcnn_conv55_dev(c, static_cast<float *>(c->planes.p), width, n, static_cast<uint8_t *>(c->out_u8.p),
                          width, width, height, nullptr);
    if (rc) return rc;
    HIP_TRY(c, hipMemcpy2DAsync(dst, dst_stride, c->out_u8.p, width, width, height, hipMemcpyDeviceToHost,
                                c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SRCNN_OK;
}

/* The reference's two call sites (src/srcnn.cpp:609, :627) with the 32-plane map kept in DEVICE memory between them:
 * host u8 plane in -> device planes, device planes -> host u8 plane out.  128 B/pixel never cross PCIe. */
int srcnn_conv99x11_to_dev(srcnn_ctx *c, const uint8_t *src, size_t src_stride, float *d_planes, size_t plane_stride,
                           size_t plane_pitch, int width, int height, const float *kernel99, const float *bias99,
                           const float *kernel11, const float *bias11)
{
    BIND(c);
    int rc = SRCNN_OK;
    if (bad_plane(src, src_stride, width, height) || !kernel99 || !bias99 || !kernel11 || !bias11)
        return fail(c, SRCNN_ERR_INVALID, "conv99x11_to_dev: bad arguments");
    if ((rc = use_layers12(c, kernel99, bias99, kernel11, bias11))) return rc;
    const size_t n = (size_t)width * height;
    if ((rc = reserve(c, c->in_u8, n))) return rc;
    HIP_TRY(c, hipMemcpy2DAsync(c->in_u8.p, width, src, src_stride, width, height, hipMemcpyHostToDevice, c->stream));
    // asynchronous from here on: srcnn_conv55_from_dev (same context, same stream) or srcnn_synchronize orders behind it
    return srcnn_conv99x11_dev(c, static_cast<uint8_t *>(c->in_u8.p), width, d_planes, plane_stride, plane_pitch, width, height);
}

int srcnn_conv55_from_dev(srcnn_ctx *c, const float *d_planes, size_t plane_stride, size_t plane_pitch, uint8_t *dst,
                          size_t dst_stride, int width, int height, const float *kernel, float bias)
{
    BIND(c);
    int rc = SRCNN_OK;
    if (bad_plane(dst, dst_stride, width, height) || !kernel)
        return fail(c, SRCNN_ERR_INVALID, "conv55_from_dev: bad arguments");
    if ((rc = use_layer3(c, kernel, bias))) return rc;
    const size_t n = (size_t)width * height;
    if ((rc = reserve(c, c->out_u8, n))) return rc;
    if ((rc = srcnn_conv55_dev(c, d_planes, plane_stride, plane_pitch, static_cast<uint8_t *>(c->out_u8.p), width, width, height,
                               nullptr)))
        return rc;
    HIP_TRY(c, hipMemcpy2DAsync(dst, dst_stride, c->out_u8.p, width, width, height, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SRCNN_OK;
}

int srcnn_conv99(srcnn_ctx *c, const uint8_t *src, size_t src_stride, float *dst, size_t dst_stride, int width,
                 int height, const float *kernel, float bias)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (bad_plane(src, src_stride, width, height) || bad_plane(dst, dst_stride, width, height) || !kernel)
        return fail(c, SRCNN_ERR_INVALID, "conv99: bad arguments");
    const size_t n = (size_t)width * height;
    if ((rc = reserve(c, c->in_u8, n))) return rc;
    if ((rc = reserve(c, c->plane1, n * 4))) return rc;
    if ((rc = reserve(c, c->kern, 1024 * 4))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->kern.p, kernel, 81 * 4, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(c->in_u8.p, width, src, src_stride, width, height, hipMemcpyHostToDevice,
                                c->stream));
    HIP_TRY(c, launch_conv99_exact(static_cast<uint8_t *>(c->in_u8.p), width, static_cast<float *>(c->plane1.p),
                                   width, width, height, static_cast<float *>(c->kern.p), bias, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(dst, dst_stride * 4, c->plane1.p, (size_t)width * 4, (size_t)width * 4, height,
                                hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SRCNN_OK;
}

int srcnn_conv11(srcnn_ctx *c, const float *const *src, size_t src_stride, float *dst, size_t dst_stride,
                 int width, int height, const float *kernel, float bias)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!src || src_stride < (size_t)width || bad_plane(dst, dst_stride, width, height) || !kernel)
        return fail(c, SRCNN_ERR_INVALID, "conv11: bad arguments");
    for (int k = 0; k < 64; ++k)
        if (!src[k]) return fail(c, SRCNN_ERR_INVALID, "conv11: null input plane %d", k);
    const size_t n = (size_t)width * height;
    if ((rc = reserve(c, c->planes, n * 64 * 4))) return rc;
    if ((rc = reserve(c, c->plane1, n * 4))) return rc;
    if ((rc = reserve(c, c->kern, 1024 * 4))) return rc;
    HIP_TRY(c, hipMemcpyAsync(c->kern.p, kernel, 64 * 4, hipMemcpyHostToDevice, c->stream));
    if ((rc = planes_from_host(c, src, src_stride, static_cast<float *>(c->planes.p), n, width, height, 64))) return rc;
    HIP_TRY(c, launch_conv11_exact(static_cast<float *>(c->planes.p), width, (long)n,
                                   static_cast<float *>(c->plane1.p), width, width, height,
                                   static_cast<float *>(c->kern.p), bias, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(dst, dst_stride * 4, c->plane1.p, (size_t)width * 4, (size_t)width * 4, height,
                                hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SRCNN_OK;
}

/* ------------------------- pipeline steps around the conv path ------------- */

int srcnn_scaled_size(int width, int height, float scale, int *out_w, int *out_h)
{
    if (!out_w || !out_h || width <= 0 || height <= 0) return SRCNN_ERR_INVALID;
    *out_w = (int)((float)width * scale);
    *out_h = (int)((float)height * scale);
    return (*out_w > 0 && *out_h > 0) ? SRCNN_OK : SRCNN_ERR_INVALID;
}

int srcnn_bgr2ycrcb(srcnn_ctx *c, const uint8_t *bgr, size_t stride, int width, int height, uint8_t *y,
                    uint8_t *cr, uint8_t *cb, size_t plane_stride)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!bgr || !y || !cr || !cb || width <= 0 || height <= 0 || stride < 3 * (size_t)width ||
        plane_stride < (size_t)width)
        return fail(c, SRCNN_ERR_INVALID, "bgr2ycrcb: bad arguments");
    const size_t n = (size_t)width * height;
    if ((rc = reserve(c, c->bgr_in, 3 * n))) return rc;
    if ((rc = reserve(c, c->ycc_lo, 3 * n))) return rc;
    HIP_TRY(c, hipMemcpy2DAsync(c->bgr_in.p, 3 * (size_t)width, bgr, stride, 3 * (size_t)width, height,
                                hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, launch_bgr2ycrcb(static_cast<uint8_t *>(c->bgr_in.p), 3L * width, width, height,
                                static_cast<uint8_t *>(c->ycc_lo.p), width, (long)n, c->stream));
    uint8_t *outs[3] = {y, cr, cb};
    for (int k = 0; k < 3; ++k)
        HIP_TRY(c, hipMemcpy2DAsync(outs[k], plane_stride, static_cast<uint8_t *>(c->ycc_lo.p) + n * k, width, width,
                                    height, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SRCNN_OK;
}

int srcnn_ycrcb2bgr(srcnn_ctx *c, const uint8_t *y, const uint8_t *cr, const uint8_t *cb, size_t plane_stride,
                    int width, int height, uint8_t *bgr, size_t stride)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!bgr || !y || !cr || !cb || width <= 0 || height <= 0 || stride < 3 * (size_t)width ||
        plane_stride < (size_t)width)
        return fail(c, SRCNN_ERR_INVALID, "ycrcb2bgr: bad arguments");
    const size_t n = (size_t)width * height;
    if ((rc = reserve(c, c->bgr_out, 3 * n))) return rc;
    if ((rc = reserve(c, c->ycc_hi, 3 * n))) return rc;
    const uint8_t *ins[3] = {y, cr, cb};
    for (int k = 0; k < 3; ++k)
        HIP_TRY(c, hipMemcpy2DAsync(static_cast<uint8_t *>(c->ycc_hi.p) + n * k, width, ins[k], plane_stride, width,
                                    height, hipMemcpyHostToDevice, c->stream));
    uint8_t *p = static_cast<uint8_t *>(c->ycc_hi.p);
    HIP_TRY(c, launch_ycrcb2bgr(p, width, p + n, width, (long)n, width, height,
                                static_cast<uint8_t *>(c->bgr_out.p), 3L * width, c->stream));
    HIP_TRY(c, hipMemcpy2DAsync(bgr, stride, c->bgr_out.p, 3 * (size_t)width, 3 * (size_t)width, height,
                                hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SRCNN_OK;
}

int srcnn_resize_cubic(srcnn_ctx *c, const uint8_t *src, size_t src_stride, int src_w, int src_h, uint8_t *dst,
                       size_t dst_stride, int dst_w, int dst_h)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (bad_plane(src, src_stride, src_w, src_h) || bad_plane(dst, dst_stride, dst_w, dst_h))
        return fail(c, SRCNN_ERR_INVALID, "resize_cubic: bad arguments");
    const size_t ns = (size_t)src_w * src_h, nd = (size_t)dst_w * dst_h;
    if ((rc = reserve(c, c->ycc_lo, ns))) return rc;
    if ((rc = reserve(c, c->ycc_hi, nd))) return rc;
    HIP_TRY(c, hipMemcpy2DAsync(c->ycc_lo.p, src_w, src, src_stride, src_w, src_h, hipMemcpyHostToDevice,
                                c->stream));
    if ((rc = resize_planes_dev(c, static_cast<uint8_t *>(c->ycc_lo.p), src_w, (long)ns, src_w, src_h,
                                static_cast<uint8_t *>(c->ycc_hi.p), dst_w, (long)nd, dst_w, dst_h, 1)))
        return rc;
    HIP_TRY(c, hipMemcpy2DAsync(dst, dst_stride, c->ycc_hi.p, dst_w, dst_w, dst_h, hipMemcpyDeviceToHost,
                                c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SRCNN_OK;
}

int srcnn_process_bgr_dev(srcnn_ctx *c, const uint8_t *d_bgr, size_t stride, int width, int height, float scale,
                          uint8_t *d_out, size_t out_stride)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    int ow = 0, oh = 0;
    if (!d_bgr || !d_out || width <= 0 || height <= 0 || stride < 3 * (size_t)width ||
        srcnn_scaled_size(width, height, scale, &ow, &oh) != SRCNN_OK || out_stride < 3 * (size_t)ow)
        return fail(c, SRCNN_ERR_INVALID, "process_bgr_dev: bad arguments");
    return process_bgr_dev(c, d_bgr, stride, width, height, scale, d_out, out_stride);
}

int srcnn_process_bgr(srcnn_ctx *c, const uint8_t *bgr, size_t stride, int width, int height, float scale,
                      uint8_t *out, size_t out_stride)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    int ow = 0, oh = 0;
    if (!bgr || !out || width <= 0 || height <= 0 || stride < 3 * (size_t)width ||
        srcnn_scaled_size(width, height, scale, &ow, &oh) != SRCNN_OK || out_stride < 3 * (size_t)ow)
        return fail(c, SRCNN_ERR_INVALID, "process_bgr: bad arguments");
    if ((rc = reserve(c, c->bgr_in, 3 * (size_t)width * height))) return rc;
    if ((rc = reserve(c, c->bgr_out, 3 * (size_t)ow * oh))) return rc;
    HIP_TRY(c, hipMemcpy2DAsync(c->bgr_in.p, 3 * (size_t)width, bgr, stride, 3 * (size_t)width, height,
                                hipMemcpyHostToDevice, c->stream));
    if ((rc = process_bgr_dev(c, static_cast<uint8_t *>(c->bgr_in.p), 3 * (size_t)width, width, height, scale,
                              static_cast<uint8_t *>(c->bgr_out.p), 3 * (size_t)ow)))
        return rc;
    HIP_TRY(c, hipMemcpy2DAsync(out, out_stride, c->bgr_out.p, 3 * (size_t)ow, 3 * (size_t)ow, oh,
                                hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return SRCNN_OK;
}



}  // extern "C"
