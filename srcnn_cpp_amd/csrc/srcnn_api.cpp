// srcnn_api.cpp -- the C-ABI layer (include/srcnn_amd.h) over the HIP kernels.
//
// Host-side mirror of the reference's call surface: srcnn_conv99 / conv11 /
// conv55 / conv99x11 take the same planes and weight tables as the reference's
// Convolution99 / Convolution11 / Convolution55 / Convolution99x11
// (src/srcnn.cpp:60-73); srcnn_forward_y is what the pipeline driver does with
// them at src/srcnn.cpp:602-627.  Host-buffer entry points stage through
// context-owned device buffers; *_dev entry points take device pointers.
#include "../../include/srcnn_amd.h"
#include "srcnn_kernels.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

using namespace srcnn;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

// Persistent host threads of the several-GPUs entry points (srcnn_forward_y_striped*, srcnn_forward_y_frames_multi):
// one worker per context beyond the first, created on first use and parked on a condition variable between calls.
// Spawning and joining n_ctx - 1 std::threads PER STEP cost tens of microseconds next to 0.47 ms of kernel per rank for
// a 7680x4320 plane on 8 GPUs.  Owned by the first context of the set (a context belongs to one host thread at a time,
// include/srcnn_amd.h), destroyed with it.
class WorkerPool {
    struct Worker {
        std::thread th;
        std::mutex m;
        std::condition_variable cv;
        std::function<int()> task;
        bool has_task = false, done = false, stop = false;
        int rc = 0;
    };
    std::vector<std::unique_ptr<Worker>> workers_;

    static void loop(Worker *w)
    {
        std::unique_lock<std::mutex> lk(w->m);
        for (;;) {
            w->cv.wait(lk, [w] { return w->has_task || w->stop; });
            if (w->stop) return;
            std::function<int()> t = std::move(w->task);
            w->has_task = false;
            lk.unlock();
            const int rc = t();
            lk.lock();
            w->rc = rc;
            w->done = true;
            w->cv.notify_all();
        }
    }

public:
    WorkerPool() = default;
    WorkerPool(const WorkerPool &) = delete;
    WorkerPool &operator=(const WorkerPool &) = delete;
    ~WorkerPool()
    {
        for (auto &w : workers_) {
            { std::lock_guard<std::mutex> lk(w->m); w->stop = true; }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
    }
    // fn(k) for k = 0 .. n - 1: k = 0 on the calling thread, the others on the parked workers; returns the first non-zero code
    template <typename Fn>
    int run(int n, Fn fn)
    {
        while ((int)workers_.size() < n - 1) {
            workers_.emplace_back(new Worker());
            Worker *w = workers_.back().get();
            w->th = std::thread(loop, w);
        }
        for (int k = 1; k < n; ++k) {
            Worker *w = workers_[(size_t)k - 1].get();
            { std::lock_guard<std::mutex> lk(w->m); w->task = [&fn, k] { return fn(k); }; w->has_task = true; w->done = false; }
            w->cv.notify_all();
        }
        int first = fn(0);
        for (int k = 1; k < n; ++k) {
            Worker *w = workers_[(size_t)k - 1].get();
            std::unique_lock<std::mutex> lk(w->m);
            w->cv.wait(lk, [w] { return w->done; });
            if (!first && w->rc) first = w->rc;
        }
        return first;
    }
    int size() const { return (int)workers_.size(); }
};

}  // namespace

struct srcnn_ctx {
    int device = 0;
    int n_cu = 256;
    int mode = SRCNN_MODE_MFMA;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    char err[512] = "no error";
    // model
    bool has_l12 = false, has_l3 = false;   // which layers of the uploaded tables came from the caller (the rest are zeros)
    float b3 = 0.f;
    DevBuf wfrag;   // packed MFMA fragments [NFRAG][64]
    DevBuf wfrag16; // split-f16 fragments (SRCNN_MODE_SPLIT16), S16_TABLE_BYTES
    bool split16_ok = false;   // the uploaded weights fit the f16 ranges of that mode (split16_range_ok)
    DevBuf wraw;    // b1|W1|b2|W2|b3|W3 in convdata.h order (exact kernels)
    // staging for the host-buffer entry points
    DevBuf in_u8, out_u8, pre_f32, planes, plane1, kern, sink;
    // seam scratch (srcnn_kernels.h) is written by one launch and read by the seam kernel behind it: one buffer per
    // stream the context launches on (its own, the two frame lanes, a caller's), so launches on different
    // streams never share it
    struct SeamScratch {
        hipStream_t stream = nullptr;
        bool used = false;
        DevBuf buf, cbuf;           // row seams, column seams
        DevBuf flag, fix_lists, fix_counters;     // SRCNN_MODE_REFBYTES: flag plane, work lists, per-launch counters
    };
    SeamScratch seam_scratch[4];
    // pipeline steps around the conv path
    DevBuf bgr_in, bgr_out, ycc_lo, ycc_hi, y_sr, tables;
    int tab_sw = 0, tab_sh = 0, tab_dw = 0, tab_dh = 0;   // geometry the uploaded cubic tables are for
    // second lane of the host-frame pipeline (srcnn_forward_y_frames)
    // explicit work items of single-round launches (build_items): a small cache of device tables, one per
    // launch geometry, so that a caller alternating between a few plane sizes never waits for an upload
    struct ItemTable {
        int key[7] = {0, 0, 0, 0, 0, 0, 0};
        int count = 0;                  // 0: this geometry uses the regular grid
        int n_seams = 0;
        DevBuf dev, dev_seams;
        DevBuf dev_winmap;              // [n_strips][rows] bytes: 1 = the row lies in a seam window of that strip (separated plans)
        bool separated = false;         // seam windows of neighbouring strips share no row: one seam launch (plan_items_balanced())
        unsigned long stamp = 0;        // last use, for eviction
    };
    static constexpr int kItemTables = 32;
    ItemTable item_tables[kItemTables];
    unsigned long item_clock = 0;
    // host copy of the uploaded tables in convdata.h order: the per-call weight arguments of the reference surface
    // (srcnn_conv99x11 / srcnn_conv55) are compared against it, and equal tables are not packed or uploaded again
    std::vector<float> host_raw = std::vector<float>(8129, 0.f);
    // pinned staging of the reference surface's 32 planes (two slots, alternating) and of single planes
    void *pin_plane[2] = {nullptr, nullptr};
    size_t pin_plane_cap = 0;
    // row-striped multi-device step (srcnn_forward_y_striped*): second stream for the halo copies, band inputs
    // [6 halo rows | 12 own rows] / [12 own rows | 6 halo rows], events ordering the two streams
    hipStream_t halo_stream = nullptr;
    hipEvent_t halo_ready = nullptr, bands_done = nullptr;
    bool bands_pending = false;
    DevBuf band_top, band_bot, stripe_ext;
    // ... one-launch form (float32 MFMA kernel, StripParams::src_top).  With peer access (or neighbours on the same device) the
    // kernel reads the neighbours' edge rows WHERE THEY LIE, over xGMI: no copy, no event.  Only when a link refuses peer
    // access are the 6 halo rows either side copied (staged by the runtime) into buffers of their own, kHaloSets sets used in
    // turn, so that the copies of a step run while the kernels of the steps before it still read the other sets;
    // halo_free[i] = the launch that last read set i has finished
    static constexpr int kHaloSets = 4;
    DevBuf halo_top[kHaloSets], halo_bot[kHaloSets];
    hipEvent_t halo_free[kHaloSets] = {nullptr, nullptr, nullptr, nullptr};
    bool halo_free_set[kHaloSets] = {false, false, false, false};
    unsigned long stripe_steps = 0;
    int halo_transport = 0;                // srcnn_halo_transport(): 0 none yet, 1 same device, 2 peer access (xGMI), 3 staged by the runtime
    hipStream_t lane_stream[2] = {nullptr, nullptr};
    DevBuf lane_in[2], lane_out[2];
    void *pin_in[2] = {nullptr, nullptr}, *pin_out[2] = {nullptr, nullptr};   // pinned host staging
    size_t pin_cap = 0;
    float fix_delta = 0.f;                 // SRCNN_MODE_REFBYTES: flag threshold for the uploaded model (fixup_delta())
    DevBuf fix_totals;                     // ... and its counters accumulated over the context's launches (srcnn_fixup_stats)
    std::unique_ptr<WorkerPool> pool;      // host threads of the several-GPUs calls this context leads (WorkerPool)
};

namespace {

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

#define HIP_TRY(ctx, expr)                                                                      \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail((ctx), e_ == hipErrorOutOfMemory ? SRCNN_ERR_NOMEM : SRCNN_ERR_HIP,     \
                        "%s failed: %s", #expr, hipGetErrorString(e_));                         \
    } while (0)

// Every entry point makes the context's device current for its own duration and puts the caller's device
// back on return (a framework sharing the thread keeps ITS current device).
struct DeviceScope {
    int prev = -1, rc = SRCNN_OK;
    explicit DeviceScope(srcnn_ctx *c)
    {
        if (!c) { rc = SRCNN_ERR_INVALID; return; }
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != c->device) {
            const hipError_t e = hipSetDevice(c->device);
            if (e != hipSuccess) {
                rc = fail(c, SRCNN_ERR_HIP, "hipSetDevice(%d) failed: %s", c->device, hipGetErrorString(e));
                prev = -1;
            }
        } else {
            prev = -1;      // nothing to restore
        }
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
#define BIND(c)                 \
    DeviceScope dev_scope_(c);  \
    if (dev_scope_.rc) return dev_scope_.rc

// Context-owned buffers may be in use by work queued on ANY stream the context was given
// (srcnn_set_stream), so growing one waits for the whole device, not just the current stream.
int reserve(srcnn_ctx *c, DevBuf &b, size_t bytes)
{
    if (bytes <= b.cap) return SRCNN_OK;
    if (b.p) {
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

// Smallest e with  bound * (1 + 2^-16) <= 2^e  (0 for a zero / non-finite bound).
int scale_exponent(double bound)
{
    if (!(bound > 0.0) || !std::isfinite(bound)) return 0;
    int e = 0;
    (void)std::frexp(bound * (1.0 + 1.0 / 65536.0), &e);      // bound' = m * 2^e, m in [0.5, 1)  ->  bound' < 2^e
    return e;
}

// Power-of-two scales of layers 1 and 2 (srcnn_kernels.h): rigorous bounds of the two maps for ANY 8-bit input --
// layer-1 channel c is at most 255 * sum(max(w1, 0)) + b1, layer-2 channel k at most sum(max(w2, 0) * bound1) + b2.
// SRCNN_MODE_REFBYTES flags a pixel for exact recomputation when the MFMA path's pre-truncation value v lies within delta of
// an integer.  delta has to exceed |v_mfma - v_ref|, the difference of two float32 evaluations of the same sums in different
// orders with different roundings -- rounding NOISE: a rigorous worst-case bound (every rounding error at its maximum, all of
// one sign) is ~10 grey levels and useless, the measured maximum over 54 MPix of varied content is 4.4e-4 with a tail that
// falls by a factor of 100 per 0.9e-4 (profiles/r03/fixup_margin.txt).  The noise scales with the magnitudes the model can
// produce, so delta is tied to the model, not to a constant: E0 = 2^-24 * ||W3||_2 * B2 (one half-ulp rounding error of a
// layer-2 activation at its rigorous bound B2, carried through the 800 layer-3 weights as independent errors) is 3.3e-4 for the
// shipped model -- the scale of the measured maximum -- and delta = 6 * E0 = 2.0e-3: 4.5 x the largest difference ever seen,
// 0.4 % of the pixels flagged.  fix_apply_kernel records the largest |v_mfma - v_ref| it meets (srcnn_fixup_stats), so the margin
// of a deployment can be watched; tests/test_gpu_refbytes.py asserts it stays below delta / 2.
// A second, absolute term covers what does NOT scale with the weights: the roundings AT the output's own magnitude -- the kernels
// add b3 last (one rounding), the reference rounds its double sum to float and adds b3 (two): at most 3 half-ulps of a value
// below 256 = 2.3e-5, rigorous.  A model with small weights and a large b3 (tests/checks/soak_models.py found one: 6 * E0 =
// 3.6e-5, deviation met 3.1e-5) lives on that term alone; delta = 6 * E0 + 4 * 2^-24 * 256 (+ 6.1e-5: 2.03e-3 for the shipped model).
float fixup_delta(const float *w1, const float *b1, const float *w2, const float *b2, const float *w3)
{
    double a1[64], m2 = 0.0, s3 = 0.0;
    for (int c = 0; c < 64; ++c) {
        double s = 0.0;
        for (int t = 0; t < 81; ++t) s += std::max(w1[c * 81 + t], 0.f);
        a1[c] = std::max(0.0, 255.0 * s + b1[c]);
    }
    for (int k = 0; k < 32; ++k) {
        double s = b2[k];
        for (int i = 0; i < 64; ++i) s += (double)std::max(w2[k * 64 + i], 0.f) * a1[i];
        m2 = std::max(m2, s);
    }
    for (int i = 0; i < 800; ++i) s3 += (double)w3[i] * w3[i];
    const double d = 6.0 * std::ldexp(1.0, -24) * std::sqrt(s3) * m2 + 4.0 * std::ldexp(1.0, -24) * 256.0;
    if (!std::isfinite(d)) return 0.25f;
    return (float)std::min(0.25, std::max(d, 1e-6));
}

void layer_scales(const float *w1, const float *b1, const float *w2, const float *b2, int *e1, int *e2)
{
    double a1[64], m1 = 0.0, m2 = 0.0;
    for (int c = 0; c < 64; ++c) {
        double s = 0.0;
        for (int t = 0; t < 81; ++t) s += std::max(w1[c * 81 + t], 0.f);
        a1[c] = std::max(0.0, 255.0 * s + b1[c]);
        m1 = std::max(m1, a1[c]);
    }
    for (int k = 0; k < 32; ++k) {
        double s = b2[k];
        for (int i = 0; i < 64; ++i) s += (double)std::max(w2[k * 64 + i], 0.f) * a1[i];
        m2 = std::max(m2, s);
    }
    *e1 = scale_exponent(m1);
    *e2 = scale_exponent(m2);
}

// Pack the reference-layout weights into per-lane MFMA A-operand fragments.
// Fragment q, lane l: i = l & 31 is the accumulator row the lane's weight
// feeds, kk = l >> 5 the k-slot (see srcnn_mfma.hip header).
void pack_fragments(const float *w1 /*[64][81]*/, const float *b1, const float *w2 /*[32][64]*/,
                    const float *b2, const float *w3 /*[32][25]*/, float *out /*[NFRAG][64]*/)
{
    int e1, e2;
    layer_scales(w1, b1, w2, b2, &e1, &e2);
    for (int l = 0; l < 64; ++l) {
        const int i = l & 31, kk = l >> 5;
        const int ch = row_chan(i);
        for (int t = 0; t < 2; ++t)
            for (int s = 0; s < 41; ++s) {
                const int tap = 2 * s + kk, c = 32 * t + ch;
                out[(t * 41 + s) * 64 + l] = std::ldexp(tap < 81 ? w1[c * 81 + tap] : b1[c], -e1);
            }
        for (int t = 0; t < 2; ++t)
            for (int r = 0; r < 16; ++r) {
                const float w = w2[ch * 64 + 32 * t + 2 * r + kk];
                out[(FRAG_L2 + t * 16 + r) * 64 + l] = std::ldexp(w, e1 - e2);
                out[(FRAG_L2U + t * 16 + r) * 64 + l] = std::ldexp(w, e1);
            }
        for (int r = 0; r < 16; ++r) {
            const int tap = l3_row_tap(i);
            const float w = tap >= 0 ? w3[(2 * r + kk) * 25 + tap] : 0.f;
            out[(FRAG_L3 + r) * 64 + l] = std::ldexp(w, e2);
            out[(FRAG_L3U + r) * 64 + l] = w;
        }
        for (int r = 0; r < 16; ++r) {
            out[(FRAG_B2 + r) * 64 + l] = std::ldexp(b2[2 * r + kk], -e2);
            out[(FRAG_B2U + r) * 64 + l] = b2[2 * r + kk];
        }
    }
}

// Split-f16 A-operand fragments (srcnn_split16.hip).  A float w becomes the f16 pair
// hi = f16(w * scale), lo = f16(w * scale - hi), both round-to-nearest; the power-of-two scales are
// listed in the kernel's header comment.
void split16(float w, float scale, uint16_t *hi, uint16_t *lo)
{
    const float ws = w * scale;
    const _Float16 h = (_Float16)ws;
    const _Float16 l = (_Float16)(ws - (float)h);
    std::memcpy(hi, &h, 2);
    std::memcpy(lo, &l, 2);
}

void pack_fragments16(const float *w1 /*[64][81]*/, const float *b1, const float *w2 /*[32][64]*/,
                      const float *b2, const float *w3 /*[32][25]*/, uint8_t *out /*S16_TABLE_BYTES*/)
{
    constexpr float SCALE = 16384.f, SCALE1 = 2048.f;     // see the scale table in srcnn_split16.hip
    uint16_t *tab = reinterpret_cast<uint16_t *>(out);
    auto slot = [&](int frag, int l, int e) -> uint16_t * { return tab + ((size_t)frag * 64 + l) * 8 + e; };
    for (int l = 0; l < 64; ++l) {
        const int m = l & 31, h = l >> 5;
        for (int e = 0; e < 8; ++e) {
            for (int t = 0; t < 2; ++t)
                for (int b = 0; b < 6; ++b) {
                    const int tap = l1s_tap(b, h, e), c = 32 * t + m;
                    const float w = tap < 0 ? 0.f : (tap == 81 ? b1[c] : w1[c * 81 + tap]);
                    split16(w, tap == 81 ? 0.125f : SCALE1, slot((2 * t) * 6 + b, l, e), slot((2 * t + 1) * 6 + b, l, e));
                }
            // layer 2, k-block b: slot 8h+e is register 8(b&1)+e of layer-1 tile b>>1 on lane-half h,
            // i.e. layer-1 channel 32(b>>1) + acc_row(8(b&1)+e, h); row m = layer-2 channel m
            for (int b = 0; b < 4; ++b) {
                const int c1 = 32 * (b >> 1) + acc_row(8 * (b & 1) + e, h);
                split16(w2[m * 64 + c1], SCALE, slot(S16_FRAG_L2 + b, l, e), slot(S16_FRAG_L2 + 4 + b, l, e));
            }
            // layer 3, k-block b: slot 8h+e is layer-2 channel acc_row(8b+e, h); row m = tap l3_row_tap(m)
            for (int b = 0; b < 2; ++b) {
                const int c2 = acc_row(8 * b + e, h), tap = l3_row_tap(m);
                split16(tap >= 0 ? w3[c2 * 25 + tap] : 0.f, SCALE, slot(S16_FRAG_L3 + b, l, e),
                        slot(S16_FRAG_L3 + 2 + b, l, e));
            }
        }
    }
    float *b2t = reinterpret_cast<float *>(out + (size_t)S16_NFRAG * 64 * 16);
    for (int h = 0; h < 2; ++h)
        for (int r = 0; r < 16; ++r) b2t[h * 16 + r] = b2[acc_row(r, h)] * 0.0625f;
}

// SRCNN_MODE_SPLIT16 keeps its scaled activations below 1024 and its scaled weights inside f16
// (srcnn_split16.hip).  Rigorous bounds for ANY 8-bit input: layer-1 channel c is at most
// 255 * sum(max(w1,0)) + b1, layer-2 channel k at most sum(max(w2,0) * bound1) + b2.
bool split16_range_ok(const float *w1, const float *b1, const float *w2, const float *b2, const float *w3)
{
    float a1[64], wmax1 = 0.f, wmax23 = 0.f, a1max = 0.f, a2max = 0.f;
    for (int c = 0; c < 64; ++c) {
        double s = 0;
        for (int t = 0; t < 81; ++t) {
            s += std::max(w1[c * 81 + t], 0.f);
            wmax1 = std::max(wmax1, std::fabs(w1[c * 81 + t]));
        }
        a1[c] = std::max(0.f, (float)(255.0 * s + b1[c]));
        a1max = std::max(a1max, a1[c]);
        wmax1 = std::max(wmax1, std::fabs(b1[c]) / 256.f);      // b1/8 must fit f16 as well
    }
    for (int k = 0; k < 32; ++k) {
        double s = b2[k];
        for (int i = 0; i < 64; ++i) {
            s += (double)std::max(w2[k * 64 + i], 0.f) * a1[i];
            wmax23 = std::max(wmax23, std::fabs(w2[k * 64 + i]));
        }
        a2max = std::max(a2max, (float)s);
    }
    for (int i = 0; i < 800; ++i) wmax23 = std::max(wmax23, std::fabs(w3[i]));
    return std::isfinite(a1max) && std::isfinite(a2max) && a1max < 8.f * 1024.f && a2max < 16.f * 1024.f &&
           wmax1 < 30.f && wmax23 < 3.9f;
}

int upload_weights(srcnn_ctx *c, const float *k99, const float *b99, const float *k11, const float *b11,
                   const float *k55, float b55)
{
    static const float zeros64[64] = {0};
    static std::vector<float> zero_w(64 * 81, 0.f);
    const float *w1 = k99 ? k99 : zero_w.data();
    const float *b1 = b99 ? b99 : zeros64;
    const float *w2 = k11 ? k11 : zero_w.data();
    const float *b2 = b11 ? b11 : zeros64;
    const float *w3 = k55 ? k55 : zero_w.data();
    std::vector<float> frag((size_t)NFRAG * 64);
    pack_fragments(w1, b1, w2, b2, w3, frag.data());
    std::vector<float> raw(8129 + 2048 + 5184);     // + W2 transposed [64][32] for the exact layer-1/2 kernel, + W1 transposed [81][64] (fix-up)
    std::memcpy(raw.data(), b1, 64 * 4);
    std::memcpy(raw.data() + 64, w1, 5184 * 4);
    std::memcpy(raw.data() + 5248, b2, 32 * 4);
    std::memcpy(raw.data() + 5280, w2, 2048 * 4);
    raw[7328] = b55;
    std::memcpy(raw.data() + 7329, w3, 800 * 4);
    for (int k = 0; k < 32; ++k)
        for (int i = 0; i < 64; ++i) raw[8129 + i * 32 + k] = w2[k * 64 + i];
    for (int ch = 0; ch < 64; ++ch)
        for (int t = 0; t < 81; ++t) raw[10177 + t * 64 + ch] = w1[ch * 81 + t];
    std::vector<uint8_t> frag16(S16_TABLE_BYTES);
    pack_fragments16(w1, b1, w2, b2, w3, frag16.data());
    int rc;
    if ((rc = reserve(c, c->wfrag, frag.size() * 4))) return rc;
    if ((rc = reserve(c, c->wfrag16, frag16.size()))) return rc;
    if ((rc = reserve(c, c->sink, 1 << 20))) return rc;   // scratch words (+ diagnostics in debug builds)
    if ((rc = reserve(c, c->wraw, raw.size() * 4))) return rc;
    // synchronous copies: the host vectors die at return; launches on any stream may still read the old tables
    HIP_TRY(c, hipDeviceSynchronize());
    HIP_TRY(c, hipMemcpy(c->wfrag.p, frag.data(), frag.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->wraw.p, raw.data(), raw.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->wfrag16.p, frag16.data(), frag16.size(), hipMemcpyHostToDevice));
    c->b3 = b55;
    c->fix_delta = fixup_delta(w1, b1, w2, b2, w3);
    c->split16_ok = split16_range_ok(w1, b1, w2, b2, w3);
    std::memcpy(c->host_raw.data(), raw.data(), 8129 * sizeof(float));
    return SRCNN_OK;
}

// The whole-path entry points need all three layers.  srcnn_conv99x11 / srcnn_conv55 (the reference surface) load only the
// layers they are given -- the others stay zero -- and must not make a later srcnn_forward_y run on half a model.
bool has_model(const srcnn_ctx *c) { return c->has_l12 && c->has_l3; }
const char *const kNoModel = "the model is not loaded: srcnn_set_weights not called (srcnn_conv99x11 / srcnn_conv55 load only their own layers)";

// Do two element ranges of the same device address space overlap?  (first byte, one past the last byte)
bool ranges_overlap(const void *a, size_t a_bytes, const void *b, size_t b_bytes)
{
    const uintptr_t a0 = (uintptr_t)a, b0 = (uintptr_t)b;
    return a0 < b0 + b_bytes && b0 < a0 + a_bytes;
}
size_t span_elems(size_t stride, size_t frame_pitch, int width, int height, int n_frames)
{
    return (size_t)(n_frames - 1) * frame_pitch + (size_t)(height - 1) * stride + (size_t)width;
}

constexpr int kHaloRows = 6;   // 4 input rows of the 9x9 layer + 2 feature rows of the 5x5 layer

struct Plan {
    int seg_rows, n_strips, n_segs;
};

// Choose the row-segment height: taller segments waste fewer halo rows (2*halo recomputed feature
// rows per segment) and fewer workgroup start-ups (weight fragments, 9-row Y prologue: worth about
// STARTUP_ROWS rows), more segments fill the 2-workgroups-per-CU slots more evenly.  Everything is
// regular, so scan.
Plan make_plan(const srcnn_ctx *c, int width, int rows, int n_frames, int halo, int wgs_per_cu = 2, int col_halo = -1)
{
    const int ow = FW - 2 * (col_halo < 0 ? halo : col_halo);
    Plan best{rows, (width + ow - 1) / ow, 1};
    const long slots = (long)wgs_per_cu * c->n_cu;
    double best_eff = -1.0;
    const int max_segs = std::min(rows, 4096);
    for (int ns = 1; ns <= max_segs; ++ns) {
        const int seg = (rows + ns - 1) / ns;
        const int real_ns = (rows + seg - 1) / seg;
        if (real_ns != ns) continue;
        const long wgs = (long)best.n_strips * ns * n_frames;
        const long rounds = (wgs + slots - 1) / slots;
        const double fill = (double)wgs / (double)(rounds * slots);
        constexpr int STARTUP_ROWS = 3;
        const double useful = (double)rows / ((double)ns * (seg + 2 * halo + STARTUP_ROWS));
        const double eff = fill * useful;
        if (eff > best_eff + 1e-9) {
            best_eff = eff;
            best.seg_rows = seg;
            best.n_segs = ns;
        }
    }
    return best;
}

// (row strides stay below 2^30 elements: the kernels add a lane's column to one row stride in 32 bits)
bool bad_plane(const void *p, size_t stride, int w, int h)
{
    return !p || w <= 0 || h <= 0 || stride < (size_t)w || stride >= ((size_t)1 << 30);
}
// the kernels address a lane's plane element with a 32-bit offset from a per-plane scalar base
bool bad_pitch(size_t plane_pitch) { return plane_pitch >= ((size_t)1 << 29); }

// Explicit work items for a launch that fits the GPU in ONE round with two workgroups per CU.
// The hardware hands the first n_cu blocks wave slot 0 of every CU; the MFMA pipe is arbitrated by
// age, so those run faster than the block that joins them later (tools/diag_stamps.py).  Exactly
// 2*n_cu items are made: every strip is cut into k or k+1 segments, the "fast" ones (first n_cu
// blocks) (1+skew) tall, the "slow" ones (1-skew) tall, so that all slots are used and the two
// workgroups of a CU finish together.  Placement only affects speed; the items tile the rows exactly.
// With ONE workgroup per CU (the pipelined split-f16 kernel) there are n_cu items of plain equal height
// per strip.  With `want_seams` the boundaries between the items of a strip become seams
// (srcnn_kernels.h): the items carry the ids of the seams above / below them, `seams` lists
// {strip, boundary row} per id.  `items` holds ITEM_INTS ints per block in block order; empty when the
// geometry does not qualify (the regular grid is used instead).
struct ItemPlan {
    std::vector<int> items, seams;
    bool separated = false;        // no two seams of neighbouring strips closer than SEAM_ROWS rows (plan_items_balanced())
    int count() const { return (int)items.size() / ITEM_INTS; }
    int n_seams() const { return (int)seams.size() / 2; }
};

// Balanced plan for two workgroups per CU with seams (the float32 fused kernel on a plane launched alone).
//
// Measured (tools/diag_light.py, profiles/r02/diag_light_*.txt): with two workgroups on a CU the first-dispatched
// one (wave slot 0 wins the age-based MFMA arbitration) takes kPairFast us per row, the one that joins it kPairSlow;
// a workgroup left alone on its CU takes kAlone -- less per row than either, but more than half of both together,
// so a CU is fastest when its two items end together, and the launch ends with the slowest CU.  The round-1 planner
// cut every strip into k or k+1 items of two heights and paired tall with short: CUs carried 247..255 rows and
// the median CU idled for the last 1.5-3.4 % of the launch.
//
// Here: same item counts per strip (every strip is tiled exactly by ITS items, whatever their order), same pairing
// to start from, then a local search moves single rows between two items of the same strip while that lowers the
// estimated finish time of the slower of the two CUs involved -- until the slowest CU cannot be improved.
// `cu_speed` (optional, one factor per CU) scales the estimate per CU.  It is not used in production: feeding back the
// per-XCD finish times of earlier launches was tried and made things worse -- which XCD runs 1-2 % slow changes from
// launch to launch (profiles/r02/ablation.txt).  Placement only affects speed: any plan computes the same plane.
// (round 3: least-squares fit of this model to the finish times of 4,096 CUs over 16 stamped launches with different row splits,
// profiles/r03/planner_fit.txt -- rms 6.7 us, of which launch-to-launch and per-XCD noise is most; round 2's constants were
// 6.85 / 8.35 / 4.3 / 3.0 / 7.2 and left 247..254 rows per CU where these leave 251..254)
double kPairFast = 6.40, kPairSlow = 8.40, kAlone = 3.76, kStartFast = 3.63, kStartSlow = 5.44;   // us

double cu_finish_estimate(int fast_rows, int slow_rows, double speed)
{
    static const bool once = [] {
        if (const char *e = std::getenv("SRCNN_DEBUG_RATES"))      // experiment knob: "fast,slow,alone[,start_fast,start_slow]"
            std::sscanf(e, "%lf,%lf,%lf,%lf,%lf", &kPairFast, &kPairSlow, &kAlone, &kStartFast, &kStartSlow);
        return true;
    }();
    (void)once;
    const double tf = kStartFast + fast_rows * kPairFast, ts = kStartSlow + slow_rows * kPairSlow;
    double t;
    if (tf <= ts) t = tf + std::max(0.0, slow_rows - (tf - kStartSlow) / kPairSlow) * kAlone;     // the slow one is left alone
    else t = ts + std::max(0.0, fast_rows - (ts - kStartFast) / kPairFast) * kAlone;
    return t / speed;
}

ItemPlan plan_items_balanced(int n_cu, int n_strips, int row_begin, int row_end, int skew_pct, const double *cu_speed = nullptr)
{
    const ItemPlan none;
    constexpr int kMinRows = 10;
    const int hs = row_end - row_begin;
    const int slots = 2 * n_cu;
    if (n_cu <= 0 || n_strips <= 0 || n_strips > n_cu || hs <= 0 || slots / n_strips < 2 || hs / (slots / n_strips + 1) < kMinRows + 2)
        return none;
    // items per strip and how many of them go to first-dispatched blocks (as many fast as slow items overall)
    const int kbase = slots / n_strips, kextra = slots % n_strips;
    std::vector<int> k((size_t)n_strips), a((size_t)n_strips);
    int fast_total = 0;
    for (int s = 0; s < n_strips; ++s) {
        k[(size_t)s] = kbase + (s < kextra ? 1 : 0);
        a[(size_t)s] = k[(size_t)s] / 2;
        fast_total += a[(size_t)s];
    }
    for (int pass = 0; pass < 2 && fast_total < n_cu; ++pass)
        for (int s = 0; s < n_strips && fast_total < n_cu; ++s)
            if ((pass == 1 || k[(size_t)s] % 2 == 1) && a[(size_t)s] < k[(size_t)s] - 1) { ++a[(size_t)s]; ++fast_total; }
    if (fast_total != n_cu) return none;
    struct Item { int strip, rows, cu; bool fast; };
    std::vector<Item> items;
    const double d = std::min(std::max(skew_pct, 0), 60) / 100.0;
    for (int s = 0; s < n_strips; ++s) {
        const int na = a[(size_t)s], nb = k[(size_t)s] - na;
        const double u = hs / (na * (1.0 + d) + nb * (1.0 - d));
        double acc = 0.0;
        int used = 0;
        for (int j = 0; j < k[(size_t)s]; ++j) {               // alternate tall / short down the strip
            const bool fast = (j % 2 == 0) ? (j / 2 < na) : !((j / 2) < nb);
            acc += fast ? (1.0 + d) * u : (1.0 - d) * u;
            const int upto = (j == k[(size_t)s] - 1) ? hs : (int)std::lround(acc);
            items.push_back({s, upto - used, -1, fast});
            used = upto;
        }
        // Strips get the same item heights, so their boundaries would line up from strip to strip.  Every second strip
        // is shifted up by kStagger rows (its first item shorter, its last one taller): the seam windows of neighbouring
        // strips then lie well apart (ItemPlan::separated: one seam launch instead of two) and the search below can
        // still move boundaries by a few rows.
        constexpr int kStagger = 16;
        if ((s & 1) && k[(size_t)s] >= 3) {
            Item &first = items[items.size() - (size_t)k[(size_t)s]], &last = items.back();
            const int x = std::min(kStagger, first.rows - kMinRows);
            if (x > 0) {
                first.rows -= x;
                last.rows += x;
            }
        }
    }
    // the alternation above may not hand out exactly na fast items per strip when na != nb: recount and fix the flags
    for (int s = 0; s < n_strips; ++s) {
        int have = 0;
        for (auto &it : items) if (it.strip == s && it.fast) ++have;
        for (auto &it : items) if (it.strip == s && have > a[(size_t)s] && it.fast) { it.fast = false; --have; }
        for (auto &it : items) if (it.strip == s && have < a[(size_t)s] && !it.fast) { it.fast = true; ++have; }
    }
    // pair the tallest fast item with the shortest slow one
    std::vector<int> fi, si;
    for (int i = 0; i < (int)items.size(); ++i) (items[(size_t)i].fast ? fi : si).push_back(i);
    if ((int)fi.size() != n_cu || (int)si.size() != n_cu) return none;
    std::stable_sort(fi.begin(), fi.end(), [&](int x, int y) { return items[(size_t)x].rows > items[(size_t)y].rows; });
    std::stable_sort(si.begin(), si.end(), [&](int x, int y) { return items[(size_t)x].rows < items[(size_t)y].rows; });
    for (int c = 0; c < n_cu; ++c) items[(size_t)fi[(size_t)c]].cu = items[(size_t)si[(size_t)c]].cu = c;
    auto speed = [&](int c) { return cu_speed && cu_speed[c] > 0.5 && cu_speed[c] < 2.0 ? cu_speed[c] : 1.0; };
    auto finish = [&](int c) { return cu_finish_estimate(items[(size_t)fi[(size_t)c]].rows, items[(size_t)si[(size_t)c]].rows, speed(c)); };
    // local search: take one row from an item of the slowest improvable CU, give it to the item of the same strip
    // whose CU stays fastest; stop when no such move lowers the pair's maximum
    std::vector<std::vector<int>> in_strip((size_t)n_strips);
    for (int i = 0; i < (int)items.size(); ++i) in_strip[(size_t)items[(size_t)i].strip].push_back(i);
    std::vector<double> fin((size_t)n_cu);
    for (int c = 0; c < n_cu; ++c) fin[(size_t)c] = finish(c);
    std::vector<int> by_time((size_t)n_cu);
    // Seam windows: the boundary rows of strip s (relative to row_begin) and whether they keep SEAM_ROWS rows away from
    // every boundary of the strips next to it (ItemPlan::separated: one seam launch instead of two).
    auto bounds = [&](int s_) {
        std::vector<int> b;
        int y = 0;
        for (size_t q = 0; q + 1 < in_strip[(size_t)s_].size(); ++q) b.push_back(y += items[(size_t)in_strip[(size_t)s_][q]].rows);
        return b;
    };
    auto apart = [&](const std::vector<int> &x, const std::vector<int> &y) {
        for (int u : x)
            for (int v : y)
                if (std::abs(u - v) < SEAM_ROWS) return false;
        return true;
    };
    auto strip_apart = [&](int s_) {
        const std::vector<int> b = bounds(s_);
        return (s_ == 0 || apart(b, bounds(s_ - 1))) && (s_ + 1 >= n_strips || apart(b, bounds(s_ + 1)));
    };
    // local search: take one row from an item of the slowest improvable CU, give it to the item of the same strip
    // whose CU stays fastest; stop when no such move lowers the pair's maximum.  `keep_apart`: only moves that leave the
    // strip's seam windows clear of its neighbours'.
    // the same test for ONE candidate move of the search, incrementally: a row from item `from` to item `to` of a strip
    // shifts the boundaries between them by one row (cur[s] = the strip's boundaries, kept in step by move_apply())
    std::vector<std::vector<int>> cur((size_t)n_strips);
    std::vector<int> pos(items.size());
    for (int s_ = 0; s_ < n_strips; ++s_)
        for (size_t q = 0; q < in_strip[(size_t)s_].size(); ++q) pos[(size_t)in_strip[(size_t)s_][q]] = (int)q;
    auto clear_of = [&](const std::vector<int> &nbr, int b) {       // b keeps SEAM_ROWS rows away from every entry of nbr (sorted)
        const auto it = std::lower_bound(nbr.begin(), nbr.end(), b);
        return (it == nbr.end() || *it - b >= SEAM_ROWS) && (it == nbr.begin() || b - *(it - 1) >= SEAM_ROWS);
    };
    auto move_ok = [&](int from, int to) {
        const int s_ = items[(size_t)from].strip, pa = pos[(size_t)from], pb = pos[(size_t)to];
        const int lo = std::min(pa, pb), hi = std::max(pa, pb) - 1, delta = pa < pb ? -1 : 1;
        for (int q = lo; q <= hi; ++q) {
            const int b = cur[(size_t)s_][(size_t)q] + delta;
            if (s_ > 0 && !clear_of(cur[(size_t)s_ - 1], b)) return false;
            if (s_ + 1 < n_strips && !clear_of(cur[(size_t)s_ + 1], b)) return false;
        }
        return true;
    };
    auto move_apply = [&](int from, int to) {
        const int s_ = items[(size_t)from].strip, pa = pos[(size_t)from], pb = pos[(size_t)to];
        const int lo = std::min(pa, pb), hi = std::max(pa, pb) - 1, delta = pa < pb ? -1 : 1;
        for (int q = lo; q <= hi; ++q) cur[(size_t)s_][(size_t)q] += delta;
    };
    auto search = [&](bool keep_apart) {
        for (int iter = 0; iter < 40 * n_cu; ++iter) {
            for (int c = 0; c < n_cu; ++c) by_time[(size_t)c] = c;
            std::sort(by_time.begin(), by_time.end(), [&](int x, int y) { return fin[(size_t)x] > fin[(size_t)y]; });
            bool moved = false;
            for (int rank = 0; rank < n_cu && !moved; ++rank) {
                const int c = by_time[(size_t)rank];
                double best_gain = 1e-6;
                int best_from = -1, best_to = -1;
                for (int from : {fi[(size_t)c], si[(size_t)c]}) {
                    if (items[(size_t)from].rows <= kMinRows) continue;
                    items[(size_t)from].rows -= 1;
                    const double mine = finish(c);
                    for (int to : in_strip[(size_t)items[(size_t)from].strip]) {
                        const int c2 = items[(size_t)to].cu;
                        if (c2 == c) continue;
                        items[(size_t)to].rows += 1;
                        const double theirs = finish(c2);
                        const double gain = fin[(size_t)c] - std::max(mine, theirs);
                        if (theirs < fin[(size_t)c] && gain > best_gain && (!keep_apart || move_ok(from, to))) {
                            best_gain = gain;
                            best_from = from;
                            best_to = to;
                        }
                        items[(size_t)to].rows -= 1;
                    }
                    items[(size_t)from].rows += 1;
                }
                if (best_from >= 0) {
                    if (keep_apart) move_apply(best_from, best_to);
                    items[(size_t)best_from].rows -= 1;
                    items[(size_t)best_to].rows += 1;
                    fin[(size_t)c] = finish(c);
                    fin[(size_t)items[(size_t)best_to].cu] = finish(items[(size_t)best_to].cu);
                    moved = true;
                }
            }
            if (!moved) break;
        }
    };
    // move colliding boundaries apart first (left to right: the shift that clears the left neighbour's windows with some
    // slack and keeps the two CUs involved fastest), then balance under that constraint; if that fails, balance freely
    // (two seam launches then)
    const std::vector<Item> start = items;
    static const char *env_sep = std::getenv("SRCNN_DEBUG_SEPARATE");     // experiment knob: 0 = never keep the seam windows apart
    // (worth trying only with neighbours to keep apart from and items tall enough to give up a few rows)
    bool separated = !(env_sep && std::atoi(env_sep) == 0) && n_strips >= 2 && hs / (kbase + 1) >= 24;
    constexpr int kSlack = 4;        // preferred extra distance: the search needs room to move boundaries
    for (int s_ = 1; s_ < n_strips && separated; ++s_) {
        const std::vector<int> left = bounds(s_ - 1);
        const std::vector<int> &mine = in_strip[(size_t)s_];
        for (size_t q = 0; q + 1 < mine.size() && separated; ++q) {
            int y = 0;
            for (size_t r = 0; r <= q; ++r) y += items[(size_t)mine[r]].rows;
            auto dist = [&](int b) {
                int dmin = 1 << 30;
                for (int v : left) dmin = std::min(dmin, std::abs(v - b));
                return dmin;
            };
            if (dist(y) >= SEAM_ROWS + kSlack) continue;
            Item &up = items[(size_t)mine[q]], &dn = items[(size_t)mine[q + 1]];
            int best_d = 0;
            double best_t = 1e30;
            for (int d = -3 * SEAM_ROWS; d <= 3 * SEAM_ROWS; ++d) {
                if (dist(y + d) < SEAM_ROWS || up.rows + d < kMinRows || dn.rows - d < kMinRows) continue;
                up.rows += d;
                dn.rows -= d;
                const double t = std::max(finish(up.cu), finish(dn.cu)) + 0.5 * std::abs(d) +
                                 4.0 * std::max(0, SEAM_ROWS + kSlack - dist(y + d));
                up.rows -= d;
                dn.rows += d;
                if (t < best_t) { best_t = t; best_d = d; }
            }
            if (best_t >= 1e30) { separated = false; break; }
            up.rows += best_d;
            dn.rows -= best_d;
        }
    }
    for (int s_ = 0; s_ < n_strips && separated; ++s_) separated = strip_apart(s_);
    auto slowest = [&] {
        double mx = 0.0;
        for (int c = 0; c < n_cu; ++c) mx = std::max(mx, fin[(size_t)c]);
        return mx;
    };
    if (separated) {
        for (int c = 0; c < n_cu; ++c) fin[(size_t)c] = finish(c);
        for (int s_ = 0; s_ < n_strips; ++s_) cur[(size_t)s_] = bounds(s_);
        search(true);
    }
    // the unconstrained plan, for comparison: keeping the windows apart must not cost more than the launch it saves
    // (short items -- 14 rows at 1280x720 -- cannot afford boundaries moved by four rows)
    const std::vector<Item> apart_items = items;
    const double apart_t = separated ? slowest() : 1e30;
    items = start;
    for (int c = 0; c < n_cu; ++c) fin[(size_t)c] = finish(c);
    search(false);
    constexpr double kLaunchSaved = 2.0;      // us, conservative
    if (separated && apart_t <= slowest() + kLaunchSaved) {
        items = apart_items;
        for (int c = 0; c < n_cu; ++c) fin[(size_t)c] = finish(c);
    } else {
        separated = false;
    }
    if (std::getenv("SRCNN_DEBUG_PLANLOG")) {
        double mx = 0, mn = 1e30;
        for (int c = 0; c < n_cu; ++c) { mx = std::max(mx, fin[(size_t)c]); mn = std::min(mn, fin[(size_t)c]); }
        std::fprintf(stderr, "plan: separated=%d finish %.1f..%.1f\n", (int)separated, mn, mx);
    }
    // positions: the items of a strip in creation order; seams between neighbours
    ItemPlan plan;
    plan.separated = separated;
    std::vector<int> y0(items.size()), up(items.size(), -1), dn(items.size(), -1);
    for (int s = 0; s < n_strips; ++s) {
        int y = row_begin, prev = -1;
        for (int i : in_strip[(size_t)s]) {
            if (items[(size_t)i].rows < 2 * SEAM_ROWS) return none;
            y0[(size_t)i] = y;
            y += items[(size_t)i].rows;
            if (prev >= 0) {
                const int id = plan.n_seams();
                plan.seams.insert(plan.seams.end(), {s, y0[(size_t)i]});
                dn[(size_t)prev] = id;
                up[(size_t)i] = id;
            }
            prev = i;
        }
        if (y != row_end) return none;
    }
    auto emit = [&](int i) {
        plan.items.insert(plan.items.end(), {items[(size_t)i].strip, y0[(size_t)i], y0[(size_t)i] + items[(size_t)i].rows, up[(size_t)i], dn[(size_t)i]});
    };
    for (int c = 0; c < n_cu; ++c) emit(fi[(size_t)c]);
    for (int c = 0; c < n_cu; ++c) emit(si[(size_t)c]);
    return plan;
}

ItemPlan plan_items_raw(int n_cu, int n_strips, int row_begin, int row_end, int skew_pct, int wgs_per_cu = 2,
                        bool want_seams = false)
{
    const ItemPlan none;
    static const char *env_plan = std::getenv("SRCNN_DEBUG_PLAN");      // experiment knob: 1 = the round-1 planner
    if (wgs_per_cu == 2 && want_seams && skew_pct > 0 && !(env_plan && std::atoi(env_plan) == 1)) {
        const ItemPlan balanced = plan_items_balanced(n_cu, n_strips, row_begin, row_end, skew_pct);
        if (balanced.count() > 0) return balanced;
    }
    const int rows = row_end - row_begin;
    int slots = wgs_per_cu * n_cu;
    // shortest useful item: with halo rows to recompute (4 per item) short items do not pay; with seams an item
    // only has to be tall enough for the hand-over (2 * SEAM_ROWS, 10 for some slack in the skewed heights)
    const int min_rows = want_seams ? 10 : 24;
    if (skew_pct <= 0 || n_strips <= 0 || n_strips > n_cu) return none;
    // one item per CU: a small plane may leave CUs idle rather than cut its strips into items shorter than that
    bool underfilled = false;
    if (wgs_per_cu == 1 && want_seams && rows / min_rows < slots / n_strips + 1) {
        slots = n_strips * (rows / min_rows);       // every strip in rows / min_rows items of >= min_rows rows
        underfilled = true;
    }
    if (slots / n_strips < 2 || (!underfilled && rows / (slots / n_strips + 1) < min_rows)) return none;
    if (wgs_per_cu == 1) skew_pct = 0;
    const int kbase = slots / n_strips, kextra = slots % n_strips;      // strips [0,kextra) get kbase+1 items
    std::vector<int> k(n_strips), a(n_strips);
    int fast_total = 0;
    for (int s = 0; s < n_strips; ++s) {
        k[s] = kbase + (s < kextra ? 1 : 0);
        a[s] = wgs_per_cu == 1 ? k[s] : k[s] / 2;
        fast_total += a[s];
    }
    for (int s = 0; fast_total < n_cu && s < n_strips; ++s)              // odd counts first, then any
        if (k[s] % 2 == 1 && a[s] < k[s] - 1) { ++a[s]; ++fast_total; }
    for (int s = 0; fast_total < n_cu && s < n_strips; ++s)
        if (a[s] < k[s] - 1) { ++a[s]; ++fast_total; }
    if (wgs_per_cu == 2 && fast_total != n_cu) return none;
    const double d = skew_pct / 100.0;
    std::vector<std::vector<int>> bounds(n_strips);
    for (int s = 0; s < n_strips; ++s) {
        const int b = k[s] - a[s];
        const double u = rows / (a[s] * (1.0 + d) + b * (1.0 - d));
        bounds[s].resize(k[s] + 1);
        for (int j = 0; j <= k[s]; ++j) {
            const double y = j <= a[s] ? j * (1.0 + d) * u : a[s] * (1.0 + d) * u + (j - a[s]) * (1.0 - d) * u;
            bounds[s][j] = row_begin + std::min(rows, std::max(0, (int)std::lround(y)));
        }
        bounds[s][0] = row_begin;
        bounds[s][k[s]] = row_end;
        for (int j = 1; j <= k[s]; ++j)
            if (bounds[s][j] <= bounds[s][j - 1]) return none;          // degenerate: regular grid instead
    }
    // Block i and block n_cu + i share a CU (measured, tools/diag_stamps.py): pair the tallest fast
    // item with the shortest slow one so that every CU carries the same number of rows.
    struct Item { int strip, y0, y1, up, dn; };
    ItemPlan plan;
    // a seam needs 4 rows of the item below and leaves 2 rows either side to the seam kernel
    for (int s = 0; s < n_strips && want_seams; ++s)
        for (int j = 0; j < k[s]; ++j)
            if (bounds[s][j + 1] - bounds[s][j] < 2 * SEAM_ROWS) want_seams = false;
    std::vector<Item> fast, slow;
    for (int s = 0; s < n_strips; ++s)
        for (int j = 0; j < k[s]; ++j) {
            int up = -1, dn = -1;
            if (want_seams && j > 0) up = plan.n_seams() - 1;                // made by the item above
            if (want_seams && j < k[s] - 1) {
                dn = plan.n_seams();
                plan.seams.insert(plan.seams.end(), {s, bounds[s][j + 1]});
            }
            (j < a[s] ? fast : slow).push_back({s, bounds[s][j], bounds[s][j + 1], up, dn});
        }
    std::stable_sort(fast.begin(), fast.end(), [](const Item &x, const Item &y) { return x.y1 - x.y0 > y.y1 - y.y0; });
    std::stable_sort(slow.begin(), slow.end(), [](const Item &x, const Item &y) { return x.y1 - x.y0 < y.y1 - y.y0; });
    plan.items.reserve(ITEM_INTS * (size_t)slots);
    for (const Item &it : fast) plan.items.insert(plan.items.end(), {it.strip, it.y0, it.y1, it.up, it.dn});
    for (const Item &it : slow) plan.items.insert(plan.items.end(), {it.strip, it.y0, it.y1, it.up, it.dn});
    return plan.count() == slots ? plan : none;
}

int skew_percent()
{
    static const char *env_skew = std::getenv("SRCNN_DEBUG_SKEW");     // experiment knob; 0 = regular grid
    return env_skew ? std::atoi(env_skew) : 10;
}

// A seam's WINDOW is the four output rows b-2 .. b+1 around its boundary row b, which the seam kernel finishes.  When no
// window of a strip shares a row with a window of a NEIGHBOURING strip, the block that finishes a seam can also finish the
// four column-seam pixels either side of its strip on those rows -- the neighbour's values there are complete exports of the
// strip kernel -- and the row-seam and column-seam kernels no longer depend on each other: one launch instead of two.
// The balanced planner builds such plans (ItemPlan::separated) where that costs no balance; other plans keep two launches.
ItemPlan plan_items(int n_cu, int n_strips, int row_begin, int row_end, int skew_pct, int wgs_per_cu = 2,
                    bool want_seams = false)
{
    return plan_items_raw(n_cu, n_strips, row_begin, row_end, skew_pct, wgs_per_cu, want_seams);
}

// Device copy of plan_items() for this geometry, from the context's table cache.  *n_items = 0: use the
// regular grid.  A table is written once, before its first use, into memory no earlier launch reads
// (a fresh slot, or an evicted one after its last reader has finished), and never modified afterwards.
int build_items(srcnn_ctx *c, int n_strips, int row_begin, int row_end, int wgs_per_cu, bool want_seams,
                const srcnn_ctx::ItemTable **table)
{
    *table = nullptr;
    const int key[7] = {n_strips, row_begin, row_end, skew_percent(), c->n_cu, wgs_per_cu, want_seams ? 1 : 0};
    srcnn_ctx::ItemTable *victim = &c->item_tables[0];
    for (auto &t : c->item_tables) {
        if (t.stamp && std::memcmp(key, t.key, sizeof(key)) == 0) {
            t.stamp = ++c->item_clock;
            *table = &t;
            return SRCNN_OK;
        }
        if (t.stamp < victim->stamp) victim = &t;
    }
    const ItemPlan plan = plan_items(c->n_cu, n_strips, row_begin, row_end, key[3], wgs_per_cu, want_seams);
    if (victim->stamp) HIP_TRY(c, hipDeviceSynchronize());              // evicting: its readers must be done
    if (plan.count() > 0) {
        int rc;
        if ((rc = reserve(c, victim->dev, plan.items.size() * sizeof(int)))) return rc;
        HIP_TRY(c, hipMemcpy(victim->dev.p, plan.items.data(), plan.items.size() * sizeof(int), hipMemcpyHostToDevice));
        if (plan.n_seams() > 0) {
            if ((rc = reserve(c, victim->dev_seams, plan.seams.size() * sizeof(int)))) return rc;
            HIP_TRY(c, hipMemcpy(victim->dev_seams.p, plan.seams.data(), plan.seams.size() * sizeof(int),
                                 hipMemcpyHostToDevice));
            if (plan.separated) {
                const int rows = row_end - row_begin;
                std::vector<unsigned char> win((size_t)n_strips * rows, 0);
                for (int id = 0; id < plan.n_seams(); ++id) {
                    const int s_ = plan.seams[2 * (size_t)id], b = plan.seams[2 * (size_t)id + 1];
                    for (int y = b - 2; y < b + 2; ++y)
                        if (y >= row_begin && y < row_end) win[(size_t)s_ * rows + (y - row_begin)] = 1;
                }
                if ((rc = reserve(c, victim->dev_winmap, win.size()))) return rc;
                HIP_TRY(c, hipMemcpy(victim->dev_winmap.p, win.data(), win.size(), hipMemcpyHostToDevice));
            }
        }
    }
    std::memcpy(victim->key, key, sizeof(key));
    victim->count = plan.count();
    victim->n_seams = plan.n_seams();
    victim->separated = plan.n_seams() > 0 && plan.separated;
    victim->stamp = ++c->item_clock;
    *table = victim;
    return SRCNN_OK;
}

// The split-f16 kernel is software-pipelined inside a wave and runs one workgroup per CU (srcnn_split16.hip).
int split16_wgs_per_cu(bool split16, int /*tune*/) { return split16 ? 1 : 2; }

// Seam scratch of the stream the context launches on (one buffer set per stream: srcnn_ctx::SeamScratch).
int seam_scratch_for_stream(srcnn_ctx *c, srcnn_ctx::SeamScratch **out)
{
    srcnn_ctx::SeamScratch *sc = nullptr;
    for (auto &e : c->seam_scratch)
        if (e.used && e.stream == c->stream) sc = &e;
    for (auto &e : c->seam_scratch)
        if (!sc && !e.used) sc = &e;
    if (!sc) {                  // more streams than slots: wait for everything, start over with slot 0
        HIP_TRY(c, hipDeviceSynchronize());
        for (auto &e : c->seam_scratch) e.used = false;
        sc = &c->seam_scratch[0];
    }
    sc->used = true;
    sc->stream = c->stream;
    *out = sc;
    return SRCNN_OK;
}

// Column seams (strips of FW output columns instead of FW - 4 plus two halo columns each side) pay when they save a strip:
// 3840 = 30 instead of 31, 1920 = 15 instead of 16.  Where the count is the same (576: 5 and 5) they only add the export
// work and the third kernel launch.
bool cseam_pays(int width)
{
    const int ns_cs = (width + FW - 1) / FW, ns_halo = (width + FW - 5) / (FW - 4);
    return ns_cs < ns_halo && (width - (ns_cs - 1) * FW >= 4 || ns_cs == 1);
}

// How many frames of a batch go into ONE launch of the fused kernel (srcnn_forward_y_dev; srcnn_query_plan reports the same).
// * A small batch of LARGE planes runs fastest as one single-plane launch per frame (each with its balanced item plan, back
//   to back on the stream) -- ms per frame, same box: 2 x 3840x2160 0.955 against 0.976 for one launch that repeats the item
//   plan frame after frame, 8 x 0.956 / 0.959, 24 x 0.949 / 0.946; 8 x 5760x3240 2.119 / 2.137; 4 x 1920x1080 0.252 against
//   0.275 on the regular grid, 8 x 0.253 / 0.258, 16 x 0.2525 / 0.252 (profiles/r02/ablation.txt section 11).
// * Other batches below kItemBatchMax frames repeat the plane's item plan frame after frame in one launch, whose row-seam
//   scratch is (2 n_cu - n_strips) seams x 43 KB per frame whatever the plane's size (21 MB at 3840x2160, + 4 MB of column
//   seams): at most kItemBatchChunk frames per launch, 200 MB of context-owned scratch per stream instead of 770 MB at 31.
// * Larger batches use the regular strip x segment x frame grid (column-seam scratch only, 4 MB per 3840x2160 frame),
//   at most 64 frames per launch.
constexpr int kItemBatchMax = 32, kItemBatchChunk = 8, kGridBatchChunk = 64;
// the modes whose fused pass is the float32 MFMA strip kernel (REFBYTES = the same kernel + flags + fix-up)
bool f32_mfma(const srcnn_ctx *c) { return c->mode == SRCNN_MODE_MFMA || c->mode == SRCNN_MODE_REFBYTES; }
int frames_per_launch(const srcnn_ctx *c, int width, int height, int n_frames)
{
    static const char *env_loop = std::getenv("SRCNN_DEBUG_FRAMELOOP");      // experiment knob: 0 = never one launch per frame
    const size_t px = (size_t)width * height;
    if (c->mode == SRCNN_MODE_REFBYTES || c->mode == SRCNN_MODE_REFBYTES16) return 1;       // flag planes are compact per frame; the FIX-UP spans up to FIX_BATCH_FRAMES of them (run_strip)
    if (c->mode != SRCNN_MODE_MFMA || n_frames <= 1) return kGridBatchChunk;
    if (!(env_loop && std::atoi(env_loop) == 0) &&
        ((px >= ((size_t)4 << 20) && n_frames < kItemBatchMax) || (px >= ((size_t)3 << 19) && n_frames <= 8)))
        return 1;
    return n_frames < kItemBatchMax ? kItemBatchChunk : kGridBatchChunk;
}

// Common launch of the three strip modes on device memory.
// fix_frame / fix_frames (SRCNN_MODE_REFBYTES only): this single-frame launch is frame `fix_frame` of a batch of `fix_frames`
// whose flagged pixels ONE fix-up finishes, queued behind the batch's last launch (fix_frames = 1: the launch's own fix-up).
int run_strip(srcnn_ctx *c, int mode, StripParams p, int n_frames, int fix_frame = 0, int fix_frames = 1)
{
    const int halo = (mode == MODE_L12) ? 0 : 2;
    // Undocumented experiment knobs (never set in production).  Only bits that leave every output byte as it is are honoured:
    // 2 / 16 = the stamped builds of the split-f16 / float32 production kernel (tools/diag_split16.py, diag_light.py), 8 = no XCD remap,
    // 128 = small batches on the regular grid.  A stray SRCNN_DEBUG_TUNE cannot change a pixel (tests/test_gpu_hardening.py).
    static const char *env_tune = std::getenv("SRCNN_DEBUG_TUNE");
    static const char *env_pad = std::getenv("SRCNN_DEBUG_LDS_PAD");
    constexpr int kTuneHarmless = 2 | 8 | 16 | 128;
    p.tune = (env_tune ? std::atoi(env_tune) : 0) & kTuneHarmless;
    const size_t pad = env_pad ? (size_t)std::atol(env_pad) : 0;
    const bool split16 = mode == MODE_FUSED && (c->mode == SRCNN_MODE_SPLIT16 || c->mode == SRCNN_MODE_REFBYTES16);
    const int wgs_per_cu = split16_wgs_per_cu(split16, p.tune);
    Plan pl = make_plan(c, p.width, p.row_end - p.row_begin, n_frames, halo, wgs_per_cu);
    static const char *env_segs = std::getenv("SRCNN_DEBUG_SEGS");     // experiment knob
    if (env_segs && std::atoi(env_segs) > 0) {
        const int rows = p.row_end - p.row_begin, ns = std::min(rows, std::atoi(env_segs));
        pl.seg_rows = (rows + ns - 1) / ns;
        pl.n_segs = (rows + pl.seg_rows - 1) / pl.seg_rows;
    }
    p.seg_rows = pl.seg_rows;
    p.n_strips = pl.n_strips;
    p.n_segs = pl.n_segs;
    p.items = nullptr;
    p.seam = nullptr;
    p.cseam = nullptr;
    p.strips_total = pl.n_strips;
    int grid_items = 0;
    const srcnn_ctx::ItemTable *table = nullptr;
    // Convolution55 alone (MODE_L3) reads 128 B per pixel and is bound by HBM: strips of exactly FW = 128 columns
    // (column seams instead of 2 halo columns each side), so that the four waves of a workgroup read four whole
    // 128-byte lines per plane and row -- with 124-column strips every strip start falls inside a line and a fifth
    // line is fetched: 1.32 x the algorithmic bytes by FETCH_SIZE (profiles/r02) -- and four workgroups per CU
    // (<= 128 VGPRs, 18 KB of LDS) to keep 64 KB of loads in flight per CU.  SRCNN_DEBUG_L3=0: the round-1 launch.
    static const char *env_l3 = std::getenv("SRCNN_DEBUG_L3");
    const int ns_l3 = (p.width + FW - 1) / FW;
    const bool l3_aligned = mode == MODE_L3 && !(env_l3 && std::atoi(env_l3) == 0) &&
                            (p.width - (ns_l3 - 1) * FW >= 4 || ns_l3 == 1);
    if (l3_aligned) {
        pl = make_plan(c, p.width, p.row_end - p.row_begin, n_frames, halo, 4, 0);
        p.seg_rows = pl.seg_rows;
        p.n_strips = pl.n_strips;
        p.n_segs = pl.n_segs;
        p.strips_total = pl.n_strips;
        srcnn_ctx::SeamScratch *sc = nullptr;
        int rc;
        if ((rc = seam_scratch_for_stream(c, &sc))) return rc;
        const size_t n = (size_t)n_frames * p.strips_total * (p.row_end - p.row_begin) * CSEAM_FLOATS * sizeof(float);
        if ((rc = reserve(c, sc->cbuf, n))) return rc;
        p.cseam = static_cast<float *>(sc->cbuf.p);
    }
    // One plane that fits the GPU in a single round: size the work items by the speed of the wave
    // slot they will land in and use every slot (build_items).
    const bool fused32 = mode == MODE_FUSED && !split16;
    // A small batch repeats the plane's item plan frame after frame in one launch (no halo rows, and the next frame's
    // blocks fill the CUs the last items of a frame leave idle): 8 x 3840x2160 0.857 against 0.839 on the regular
    // grid; from kItemBatchMax frames on the regular grid's tall segments are as good (64 frames: 0.863 vs 0.865).
    // (srcnn_forward_y_dev hands over at most kItemBatchChunk frames of such a batch per call: frames_per_launch())
    if (mode != MODE_L12 && (n_frames == 1 || (fused32 && n_frames < kItemBatchMax && !(p.tune & 128))) && !l3_aligned) {   // tune 128: regular grid (A/B)
        // float32 fused kernel only: seams instead of halo rows between the items of a strip, and column seams
        // instead of halo columns between strips (srcnn_kernels.h).  SRCNN_DEBUG_SEAMS: 0 = neither, 1 = rows only.
        static const char *env_seams = std::getenv("SRCNN_DEBUG_SEAMS");
        const int seam_knob = env_seams ? std::atoi(env_seams) : 3;
        const bool want_seams = mode == MODE_FUSED && !split16 && (seam_knob & 1);
        int rc;
        // a plane too small for two items per CU of useful height: one (taller) item per CU still beats the regular grid
        // with its halo rows
        auto items_for = [&](int n_strips_, bool seams_) -> int {
            int rc2 = build_items(c, n_strips_, p.row_begin, p.row_end, wgs_per_cu, seams_, &table);
            if (!rc2 && table->count == 0 && wgs_per_cu == 2 && seams_)
                rc2 = build_items(c, n_strips_, p.row_begin, p.row_end, 1, seams_, &table);
            return rc2;
        };
        bool col_seams = false;
        if (want_seams && (seam_knob & 2) && cseam_pays(p.width)) {
            // strips of FW output columns; the last strip must hold the 4 columns its left neighbour's pixels need
            const int ns_cs = (p.width + FW - 1) / FW;
            if ((rc = items_for(ns_cs, true))) return rc;
            if (table->count > 0) {
                p.strips_total = ns_cs;
                col_seams = true;
            } else {
                table = nullptr;
            }
        }
        if (!table && (rc = items_for(pl.n_strips, want_seams))) return rc;
        grid_items = table->count;
        if (grid_items > 0) {
            p.items = static_cast<const int *>(table->dev.p);
            if (table->n_seams > 0 || col_seams) {
                srcnn_ctx::SeamScratch *sc = nullptr;
                for (auto &e : c->seam_scratch)
                    if (e.used && e.stream == c->stream) sc = &e;
                for (auto &e : c->seam_scratch)
                    if (!sc && !e.used) sc = &e;
                if (!sc) {                  // more streams than slots: wait for everything, start over with slot 0
                    HIP_TRY(c, hipDeviceSynchronize());
                    for (auto &e : c->seam_scratch) e.used = false;
                    sc = &c->seam_scratch[0];
                }
                sc->used = true;
                sc->stream = c->stream;
                if (table->n_seams > 0) {
                    if ((rc = reserve(c, sc->buf, (size_t)n_frames * table->n_seams * SEAM_FLOATS * NTHREADS * sizeof(float)))) return rc;
                    p.seam = static_cast<float *>(sc->buf.p);
                }
                if (col_seams) {
                    const size_t n = (size_t)n_frames * p.strips_total * (p.row_end - p.row_begin) * CSEAM_FLOATS * sizeof(float);
                    if ((rc = reserve(c, sc->cbuf, n))) return rc;
                    p.cseam = static_cast<float *>(sc->cbuf.p);
                }
            }
            p.n_strips = 1;            // grid = n_strips * n_segs * n_frames blocks
            p.n_segs = grid_items;
            p.items_per_frame = grid_items;
            p.seams_per_frame = std::max(1, table->n_seams);
        }
    }
    // Batches (regular grid): column seams only -- the planner already makes the segments tall, and a row seam
    // costs 74 KB of scratch.
    if (fused32 && n_frames > 1 && grid_items == 0) {
        static const char *env_seams = std::getenv("SRCNN_DEBUG_SEAMS");
        const int ns_cs = (p.width + FW - 1) / FW;
        (void)ns_cs;
        if ((!env_seams || (std::atoi(env_seams) & 2)) && cseam_pays(p.width)) {
            pl = make_plan(c, p.width, p.row_end - p.row_begin, n_frames, halo, wgs_per_cu, 0);
            p.seg_rows = pl.seg_rows;
            p.n_strips = pl.n_strips;
            p.n_segs = pl.n_segs;
            p.strips_total = pl.n_strips;
            srcnn_ctx::SeamScratch *sc = nullptr;
            for (auto &e : c->seam_scratch)
                if (e.used && e.stream == c->stream) sc = &e;
            for (auto &e : c->seam_scratch)
                if (!sc && !e.used) sc = &e;
            if (!sc) {
                HIP_TRY(c, hipDeviceSynchronize());
                for (auto &e : c->seam_scratch) e.used = false;
                sc = &c->seam_scratch[0];
            }
            sc->used = true;
            sc->stream = c->stream;
            const size_t n = (size_t)n_frames * p.strips_total * (p.row_end - p.row_begin) * CSEAM_FLOATS * sizeof(float);
            int rc;
            if ((rc = reserve(c, sc->cbuf, n))) return rc;
            p.cseam = static_cast<float *>(sc->cbuf.p);
        }
    }
    // SRCNN_MODE_REFBYTES: the fused float32 kernel also writes a flag byte per pixel; fix_collect / fix_apply then recompute
    // the flagged pixels in the reference's arithmetic (srcnn_exact.hip).  One frame per strip launch; the fix-up of up to
    // FIX_BATCH_FRAMES consecutive frames of a batch is ONE pair of launches behind the last of them (srcnn_forward_y_dev): its
    // items are drawn from one list, so the draw's tail -- 3.3 rounds of items on a single 3840x2160 plane leave 18 % of the wave
    // slots empty -- is paid once per batch.
    const bool fix = mode == MODE_FUSED && (c->mode == SRCNN_MODE_REFBYTES || c->mode == SRCNN_MODE_REFBYTES16) && !p.pre;
    srcnn_ctx::SeamScratch *fsc = nullptr;
    size_t fix_scat_cap = 0, fix_dense_cap = 0, fix_flag_pitch = 0;
    float fix_delta_used = 0.f;
    if (fix) {
        if (n_frames != 1 || fix_frame < 0 || fix_frame >= fix_frames || fix_frames > FIX_BATCH_FRAMES)
            return fail(c, SRCNN_ERR_STATE, "REFBYTES strip launches hold one frame");
        int rc;
        if ((rc = seam_scratch_for_stream(c, &fsc))) return rc;
        const int rows = p.row_end - p.row_begin;
        fix_scat_cap = fixup_list_entries(p.width, rows, fix_frames, &fix_dense_cap);
        fix_flag_pitch = (size_t)rows * (size_t)p.dst_stride;
        // (sized for the whole batch at its first frame: no buffer moves while earlier frames' flags wait for the fix-up)
        if ((rc = reserve(c, fsc->flag, fix_flag_pitch * (size_t)fix_frames))) return rc;
        if ((rc = reserve(c, fsc->fix_lists, (fix_scat_cap + fix_dense_cap) * sizeof(unsigned)))) return rc;
        if ((rc = reserve(c, fsc->fix_counters, FIX_COUNTERS * sizeof(unsigned)))) return rc;
        if (!c->fix_totals.p) {
            if ((rc = reserve(c, c->fix_totals, FIX_TOTALS * sizeof(unsigned)))) return rc;
            HIP_TRY(c, hipMemsetAsync(c->fix_totals.p, 0, FIX_TOTALS * sizeof(unsigned), c->stream));
        }
        // flag[o] for the same element offsets o as dst: o >= (row_begin - dst_row0) * dst_stride
        p.flag = static_cast<uint8_t *>(fsc->flag.p) + (size_t)fix_frame * fix_flag_pitch - (long)(p.row_begin - p.dst_row0) * p.dst_stride;
        // (the split-f16 kernel's noise is a little wider than the float32 kernel's -- soak: 4.3e-4 against 3.7e-4 -- and has no
        // CPU model to take statistics from: 8 * E0 instead of 6 * E0, and the same monitor)
        fix_delta_used = c->mode == SRCNN_MODE_REFBYTES16 ? c->fix_delta * (8.f / 6.f) : c->fix_delta;
        p.fix_delta = fix_delta_used;
        p.fix_scale = 253.f / (2.f * fix_delta_used);
        p.fix_counters = static_cast<unsigned *>(fsc->fix_counters.p);
    }
    p.wfrag = static_cast<const float *>(c->wfrag.p);
    p.wfrag16 = static_cast<const uint32_t *>(c->wfrag16.p);
    p.sink = static_cast<float *>(c->sink.p);
    p.b3 = c->b3;
    if (split16) {
        if (!c->split16_ok)
            return fail(c, SRCNN_ERR_STATE, "SRCNN_MODE_SPLIT16: these weights exceed the f16 ranges of the mode "
                                            "(layer maps must stay below 8192 / 16384 for 8-bit input); use SRCNN_MODE_MFMA");
        HIP_TRY(c, launch_split16(p, n_frames, c->stream, pad));
    }
    else HIP_TRY(c, launch_strip(mode, p, n_frames, c->stream, pad));
    // Row seams and column seams in ONE launch when the plan keeps the seam windows of neighbouring strips apart
    // (plan_items_balanced()): the blocks that finish a row seam then also finish the column-seam pixels of their four
    // rows, the column-seam blocks skip those rows, and neither waits for the other.
    static const char *env_merge = std::getenv("SRCNN_DEBUG_SEAM_MERGE");      // experiment knob: 0 = two launches
    if (p.seam && p.cseam && table->separated && !(env_merge && std::atoi(env_merge) == 0)) {
        HIP_TRY(c, launch_seams_merged(p, table->n_seams * n_frames, static_cast<const int *>(table->dev_seams.p),
                                       static_cast<const unsigned char *>(table->dev_winmap.p), n_frames, c->stream));
    } else {
        if (p.seam) HIP_TRY(c, launch_seams(p, table->n_seams * n_frames, static_cast<const int *>(table->dev_seams.p), c->stream));
        if (p.cseam) HIP_TRY(c, launch_cseams(p, n_frames, c->stream));
    }
    if (fix && fix_frame == fix_frames - 1) {
        FixParams f{};
        f.n_frames = fix_frames;                         // frame 0 of the batch lies fix_frame frames before this launch's
        f.src_frame_pitch = p.src_frame_pitch;
        f.dst_frame_pitch = p.dst_frame_pitch;
        f.flag_frame_pitch = (long)fix_flag_pitch;
        f.src = p.src - (long)fix_frame * p.src_frame_pitch;
        f.src_stride = p.src_stride;
        f.src_row0 = p.src_row0;
        f.src_top = p.src_top;
        f.src_bot = p.src_bot;
        f.halo_stride = p.halo_stride;
        f.src_row1 = p.src_row1;
        f.dst = p.dst - (long)fix_frame * p.dst_frame_pitch;
        f.flag = p.flag - (long)fix_frame * (long)fix_flag_pitch;
        f.dst_stride = p.dst_stride;
        f.dst_row0 = p.dst_row0;
        f.width = p.width;
        f.height = p.height;
        f.row_begin = p.row_begin;
        f.row_end = p.row_end;
        f.wraw = static_cast<const float *>(c->wraw.p);
        f.counters = p.fix_counters;
        f.totals = static_cast<unsigned *>(c->fix_totals.p);
        f.scat = static_cast<unsigned *>(fsc->fix_lists.p);
        f.dense = f.scat + fix_scat_cap;
        f.delta = fix_delta_used;
        f.code_step = 2.f * fix_delta_used / 253.f;
        HIP_TRY(c, launch_fixup(f, c->n_cu, c->stream));
    }
    return SRCNN_OK;
}


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
    static const char *env_pipe3 = std::getenv("SRCNN_DEBUG_PIPE3");
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

// rows of `width` elements between a packed buffer and a strided one, split over a few threads
template <typename T>
void copy_rows_mt(T *dst, size_t dst_stride, const T *src, size_t src_stride, int width, int height)
{
    static const int n_thr = [] {
        const char *e = std::getenv("SRCNN_HOST_COPY_THREADS");
        const int hw = (int)std::thread::hardware_concurrency();
        return std::max(1, e ? std::atoi(e) : std::min(8, hw > 0 ? hw / 2 : 4));
    }();
    auto part = [=](int y0, int y1) {
        if (dst_stride == (size_t)width && src_stride == (size_t)width)
            std::memcpy(dst + (size_t)y0 * width, src + (size_t)y0 * width, (size_t)(y1 - y0) * width * sizeof(T));
        else
            for (int y = y0; y < y1; ++y) std::memcpy(dst + (size_t)y * dst_stride, src + (size_t)y * src_stride, (size_t)width * sizeof(T));
    };
    const int nt = (int)std::min<size_t>((size_t)n_thr, std::max<size_t>(1, (size_t)width * height * sizeof(T) >> 20));   // >= 1 MB per thread
    if (nt <= 1) { part(0, height); return; }
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(part, (int)((long)height * t / nt), (int)((long)height * (t + 1) / nt));
    part(0, height / nt);
    for (auto &th : pool) th.join();
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

}  // namespace

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
    if (hipSetDevice(device) != hipSuccess ||
        hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return SRCNN_ERR_HIP;
    }
    c->stream = c->own_stream;
    *out = c;
    return SRCNN_OK;
}

void srcnn_destroy(srcnn_ctx *c)
{
    if (!c) return;
    c->pool.reset();                       // parks no more workers: joins them
    DeviceScope dev_scope_(c);
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
    c->mode = mode;
    return SRCNN_OK;
}

int srcnn_get_mode(const srcnn_ctx *c) { return c ? c->mode : SRCNN_ERR_INVALID; }

int srcnn_set_stream(srcnn_ctx *c, void *hip_stream)
{
    if (!c) return SRCNN_ERR_INVALID;
    c->stream = hip_stream ? static_cast<hipStream_t>(hip_stream) : c->own_stream;
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

int srcnn_set_weights(srcnn_ctx *c, const float *k99, const float *b99, const float *k11, const float *b11,
                      const float *k55, float b55)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!k99 || !b99 || !k11 || !b11 || !k55) return fail(c, SRCNN_ERR_INVALID, "null weight table");
    // a caller that passes its const tables on every call (the reference does, src/srcnn.cpp:609,627) packs and uploads once
    const float *hr = c->host_raw.data();
    if (c->has_l12 && c->has_l3 && hr[7328] == b55 && !std::memcmp(hr, b99, 64 * 4) && !std::memcmp(hr + 64, k99, 5184 * 4) &&
        !std::memcmp(hr + 5248, b11, 32 * 4) && !std::memcmp(hr + 5280, k11, 2048 * 4) && !std::memcmp(hr + 7329, k55, 800 * 4))
        return SRCNN_OK;
    if ((rc = upload_weights(c, k99, b99, k11, b11, k55, b55))) return rc;
    c->has_l12 = c->has_l3 = true;
    return SRCNN_OK;
}

int srcnn_fixup_stats(srcnn_ctx *c, unsigned long long out[4], float *delta, float *max_dev)
{
    BIND(c);
    if (!out) return fail(c, SRCNN_ERR_INVALID, "fixup_stats: null output");
    unsigned t[FIX_TOTALS] = {0, 0, 0, 0};
    if (c->fix_totals.p) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        HIP_TRY(c, hipMemcpy(t, c->fix_totals.p, sizeof(t), hipMemcpyDeviceToHost));
    }
    out[0] = t[FIX_N_SCAT];
    out[1] = t[FIX_N_DENSE];
    out[2] = t[FIX_N_CHANGED];
    out[3] = 0;
    if (delta) *delta = c->mode == SRCNN_MODE_REFBYTES16 ? c->fix_delta * (8.f / 6.f) : c->fix_delta;
    if (max_dev) std::memcpy(max_dev, &t[FIX_MAX_DEV], sizeof(float));
    return SRCNN_OK;
}

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

/* Undocumented test hook (not part of the ABI, needs no device): the work-item planner.  Fills `items`
 * (ITEM_INTS ints each) and `seams` (2 ints each) up to the given capacities; returns the item count, or
 * SRCNN_ERR_INVALID when a buffer is too small. */
extern "C" int srcnn_debug_plan_items(int n_cu, int n_strips, int row_begin, int row_end, int skew_pct, int wgs_per_cu,
                                      int want_seams, int *items, int max_items, int *seams, int max_seams,
                                      int *n_seams)
{
    const ItemPlan plan = plan_items(n_cu, n_strips, row_begin, row_end, skew_pct, wgs_per_cu, want_seams != 0);
    if (plan.count() > max_items || plan.n_seams() > max_seams || !items || !seams || !n_seams) return SRCNN_ERR_INVALID;
    // an empty vector's data() may be null, which memcpy must not be given even for 0 bytes (found by UBSan)
    if (!plan.items.empty()) std::memcpy(items, plan.items.data(), plan.items.size() * sizeof(int));
    if (!plan.seams.empty()) std::memcpy(seams, plan.seams.data(), plan.seams.size() * sizeof(int));
    *n_seams = plan.n_seams();
    return plan.count();
}

int srcnn_query_plan(srcnn_ctx *c, int width, int height, int n_frames, int out[6])
{
    if (!c || !out || width <= 0 || height <= 0 || n_frames <= 0) return SRCNN_ERR_INVALID;
    static const char *env_seams = std::getenv("SRCNN_DEBUG_SEAMS");
    const int wgs_per_cu = split16_wgs_per_cu(c->mode == SRCNN_MODE_SPLIT16 || c->mode == SRCNN_MODE_REFBYTES16, 0);
    // mirrors srcnn_forward_y_dev() and run_strip(): `nl` frames go into one launch (1 = one single-plane launch per frame);
    // the float32 fused kernel uses column seams (strips of FW columns) when the geometry allows
    const int nl = std::min(n_frames, frames_per_launch(c, width, height, n_frames));
    const int seam_knob = env_seams ? std::atoi(env_seams) : 3;
    const int ns_cs = (width + FW - 1) / FW;
    bool col_seams = f32_mfma(c) && (seam_knob & 2) && (nl > 1 || (seam_knob & 1)) && cseam_pays(width);
    int items_per_cu = wgs_per_cu;
    const bool row_seams = f32_mfma(c) && (seam_knob & 1);
    auto fits = [&](int n_strips_, int per_cu) { return !plan_items(c->n_cu, n_strips_, 0, height, skew_percent(), per_cu, row_seams).items.empty(); };
    if (col_seams && nl == 1 && !fits(ns_cs, wgs_per_cu)) {
        if (wgs_per_cu == 2 && fits(ns_cs, 1)) items_per_cu = 1;
        else col_seams = false;
    }
    if (!col_seams && nl == 1 && row_seams && wgs_per_cu == 2) {
        const int ns_halo = (width + FW - 5) / (FW - 4);
        if (!fits(ns_halo, 2) && fits(ns_halo, 1)) items_per_cu = 1;
    }
    const Plan pl = make_plan(c, width, height, nl, 2, wgs_per_cu, col_seams ? 0 : -1);
    out[0] = pl.n_strips * pl.n_segs * n_frames;      // over all launches of the batch
    out[1] = pl.seg_rows;
    out[2] = pl.n_strips;
    out[3] = pl.n_segs;
    if (nl == 1 || (f32_mfma(c) && n_frames < kItemBatchMax)) {   // explicit work items (plan_items), repeated per frame of a small batch
        const std::vector<int> items =
            plan_items(c->n_cu, pl.n_strips, 0, height, skew_percent(), items_per_cu,
                       f32_mfma(c) && (seam_knob & 1)).items;
        if (!items.empty()) {
            out[0] = (int)items.size() / ITEM_INTS * n_frames;
            out[1] = 0;
            for (size_t i = 0; i < items.size(); i += ITEM_INTS) out[1] = std::max(out[1], items[i + 2] - items[i + 1]);
            out[3] = ((int)items.size() / ITEM_INTS + pl.n_strips - 1) / pl.n_strips;
        }
    }
    out[4] = (int)strip_lds_bytes(MODE_FUSED);
    out[5] = NTHREADS;
    return SRCNN_OK;
}

/* ------------------------- device-resident entry points -------------------- */

int srcnn_conv99x11_dev(srcnn_ctx *c, const uint8_t *d_src, size_t src_stride, float *d_planes,
                        size_t plane_stride, size_t plane_pitch, int width, int height)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!c->has_l12) return fail(c, SRCNN_ERR_STATE, "layers 1-2 not loaded (srcnn_set_weights / srcnn_conv99x11)");
    if (bad_plane(d_src, src_stride, width, height) || bad_plane(d_planes, plane_stride, width, height) ||
        plane_pitch < plane_stride * (size_t)height || bad_pitch(plane_pitch))
        return fail(c, SRCNN_ERR_INVALID, "conv99x11_dev: bad plane geometry");
    if (c->mode == SRCNN_MODE_EXACT) {
        HIP_TRY(c, launch_conv99x11_exact(d_src, (long)src_stride, 0, d_planes, (long)plane_stride,
                                          (long)plane_pitch, 0, width, height, 1,
                                          static_cast<const float *>(c->wraw.p), c->stream));
        return SRCNN_OK;
    }
    StripParams p{};
    p.src = d_src;
    p.src_stride = (long)src_stride;
    p.planes_out = d_planes;
    p.pl_stride = (long)plane_stride;
    p.pl_pitch = (long)plane_pitch;
    p.width = width;
    p.height = height;
    p.row_begin = 0;
    p.row_end = height;
    return run_strip(c, MODE_L12, p, 1);
}

int srcnn_conv55_dev(srcnn_ctx *c, const float *d_planes, size_t plane_stride, size_t plane_pitch,
                     uint8_t *d_dst, size_t dst_stride, int width, int height, float *d_preclamp)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!c->has_l3) return fail(c, SRCNN_ERR_STATE, "layer 3 not loaded (srcnn_set_weights / srcnn_conv55)");
    if (bad_plane(d_planes, plane_stride, width, height) || bad_plane(d_dst, dst_stride, width, height) ||
        plane_pitch < plane_stride * (size_t)height || bad_pitch(plane_pitch))
        return fail(c, SRCNN_ERR_INVALID, "conv55_dev: bad plane geometry");
    if (c->mode == SRCNN_MODE_EXACT) {
        HIP_TRY(c, launch_conv55_exact(d_planes, (long)plane_stride, (long)plane_pitch, 0, d_dst, d_preclamp,
                                       (long)dst_stride, 0, width, height, 1,
                                       static_cast<const float *>(c->wraw.p) + 7329, c->b3, c->stream));
        return SRCNN_OK;
    }
    StripParams p{};
    p.planes_in = d_planes;
    p.pl_stride = (long)plane_stride;
    p.pl_pitch = (long)plane_pitch;
    p.dst = d_dst;
    p.pre = d_preclamp;
    p.dst_stride = (long)dst_stride;
    p.width = width;
    p.height = height;
    p.row_begin = 0;
    p.row_end = height;
    return run_strip(c, MODE_L3, p, 1);
}

int srcnn_forward_y_unfused_dev(srcnn_ctx *c, const uint8_t *d_src, size_t src_stride, size_t src_frame_pitch,
                                uint8_t *d_dst, size_t dst_stride, size_t dst_frame_pitch, int width,
                                int height, int n_frames, float *d_work)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    if (bad_plane(d_src, src_stride, width, height) || bad_plane(d_dst, dst_stride, width, height) || !d_work ||
        n_frames <= 0)
        return fail(c, SRCNN_ERR_INVALID, "forward_y_unfused_dev: bad arguments");
    static const char *env_plpad = std::getenv("SRCNN_DEBUG_PLPAD");     // experiment: floats added to the plane pitch
    const long pitch = (long)width * height + (env_plpad ? std::atol(env_plpad) : 0);
    if (bad_pitch((size_t)pitch)) return fail(c, SRCNN_ERR_INVALID, "forward_y_unfused_dev: plane too large");
    if (c->mode == SRCNN_MODE_EXACT) {
        HIP_TRY(c, launch_conv99x11_exact(d_src, (long)src_stride, (long)src_frame_pitch, d_work, width, pitch,
                                          32 * pitch, width, height, n_frames,
                                          static_cast<const float *>(c->wraw.p), c->stream));
        HIP_TRY(c, launch_conv55_exact(d_work, width, pitch, 32 * pitch, d_dst, nullptr, (long)dst_stride,
                                       (long)dst_frame_pitch, width, height, n_frames,
                                       static_cast<const float *>(c->wraw.p) + 7329, c->b3, c->stream));
        return SRCNN_OK;
    }
    StripParams p{};
    p.src = d_src;
    p.src_stride = (long)src_stride;
    p.src_frame_pitch = (long)src_frame_pitch;
    p.planes_out = d_work;
    p.pl_stride = width;
    p.pl_pitch = pitch;
    p.pl_frame_pitch = 32 * pitch;
    p.width = width;
    p.height = height;
    p.row_begin = 0;
    p.row_end = height;
    if ((rc = run_strip(c, MODE_L12, p, n_frames))) return rc;
    StripParams q{};
    q.planes_in = d_work;
    q.pl_stride = width;
    q.pl_pitch = pitch;
    q.pl_frame_pitch = 32 * pitch;
    q.dst = d_dst;
    q.dst_stride = (long)dst_stride;
    q.dst_frame_pitch = (long)dst_frame_pitch;
    q.width = width;
    q.height = height;
    q.row_begin = 0;
    q.row_end = height;
    return run_strip(c, MODE_L3, q, n_frames);
}

int srcnn_forward_y_dev(srcnn_ctx *c, const uint8_t *d_src, size_t src_stride, size_t src_frame_pitch,
                        uint8_t *d_dst, size_t dst_stride, size_t dst_frame_pitch, int width, int height,
                        int n_frames, float *d_preclamp)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    if (bad_plane(d_src, src_stride, width, height) || bad_plane(d_dst, dst_stride, width, height) ||
        n_frames <= 0)
        return fail(c, SRCNN_ERR_INVALID, "forward_y_dev: bad arguments");
    // every output pixel reads a 13x13 input window that other workgroups may already have overwritten
    if (ranges_overlap(d_src, span_elems(src_stride, src_frame_pitch, width, height, n_frames), d_dst,
                       span_elems(dst_stride, dst_frame_pitch, width, height, n_frames)))
        return fail(c, SRCNN_ERR_INVALID, "forward_y_dev: src and dst overlap (the path cannot run in place)");
    // (a pre-clamp request in REFBYTES mode wants the REFERENCE's float too: the exact kernels deliver both)
    if (c->mode == SRCNN_MODE_EXACT || ((c->mode == SRCNN_MODE_REFBYTES || c->mode == SRCNN_MODE_REFBYTES16) && d_preclamp)) {
        // frame by frame through ONE 32-plane workspace (128 B/pixel), whatever the batch size
        const long pitch = (long)width * height;
        if ((rc = reserve(c, c->planes, (size_t)32 * pitch * 4))) return rc;
        float *work = static_cast<float *>(c->planes.p);
        for (int f = 0; f < n_frames; ++f) {
            HIP_TRY(c, launch_conv99x11_exact(d_src + (size_t)f * src_frame_pitch, (long)src_stride, 0, work, width,
                                              pitch, 0, width, height, 1, static_cast<const float *>(c->wraw.p),
                                              c->stream));
            HIP_TRY(c, launch_conv55_exact(work, width, pitch, 0, d_dst + (size_t)f * dst_frame_pitch,
                                           d_preclamp ? d_preclamp + (size_t)f * dst_frame_pitch : nullptr,
                                           (long)dst_stride, 0, width, height, 1,
                                           static_cast<const float *>(c->wraw.p) + 7329, c->b3, c->stream));
        }
        return SRCNN_OK;
    }
    // the seam scratch of a launch grows with its frames: frames_per_launch() bounds it
    const int kMaxFrames = frames_per_launch(c, width, height, n_frames);
    for (int f0 = 0; f0 < n_frames; f0 += kMaxFrames) {
        StripParams p{};
        p.src = d_src + (size_t)f0 * src_frame_pitch;
        p.src_stride = (long)src_stride;
        p.src_frame_pitch = (long)src_frame_pitch;
        p.dst = d_dst + (size_t)f0 * dst_frame_pitch;
        p.pre = d_preclamp ? d_preclamp + (size_t)f0 * dst_frame_pitch : nullptr;
        p.dst_stride = (long)dst_stride;
        p.dst_frame_pitch = (long)dst_frame_pitch;
        p.width = width;
        p.height = height;
        p.row_begin = 0;
        p.row_end = height;
        const bool refbytes = c->mode == SRCNN_MODE_REFBYTES || c->mode == SRCNN_MODE_REFBYTES16;      // (kMaxFrames is 1)
        // frames per fix-up: the work lists' 32-bit pixel codes (frame * height + y) * width + x must not wrap
        const int fix_batch = (int)std::max<unsigned long long>(
            1ull, std::min<unsigned long long>(FIX_BATCH_FRAMES, 0xffffffffull / ((unsigned long long)width * height)));
        const int batch0 = f0 - f0 % fix_batch;
        if ((rc = run_strip(c, MODE_FUSED, p, std::min(kMaxFrames, n_frames - f0), refbytes ? f0 - batch0 : 0,
                            refbytes ? std::min(fix_batch, n_frames - batch0) : 1)))
            return rc;
    }
    return SRCNN_OK;
}

int srcnn_forward_y_rows_dev(srcnn_ctx *c, const uint8_t *d_src, size_t src_stride, int src_row0,
                             uint8_t *d_dst, size_t dst_stride, int dst_row0, int width, int height,
                             int row_begin, int row_end)
{
    BIND(c);
    int rc = SRCNN_OK;
    (void)rc;
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    if (bad_plane(d_src, src_stride, width, height) || bad_plane(d_dst, dst_stride, width, height) ||
        row_begin < 0 || row_end > height || row_begin >= row_end ||
        src_row0 > std::max(0, row_begin - 6) || dst_row0 > row_begin || src_row0 < 0 || dst_row0 < 0)
        return fail(c, SRCNN_ERR_INVALID, "forward_y_rows_dev: bad arguments");
    if (c->mode == SRCNN_MODE_EXACT) return fail(c, SRCNN_ERR_STATE, "row stripes are MFMA-mode only");
    StripParams p{};
    p.src = d_src;
    p.src_stride = (long)src_stride;
    p.src_row0 = src_row0;
    p.dst = d_dst;
    p.dst_stride = (long)dst_stride;
    p.dst_row0 = dst_row0;
    p.width = width;
    p.height = height;
    p.row_begin = row_begin;
    p.row_end = row_end;
    return run_strip(c, MODE_FUSED, p, 1);
}

int srcnn_forward_y_rows_halo_dev(srcnn_ctx *c, const uint8_t *d_src, size_t src_stride, int src_row0, int src_rows,
                                  const uint8_t *d_halo_top, const uint8_t *d_halo_bot, size_t halo_stride,
                                  uint8_t *d_dst, size_t dst_stride, int dst_row0, int width, int height,
                                  int row_begin, int row_end)
{
    BIND(c);
    if (!has_model(c)) return fail(c, SRCNN_ERR_STATE, "%s", kNoModel);
    const int src_row1 = src_row0 + src_rows;
    if (bad_plane(d_src, src_stride, width, height) || bad_plane(d_dst, dst_stride, width, height) || src_rows <= 0 ||
        row_begin < 0 || row_end > height || row_begin >= row_end || src_row0 < 0 || src_row1 > height ||
        dst_row0 > row_begin || dst_row0 < 0 || ((d_halo_top || d_halo_bot) && halo_stride < (size_t)width) ||
        halo_stride >= ((size_t)1 << 30))
        return fail(c, SRCNN_ERR_INVALID, "forward_y_rows_halo_dev: bad arguments");
    // the 13x13 receptive field of rows [row_begin, row_end) must lie in top | src | bot
    const int need0 = std::max(0, row_begin - kHaloRows), need1 = std::min(height, row_end + kHaloRows);
    if ((need0 < src_row0 && (!d_halo_top || src_row0 < kHaloRows || need0 < src_row0 - kHaloRows)) ||
        (need1 > src_row1 && (!d_halo_bot || need1 > src_row1 + kHaloRows)))
        return fail(c, SRCNN_ERR_INVALID, "forward_y_rows_halo_dev: rows [%d,%d) need input rows [%d,%d); src holds [%d,%d) and "
                                          "the halo buffers 6 rows either side", row_begin, row_end, need0, need1, src_row0, src_row1);
    if (c->mode != SRCNN_MODE_MFMA && c->mode != SRCNN_MODE_REFBYTES)
        return fail(c, SRCNN_ERR_STATE, "separate halo buffers are read by the float32 MFMA kernel only (SRCNN_MODE_MFMA / REFBYTES)");
    StripParams p{};
    p.src = d_src;
    p.src_stride = (long)src_stride;
    p.src_row0 = src_row0;
    p.src_row1 = src_row1;
    // a side the launch reads nothing from keeps a null pointer: with both null this is srcnn_forward_y_rows_dev
    p.src_top = need0 < src_row0 ? d_halo_top : nullptr;
    p.src_bot = need1 > src_row1 ? d_halo_bot : nullptr;
    p.halo_stride = (long)halo_stride;
    p.dst = d_dst;
    p.dst_stride = (long)dst_stride;
    p.dst_row0 = dst_row0;
    p.width = width;
    p.height = height;
    p.row_begin = row_begin;
    p.row_end = row_end;
    return run_strip(c, MODE_FUSED, p, 1);
}

/* ------------------------- host-buffer entry points ------------------------- */

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
    static const char *env_bands = std::getenv("SRCNN_DEBUG_BANDS");
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

}  // extern "C"

namespace {

// The per-call tables of the reference surface (src/srcnn.cpp:609, :627: the same const arrays on every call).
// Layers 1-2 of the model are replaced, layer 3 of any loaded model is kept (and the other way round for layer 3);
// tables equal to the uploaded ones are not packed or uploaded again.
int use_layers12(srcnn_ctx *c, const float *kernel99, const float *bias99, const float *kernel11, const float *bias11)
{
    const float *hr = c->host_raw.data();
    const bool same = c->has_l12 && !std::memcmp(hr, bias99, 64 * 4) && !std::memcmp(hr + 64, kernel99, 5184 * 4) &&
                      !std::memcmp(hr + 5248, bias11, 32 * 4) && !std::memcmp(hr + 5280, kernel11, 2048 * 4);
    if (same) return SRCNN_OK;
    const std::vector<float> w3(hr + 7329, hr + 8129);      // upload_weights rewrites host_raw
    const int rc = upload_weights(c, kernel99, bias99, kernel11, bias11, w3.data(), c->b3);
    if (rc == SRCNN_OK) c->has_l12 = true;
    return rc;
}
int use_layer3(srcnn_ctx *c, const float *kernel, float bias)
{
    const float *hr = c->host_raw.data();
    if (c->has_l3 && hr[7328] == bias && !std::memcmp(hr + 7329, kernel, 800 * 4)) return SRCNN_OK;
    const std::vector<float> raw(c->host_raw);               // upload_weights rewrites host_raw
    const int rc = upload_weights(c, raw.data() + 64, raw.data(), raw.data() + 5280, raw.data() + 5248, kernel, bias);
    if (rc == SRCNN_OK) c->has_l3 = true;
    return rc;
}

}  // namespace

extern "C" {

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
    HIP_TRY(c, hipStreamSynchronize(c->stream));          // queued work may still use it
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

constexpr int kHalo = 6;       // 4 input rows of the 9x9 layer + 2 feature rows of the 5x5 layer

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

// The streams and events of the striped step, and the link to the neighbouring devices: peer access is asked for ONCE and the
// answer kept -- a refused link still works (hipMemcpyPeerAsync then stages through host memory) but is not the xGMI path
// BASELINE configs[3] names, so the context says so (srcnn_halo_transport(), srcnn_last_error()).
int stripe_setup(srcnn_ctx *const *ctxs, int n_ctx, int k)
{
    srcnn_ctx *c = ctxs[k];
    if (c->halo_stream) return SRCNN_OK;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->halo_stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&c->halo_ready, hipEventDisableTiming));
    HIP_TRY(c, hipEventCreateWithFlags(&c->bands_done, hipEventDisableTiming));
    for (int i = 0; i < srcnn_ctx::kHaloSets; ++i) HIP_TRY(c, hipEventCreateWithFlags(&c->halo_free[i], hipEventDisableTiming));
    c->halo_transport = 1;
    static const char *env_staged = std::getenv("SRCNN_DEBUG_HALO_STAGED");      // test knob: take the no-peer-access path
    if (env_staged && std::atoi(env_staged)) c->halo_transport = 3;
    for (int n : {k - 1, k + 1}) {
        if (n < 0 || n >= n_ctx || ctxs[n]->device == c->device) continue;
        int can = 0;
        hipError_t e = hipDeviceCanAccessPeer(&can, c->device, ctxs[n]->device);
        if (e == hipSuccess && can) {
            e = hipDeviceEnablePeerAccess(ctxs[n]->device, 0);
            if (e == hipErrorPeerAccessAlreadyEnabled) { (void)hipGetLastError(); e = hipSuccess; }
        }
        if (e == hipSuccess && can) {
            c->halo_transport = std::max(c->halo_transport, 2);
        } else {
            (void)hipGetLastError();
            c->halo_transport = 3;
            (void)fail(c, SRCNN_OK, "row stripes: device %d has no peer access to device %d (%s): halo rows are staged through host "
                                    "memory, not copied over xGMI", c->device, ctxs[n]->device,
                       e == hipSuccess ? "hipDeviceCanAccessPeer says no" : hipGetErrorString(e));
        }
    }
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

}  // extern "C"
