// srcnn_exact.hip -- vector-ALU kernels that reproduce the reference's
// arithmetic EXACTLY (rounded f32 multiply, then rounded f32 add, in the
// reference's summation order; double 25-term sums in layer 3), so their
// results are bit-identical to the reference CPU/OpenMP path.
//
// They serve (a) the per-filter entry points Convolution99 / Convolution11
// (src/srcnn.cpp:92-140, :151-178), which the reference's CLI never calls and
// which are one-filter-per-call by signature, and (b) SRCNN_MODE_EXACT of the
// whole path.  The MFMA kernels (srcnn_mfma.hip) are the fast path.
//
// This translation unit MUST be compiled with -ffp-contract=off (the build
// does so); the pragma below is a second line of defence.  No v_fma / v_mac /
// v_fmac may appear in these kernels' ISA (checked by tests/test_abi.py).
#include "srcnn_kernels.h"

#include <type_traits>

#pragma clang fp contract(off)

namespace srcnn {

__device__ __forceinline__ int clampi_e(int v, int lo, int hi) { return min(max(v, lo), hi); }

// ---- Convolution99: one 9x9 filter --------------------------------------
__global__ __launch_bounds__(256) void conv99_exact_kernel(const uint8_t *__restrict__ src, long sstride,
                                                           float *__restrict__ dst, long dstride,
                                                           int w, int h,
                                                           const float *__restrict__ kernel, float bias)
{
    __shared__ float kw[81];
    if (threadIdx.x < 81) kw[threadIdx.x] = kernel[threadIdx.x];
    __syncthreads();
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (col >= w || row >= h) return;
    float temp = 0.f;
    for (int i = 0; i < 9; ++i) {
        const uint8_t *sr = src + (long)clampi_e(row + i - 4, 0, h - 1) * sstride;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const float pr = kw[i * 9 + j] * (float)sr[clampi_e(col + j - 4, 0, w - 1)];
            temp = temp + pr;
        }
    }
    temp = temp + bias;
    temp = (temp < 0) ? 0.f : temp;
    dst[(long)row * dstride + col] = temp;
}

// ---- Convolution11: one output channel of the 1x1 layer -------------------
// "Direct LDS-tiled pointwise kernel": the 64 weights sit in LDS (broadcast
// reads); every plane read is a coalesced 256-B row segment per wave.
__global__ __launch_bounds__(256) void conv11_exact_kernel(const float *__restrict__ planes, long stride,
                                                           long pitch, float *__restrict__ dst, long dstride,
                                                           int w, int h,
                                                           const float *__restrict__ kernel, float bias)
{
    __shared__ float kw[64];
    if (threadIdx.x < 64) kw[threadIdx.x] = kernel[threadIdx.x];
    __syncthreads();
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (col >= w || row >= h) return;
    const float *s = planes + (long)row * stride + col;
    float temp = 0.f;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) {
        const float pr = s[(long)i * pitch] * kw[i];
        temp = temp + pr;
    }
    temp = temp + bias;
    temp = (temp < 0) ? 0.f : temp;
    dst[(long)row * dstride + col] = temp;
}

// ---- Convolution99x11, exact ---------------------------------------------
// One feature position per lane.  The 64 layer-1 sums of the position live in registers and advance tap by tap; layer 2
// then sums its inputs in ascending channel order (src/srcnn.cpp:312-315) -- same products, same order, same roundings as
// the reference.  Weights are wave-uniform and come through the scalar cache: weights = b1[64] | W1[64][81] | b2[32] |
// W2[32][64] | b3 | W3 (convdata.h order, 8,129 floats), then W2 transposed to [64][32] and W1 transposed to [81][64], so that
// what one step of either loop needs is contiguous (+ one tap of zeros behind it: the fix-up kernel fetches a tap ahead).
// The weight tables are read through the CONSTANT address space: wave-uniform addresses there become scalar loads (s_load,
// through the scalar cache, no vector-memory latency in the channel loop) whatever else the kernel stores to global memory --
// for a plain global pointer the compiler keeps scalar loads only while it can prove no store of the kernel may alias them,
// which the fix-up kernel's byte stores defeat: 16 vector loads per channel, each waited for at once, 3 x the time.
typedef const __attribute__((address_space(4))) float *cfloat_p;
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ cfloat_p as_constant(const float *p) { return (cfloat_p)p; }

// The fix-up kernel's layers 1-2 (exact_layer1_half / exact_layer2_quarter below) read the luma window from LDS --
// win[i * PITCH + j] is the (already border-replicated) value under tap (i, j) -- and run TAP-OUTER: the layer-1 sums of a
// position advance together, one luma value against the weights of its tap (W1 transposed [81][64] at wraw + 10177) -- per
// channel the same rounded products added in the same order as the reference's loop (src/srcnn.cpp:283-305), but independent
// chains instead of one dependent chain per channel, two channels per packed instruction (v_pk_mul_f32 / v_pk_add_f32 with the
// weights as an SGPR pair: profiles/r03/pk_f32_probe.txt), and no 81-register window.  History of that loop, same-box steps on a
// 3840x2160 plane: register-window form 328 us -> tap-outer, one scalar load at a time 334 -> a tap's four loads issued together
// 252 (round 3) -> two lanes per position, loads one tap ahead, layer 2 packed too: 232 (profiles/r04/fix_apply_ab.txt).
typedef float f32x16 __attribute__((ext_vector_type(16)));
// (the tables these loads read are only DWORD aligned -- W1 transposed starts at float 10177 of the table, a layer-3 channel
// every 25 floats: the pointee types say so, an s_load_dwordx16 / x8 needs no more)
typedef f32x16 f32x16_a4 __attribute__((aligned(4)));
typedef const __attribute__((address_space(4))) f32x16_a4 *cvec16_p;

// The same arithmetic for HALF a feature position: the 32 layer-1 channels [32 hh, 32 hh + 32) of the lane's position (hh is
// wave-uniform: the weights stay scalar operands); layer 2 then gives the lane 16 of the 32 output chains (exact_layer2_quarter).
// Two lanes per position halve the time one work item of fix_apply_kernel takes; every chain keeps the reference's order
// (src/srcnn.cpp:288-317).
// Scalar loads return out of order, so the only wait for one is lgkmcnt(0) -- which also waits for every load issued since.
// A tap's weights are therefore fetched ONE TAP AHEAD and the wait is pinned (an empty asm that reads the registers) in front of
// the next fetch: [wait for tap t] [fetch tap t + 1] [32 packed operations of tap t] -- the fetch has the whole tap to arrive.
// With 64 channels per lane that needs 128 scalar registers; with 32 it fits.  The luma values of a window row are read from
// LDS one ROW ahead for the same reason (LDS reads share the counter).
// The luma window arrives with the image border ALREADY REPLICATED (the staging loops clamp the coordinates they read): `win`
// points at the value under tap (0, 0) of the lane's position, tap (i, j) lies at win[i * PITCH + j] -- one ds_read_b32 with an
// immediate offset per value and ONE address add per window row.  Round 4 clamped per tap in here: 15 address instructions and 9
// register moves per window row beside its 288 packed operations; the two-rows-per-iteration form below has neither.
template <int PITCH>
__device__ __forceinline__ void exact_layer1_half(const float *win, const float *wraw, int hh, f32x2 (&acc)[16])
{
    const cfloat_p wr = as_constant(wraw);
    const cfloat_p b1 = wr + 32 * hh;
    cvec16_p wrow = (cvec16_p)(wr + 10177 + 32 * hh);     // tap t at wrow[4 t], [4 t + 1]; the table is padded by one tap (upload_weights)
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = f32x2{0.f, 0.f};
    float ya[9], yb[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) ya[j] = win[j];
    f32x16 w0c = wrow[0], w1c = wrow[1];
    // one window row: its 9 taps from ycur[], the next row's values fetched into ynext[] meanwhile
    auto row = [&](const float (&ycur)[9], float (&ynext)[9], const float *next) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            asm volatile("" ::"s"(w0c), "s"(w1c));                             // tap (i, j)'s weights have arrived
            const f32x16 w0n = wrow[4 * (j + 1)], w1n = wrow[4 * (j + 1) + 1];   // tap (i, j + 1), or (i + 1, 0): rows are contiguous
            if (j == 0) {
#pragma unroll
                for (int jj = 0; jj < 9; ++jj) ynext[jj] = next[jj];
            }
            __builtin_amdgcn_sched_barrier(0);
            const f32x2 yy = {ycur[j], ycur[j]};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const f32x2 a0 = {w0c[2 * c], w0c[2 * c + 1]}, a1 = {w1c[2 * c], w1c[2 * c + 1]};
                const f32x2 p0 = a0 * yy, p1 = a1 * yy;
                acc[c] = acc[c] + p0;
                acc[8 + c] = acc[8 + c] + p1;
            }
            __builtin_amdgcn_sched_barrier(0);
            w0c = w0n;
            w1c = w1n;
        }
        wrow += 4 * 9;
    };
#pragma unroll 1
    for (int i = 0; i < 8; i += 2) {
        row(ya, yb, win + (i + 1) * PITCH);
        row(yb, ya, win + (i + 2) * PITCH);
    }
    row(ya, yb, win + 8 * PITCH);          // (the last row re-reads itself: nothing lies behind it)
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const f32x2 bv = {b1[2 * c], b1[2 * c + 1]};
        f32x2 a = acc[c] + bv;
        a.x = (a.x < 0) ? 0.f : a.x;
        a.y = (a.y < 0) ? 0.f : a.y;
        acc[c] = a;
    }
}
// Layer 2: a lane folds 32 consecutive input channels [in0, in0 + 32) -- a(i) delivers channel in0 + i: its own layer-1
// activations from registers, or its partner lane's through LDS -- into the 16 output chains [16 hh, 16 hh + 16) it owns.  Both
// lanes of a position work at the same time: the chains of the hh = 0 lane take inputs 0-31 from its registers, then 32-63 from
// the partner; those of the hh = 1 lane 0-31 from the partner, then its own.  Every chain still sees its 64 inputs in ascending
// order (src/srcnn.cpp:312-315).  One 16-float scalar load per input channel, fetched one ahead, two output channels per packed
// instruction with the weights as a scalar-register pair.  (Round 4 split layer 2 by INPUT channel: each lane carried all 32
// chains through its 32 inputs, the hh = 1 lane continuing where the hh = 0 lane stopped -- the two halves ran one after the
// other, 2,048 packed instructions on an item's critical path instead of 1,024, and half the workgroup's waves idle meanwhile:
// fix_apply 166 -> 162 us on a 3840x2160 plane, profiles/r05/fix_apply_ab.txt step 5.)
template <class In>
__device__ __forceinline__ void exact_layer2_quarter(In a, const float *wraw, int hh, int in0, f32x2 (&r)[8])
{
    const cvec16_p w2t = (cvec16_p)(as_constant(wraw) + 8129 + in0 * 32 + 16 * hh);      // input i at w2t[2 i] (a row of W2T is 32 floats)
    f32x16 wc = w2t[0];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
        asm volatile("" ::"s"(wc));
        // (the fetch behind the last channel reads into the W1T table that follows: in bounds)
        const f32x16 wn = w2t[2 * (i + 1)];
        __builtin_amdgcn_sched_barrier(0);
        const float ai = a(i);
        const f32x2 aa = {ai, ai};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 wa = {wc[2 * k], wc[2 * k + 1]};
            const f32x2 p0 = wa * aa;
            r[k] = r[k] + p0;
        }
        __builtin_amdgcn_sched_barrier(0);
        wc = wn;
    }
}

// ---- the same two loops with the weights in LDS (round 6: fix_apply_kernel) --------------------------------------------
// The scalar-load form above fetches a tap's weights ONE tap ahead -- all the scalar registers allow -- and the tables (W1
// transposed 20.7 KB + W2 transposed 8 KB) do not fit the 16 KB scalar cache: every tap is an L2 round trip of 300-500 cycles
// behind ~160 cycles of arithmetic.  Five workgroups per CU hide that at full load and nothing hides it on a small plane, whose
// few hundred items are ONE round of the draw: an item took ~45 us whatever the occupancy (fix_apply 49.5 us on a 1920x1080 plane
// for 646 items, 101 us on 3840x2160 for 2,665: 75 % of the vector ALU's issue rate).  Here a workgroup stages both tables in LDS
// once (29 KB) and every lane reads a tap's 32 weights as eight broadcast ds_read_b128 (all lanes one address: no bank conflict),
// two taps ahead -- LDS returns in order, so the waits are counted, not drained.  Same products, same order, same roundings.
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PITCH>
__device__ __forceinline__ void exact_layer1_half_lds(const float *win, const float *s_w1t /* LDS [82][64] */, const float *b1g, int hh, f32x2 (&acc)[16])
{
    const cfloat_p b1 = as_constant(b1g) + 32 * hh;
    const f32x4 *wrow = reinterpret_cast<const f32x4 *>(s_w1t + 32 * hh);      // tap t at wrow[16 t .. 16 t + 7]
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = f32x2{0.f, 0.f};
    float ya[9], yb[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) ya[j] = win[j];
    f32x4 wc[8], wn[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) wc[q] = wrow[q];
    auto row = [&](const float (&ycur)[9], float (&ynext)[9], const float *next) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
#pragma unroll
            for (int q = 0; q < 8; ++q) wn[q] = wrow[16 * (j + 1) + q];            // tap (i, j + 1), or (i + 1, 0); the table is padded by one tap
            if (j == 0) {
#pragma unroll
                for (int jj = 0; jj < 9; ++jj) ynext[jj] = next[jj];
            }
            const f32x2 yy = {ycur[j], ycur[j]};
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                const f32x2 a = {wc[c >> 1][2 * (c & 1)], wc[c >> 1][2 * (c & 1) + 1]};
                const f32x2 p0 = a * yy;
                acc[c] = acc[c] + p0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) wc[q] = wn[q];
        }
        wrow += 16 * 9;
    };
#pragma unroll 1
    for (int i = 0; i < 8; i += 2) {
        row(ya, yb, win + (i + 1) * PITCH);
        row(yb, ya, win + (i + 2) * PITCH);
    }
    row(ya, yb, win + 8 * PITCH);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const f32x2 bv = {b1[2 * c], b1[2 * c + 1]};
        f32x2 a = acc[c] + bv;
        a.x = (a.x < 0) ? 0.f : a.x;
        a.y = (a.y < 0) ? 0.f : a.y;
        acc[c] = a;
    }
}
template <class In>
__device__ __forceinline__ void exact_layer2_quarter_lds(In a, const float *s_w2t /* LDS [65][32] */, int hh, int in0, f32x2 (&r)[8])
{
    const f32x4 *w2t = reinterpret_cast<const f32x4 *>(s_w2t + in0 * 32 + 16 * hh);      // input i at w2t[8 i .. 8 i + 3]
    f32x4 wc[4], wn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wc[q] = w2t[q];
#pragma unroll
    for (int i = 0; i < 32; ++i) {
#pragma unroll
        for (int q = 0; q < 4; ++q) wn[q] = w2t[8 * (i + 1) + q];                  // (behind the last channel: the table's padding row)
        const float ai = a(i);
        const f32x2 aa = {ai, ai};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const f32x2 wa = {wc[k >> 1][2 * (k & 1)], wc[k >> 1][2 * (k & 1) + 1]};
            const f32x2 p0 = wa * aa;
            r[k] = r[k] + p0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) wc[q] = wn[q];
    }
}

// The whole-plane kernel keeps the register form: the 81 window values of the lane's pixel in registers, one layer-1 channel
// after the other, each activation folded into the 32 layer-2 sums as soon as it exists.  (Measured on one box, 3840x2160:
// this form 1.90 ms, the tap-outer form above from an LDS tile 2.13 ms packed / 2.06 ms unpacked -- the opposite order of
// the fix-up kernel's 328 / 252 / 277 us, whose lanes gather scattered windows and share the CU with fewer waves.)
__device__ __forceinline__ void exact_layers12(const float (&px)[81], const float *weights_, const float *w2t_, float (&r)[32])
{
    const cfloat_p weights = as_constant(weights_), w2t = as_constant(w2t_);
    const cfloat_p b1 = weights, w1 = weights + 64, b2 = w1 + 64 * 81;
#pragma unroll
    for (int k = 0; k < 32; ++k) r[k] = 0.f;
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
        const cfloat_p wi = w1 + i * 81;
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 81; ++q) {       // (products computed nine at a time ahead of their adds: 2.80 -> 3.8 ms, profiles/r03)
            const float pr = wi[q] * px[q];
            a = a + pr;
        }
        a = a + b1[i];
        a = (a < 0) ? 0.f : a;
        // (layer 2 packed two output channels per instruction, as the fix-up kernel does it: 2.79 -> 2.83 ms here, round 4 -- this
        // kernel is bound by its dependent layer-1 chain, not by instruction issue)
        const cfloat_p w2i = w2t + i * 32;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float pr = a * w2i[k];
            r[k] = r[k] + pr;
        }
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        float v = r[k] + b2[k];
        r[k] = (v < 0) ? 0.f : v;
    }
}

__global__ __launch_bounds__(256) void conv99x11_exact_kernel(const uint8_t *__restrict__ src, long sstride,
                                                              long src_frame_pitch,
                                                              float *__restrict__ planes, long stride,
                                                              long pitch, long frame_pitch, int w, int h,
                                                              const float *__restrict__ weights,
                                                              const float *__restrict__ w2t,
                                                              int row0, int row1, int src_row0, int pl_row0)
{
    // rows [row0, row1) of the map; src points at image row src_row0, planes at map row pl_row0 (whole planes: 0, h, 0, 0)
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = blockIdx.z;
    if (row >= row1) return;
    const uint8_t *sf = src + (long)frame * src_frame_pitch;
    float px[81];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint8_t *sr = sf + (long)(clampi_e(row + i - 4, 0, h - 1) - src_row0) * sstride;
#pragma unroll
        for (int j = 0; j < 9; ++j) px[i * 9 + j] = (float)sr[clampi_e(col + j - 4, 0, w - 1)];
    }
    float r[32];
    exact_layers12(px, weights, w2t, r);
    if (col >= w) return;
    float *o = planes + (long)frame * frame_pitch + (long)(row - pl_row0) * stride + col;
#pragma unroll
    for (int k = 0; k < 32; ++k) o[(long)k * pitch] = r[k];
}

// ---- Convolution55, exact ---------------------------------------------------
// float product, double 25-term sum per channel, float running sum over the
// channels (src/srcnn.cpp:218-240).  A workgroup owns a 64x4 pixel tile; each
// channel's 68x8 window (replicate border applied while loading) is staged in LDS,
// double-buffered, so every plane element is fetched from HBM/L2 about once instead
// of 25 times and the taps are LDS reads.
__global__ __launch_bounds__(256) void conv55_exact_kernel(const float *__restrict__ planes, long stride,
                                                           long pitch, long frame_pitch,
                                                           uint8_t *__restrict__ dst, float *__restrict__ pre,
                                                           long dstride, long dst_frame_pitch, int w, int h,
                                                           const float *__restrict__ kernel, float bias,
                                                           int out_row0, int out_row1, int pl_row0, int pl_row1, int dst_row0)
{
    // output rows [out_row0, out_row1); planes holds map rows [pl_row0, pl_row1), dst / pre point at image row dst_row0 (whole
    // planes: 0, h, 0, h, 0).  A tile's window may reach past the rows the launch produces (its last tile is 4 rows tall whatever
    // out_row1 is): such rows feed no pixel that is stored, and are clamped into the map rows that exist.
    constexpr int TW = 64, TH = 4, WW = TW + 4, WH = TH + 4, WN = WW * WH;   // 68 x 8 window
    __shared__ float win[2][WN];
    // (the weights are wave-uniform: they come through the scalar cache; from LDS: 2 % slower)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col0 = blockIdx.x * TW, row0 = out_row0 + blockIdx.y * TH;
    const int col = col0 + tx, row = row0 + ty;
    const int frame = blockIdx.z;
    const float *pf = planes + (long)frame * frame_pitch;
    // window elements this thread stages (WN = 544 = 2 x 256 + 32)
    long off[3];
    int idx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int e = threadIdx.x + 256 * q;
        idx[q] = e < WN ? e : -1;
        const int wy = e / WW, wx = e % WW;
        off[q] = (long)(clampi_e(clampi_e(row0 + wy - 2, 0, h - 1), pl_row0, pl_row1 - 1) - pl_row0) * stride + clampi_e(col0 + wx - 2, 0, w - 1);
    }
    auto stage = [&](int ch, int buf) {
        const float *pl = pf + (long)ch * pitch;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            if (idx[q] >= 0) win[buf][idx[q]] = pl[off[q]];
    };
    stage(0, 0);
    __syncthreads();
    float temp = 0.f;
    for (int i = 0; i < 32; ++i) {
        if (i + 1 < 32) stage(i + 1, (i + 1) & 1);
        const float *wv = &win[i & 1][ty * WW + tx];
        double tp = 0.0;
        // the channel's 25 weights as three scalar loads issued together (see exact_layer1_half): 893 -> 878 us per 3840x2160 plane
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        typedef f32x8 f32x8_a4 __attribute__((aligned(4)));
        const cfloat_p kc = as_constant(kernel) + i * 25;
        const f32x16 wa = *(cvec16_p)kc;
        const f32x8 wb = *(const __attribute__((address_space(4))) f32x8_a4 *)(kc + 16);
        const float wc = kc[24];
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const int t = m * 5 + n;
                const float wt = t < 16 ? wa[t < 16 ? t : 0] : t < 24 ? wb[t >= 16 && t < 24 ? t - 16 : 0] : wc;
                const float pr = wt * wv[m * WW + n];
                tp = tp + (double)pr;
            }
        temp = (float)((double)temp + tp);
        __syncthreads();
    }
    if (col >= w || row >= out_row1) return;
    temp = temp + bias;
    const long o = (long)frame * dst_frame_pitch + (long)(row - dst_row0) * dstride + col;
    if (pre) pre[o] = temp;
    int q = (int)temp;                       // truncation toward zero, src/srcnn.cpp:238
    q = clampi_e(q, 0, 255);
    dst[o] = (uint8_t)q;
}

// ---- SRCNN_MODE_REFBYTES: the reference's bytes at (nearly) MFMA speed ------------------------------------------------
// The MFMA path's pre-truncation value v differs from the reference's by rounding noise (measured <= 4.4e-4 on 54 MPix of
// varied content, profiles/r03/fixup_margin.txt; <= 6.4e-4 on adversarially searched windows, profiles/r04), so its byte can
// differ from the reference's only where v lies within that distance of an integer -- the store truncates
// (src/srcnn.cpp:238-240).  The fused strip kernel therefore writes, beside every output byte, a FLAG byte: 0, or a code
// 1..254 for (v - rint(v)) in [-delta, +delta] (StripParams::flag, fix_delta).  Three kernels then finish the plane:
//   fix_collect_kernel  cuts the rows of the launch into 12 x 12 TILES and turns the flag plane into work items: a tile with
//                       >= FIX_DENSE_MIN flagged pixels becomes ONE dense item (flat or periodic content flags every pixel of a
//                       region: recomputing the whole tile costs 3 x 128 feature positions, as much as 15 scattered pixels),
//                       the flagged pixels of the other tiles go to the scattered list, FIX_GROUP to an item;
//   fix_apply_kernel    recomputes those pixels in the reference's arithmetic (the per-lane code of the exact kernels above:
//                       rounded multiply then rounded add, double 25-term sums), two lanes per feature position of the 5 x 5
//                       window, 128 positions per item, stores the byte where it differs, records the largest
//                       |v_mfma - v_ref| it meets (a ~0.3 % random sample of the plane);
//   fix_rerun_kernel    the safety net of the mode, ON THE DEVICE: queued behind fix_apply unconditionally, every workgroup of
//                       it compares that deviation with delta / 2 and returns at once unless it is exceeded; then the launch
//                       recomputes EVERY tile of the fix-up's rows as a dense tile -- the
//                       reference's arithmetic on every pixel, no threshold involved.  No host read, no workspace: a queued
//                       stream of frames is never stalled (round 4 synchronised the host per fix-up and re-ran through a
//                       32-plane workspace shared between streams).
// Result: the reference's byte in every pixel whose |v_mfma - v_ref| <= delta -- all of them, on every input tried
// (tests/test_gpu_refbytes.py), and wherever the monitored sample says the margin is gone, in every pixel.
// (Round 5 also built the collect step INTO fix_apply -- every workgroup scanning 12 x 96 segments of the flag plane into an LDS
// list of its own: one launch less, but each of the 1,280 workgroups then ends on a partial group of < FIX_GROUP pixels at the
// full cost of one, 28 % more item executions on a 3840x2160 plane: 230 us against 175.  Not kept; profiles/r05/fix_apply_ab.txt.)
constexpr int FIX_TILE = 12, FIX_POS = FIX_TILE + 4, FIX_GROUP = 5, FIX_DENSE_MIN = 16;
constexpr int FIX_HALF = 128, FIX_SUB_ROWS = 4;         // positions per item; output rows per dense sub-pass
static_assert(FIX_GROUP * 25 <= FIX_HALF && (FIX_SUB_ROWS + 4) * FIX_POS == FIX_HALF && FIX_TILE % FIX_SUB_ROWS == 0, "two lanes per position");
constexpr int FIX_DWIN_W = FIX_POS + 8, FIX_DWIN_H = FIX_SUB_ROWS + 12;        // luma window of a dense sub-pass: 24 x 16

// non-zero bytes of a dword, as a count
__device__ __forceinline__ int nz_bytes(unsigned dw)
{
    return __builtin_popcount((((dw & 0x7f7f7f7fu) + 0x7f7f7f7fu) | dw) & 0x80808080u);
}
__device__ __forceinline__ unsigned *fix_word(unsigned *counters, int k) { return counters + k * FIX_WORD_STRIDE; }

// One workgroup per 12-row band x 768-column segment (64 tiles).  Thread t < 192 owns dword column t of the segment (4 flag
// bytes x 12 rows, read coalesced: 768 B per row) and adds its count to its tile's (t / 3) counter in LDS; tiles are then
// classified, the flagged pixels of the scattered ones collected in an LDS list, and ONE reservation per workgroup made in
// each global list (a returning atomic on one word sustains ~88 per microsecond chip-wide: one per tile would cost a
// 3840x2160 plane 0.6 ms; even one per workgroup is 10 us of it, so the lists come in FIX_REGIONS regions with a word each).  Rows that are not dword aligned (odd strides) are read byte by byte.
constexpr int FIX_SEG_TILES = 64, FIX_SEG_COLS = FIX_SEG_TILES * FIX_TILE;     // 768 columns = 192 dwords
// Workgroups of fix_collect_kernel per list region (workgroup b fills region b % FIX_REGIONS): a region holds what they can
// produce at most -- (FIX_DENSE_MIN - 1) scattered pixels and one dense tile per tile of a segment.
__host__ __device__ inline unsigned fix_region_wgs(int width, int rows, int n_frames)
{
    const int tiles_x = (width + FIX_TILE - 1) / FIX_TILE, bands = (rows + FIX_TILE - 1) / FIX_TILE;
    const int segs = (tiles_x + FIX_SEG_TILES - 1) / FIX_SEG_TILES;
    return ((unsigned)(bands * segs * n_frames) + FIX_REGIONS - 1) / FIX_REGIONS;
}
__global__ __launch_bounds__(256) void fix_collect_kernel(const FixParams p)
{
    __shared__ unsigned s_cnt[FIX_SEG_TILES], s_scat[FIX_SEG_TILES * (FIX_DENSE_MIN - 1)], s_dense[FIX_SEG_TILES];
    __shared__ unsigned s_nscat, s_ndense, s_base_scat, s_base_dense;
    const int tid = threadIdx.x;
    if (tid < FIX_SEG_TILES) s_cnt[tid] = 0;
    if (tid == 0) { s_nscat = 0; s_ndense = 0; }
    __syncthreads();
    const int tiles_x = (p.width + FIX_TILE - 1) / FIX_TILE;
    const int segs = (tiles_x + FIX_SEG_TILES - 1) / FIX_SEG_TILES;
    const int bands = (p.row_end - p.row_begin + FIX_TILE - 1) / FIX_TILE;
    const int frame = blockIdx.x / (bands * segs), bs = blockIdx.x - frame * (bands * segs);
    const int band = bs / segs, seg = bs - band * segs;
    const int y0 = p.row_begin + band * FIX_TILE, nr = min(FIX_TILE, p.row_end - y0);
    const int x = seg * FIX_SEG_COLS + 4 * tid;                  // this thread's 4 columns
    const int tile = tid / 3;                                    // within the segment
    const uint8_t *f0 = p.flag + (long)frame * p.flag_frame_pitch + (long)(y0 - p.dst_row0) * p.dst_stride + x;
    const bool aligned = ((p.dst_stride | (long)(size_t)p.flag) & 3) == 0;      // uniform
    unsigned rows_dw[FIX_TILE];
    unsigned cnt = 0;
    if (tid < FIX_SEG_COLS / 4 && x < p.width) {
#pragma unroll
        for (int r = 0; r < FIX_TILE; ++r) {
            unsigned dw = 0;
            if (r < nr) {
                if (aligned && x + 4 <= p.width) {
                    dw = *reinterpret_cast<const unsigned *>(f0 + (long)r * p.dst_stride);
                } else {
                    for (int b = 0; b < 4; ++b)
                        if (x + b < p.width) dw |= (unsigned)f0[(long)r * p.dst_stride + b] << (8 * b);
                }
            }
            rows_dw[r] = dw;
            cnt += (unsigned)nz_bytes(dw);
        }
        if (cnt) atomicAdd(&s_cnt[tile], cnt);
    }
    __syncthreads();
    if (tid < FIX_SEG_TILES && s_cnt[tid] >= FIX_DENSE_MIN)
        s_dense[atomicAdd(&s_ndense, 1u)] = (unsigned)((frame * bands + band) * tiles_x + seg * FIX_SEG_TILES + tid);
    if (cnt && s_cnt[tile] < FIX_DENSE_MIN) {
        unsigned at = atomicAdd(&s_nscat, cnt);
#pragma unroll
        for (int r = 0; r < FIX_TILE; ++r) {
            unsigned dw = rows_dw[r];
            for (int b = 0; dw; ++b, dw >>= 8)
                if (dw & 0xffu) s_scat[at++] = (unsigned)(frame * p.height + y0 + r) * (unsigned)p.width + (unsigned)(x + b);
        }
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned region = blockIdx.x % FIX_REGIONS;     // neighbouring workgroups reserve on different words
        unsigned *rc = p.counters + FIX_REGION0 + region * FIX_WORD_STRIDE;
        const unsigned rw = fix_region_wgs(p.width, p.row_end - p.row_begin, p.n_frames);
        s_base_scat = region * rw * (FIX_SEG_TILES * (FIX_DENSE_MIN - 1)) + (s_nscat ? atomicAdd(rc, s_nscat) : 0u);
        s_base_dense = region * rw * FIX_SEG_TILES + (s_ndense ? atomicAdd(rc + 32, s_ndense) : 0u);
    }
    __syncthreads();
    for (unsigned i = tid; i < s_nscat; i += 256) p.scat[s_base_scat + i] = s_scat[i];
    for (unsigned i = tid; i < s_ndense; i += 256) p.dense[s_base_dense + i] = s_dense[i];
}

// Items [0, n_dense) are dense tiles, items [n_dense, n_dense + ceil(n_scat / FIX_GROUP)) groups of FIX_GROUP scattered pixels.
// The grid is what the GPU holds at once; workgroups draw items from a shared counter until none is left.
//
// TWO LANES PER FEATURE POSITION.  Lane t works on position q = t & 127 -- tap q % 25 of scattered pixel q / 25, or position
// (q / 16, q % 16) of a dense sub-window -- with hh = t >> 7 (wave-uniform, so the weights stay scalar operands): it computes the
// layer-1 channels [32 hh, 32 hh + 32) of the position (exact_layer1_half) and 16 of its 32 layer-2 output chains
// (exact_layer2_quarter): the two lanes swap their 32 activations through LDS, half at a time, so that each chain sees its 64
// inputs in ascending order (src/srcnn.cpp:312-315).  Round 3 ran one lane per position, 250 per
// item: an item took ~62 us, 3,390 of them over 1,024 resident workgroups = 3.3 rounds, the last one two thirds empty, and on a
// 1920x1080 plane the whole fix-up was ONE item's latency.  Now an item is half as long (5 scattered pixels), the last round
// costs half as much, and the registers a lane no longer needs (32 channel sums instead of 64) buy a fifth workgroup per CU.
// A dense tile's 16 x 16 window goes through in three sub-windows of 8 x 16 positions (rows 0-7, 4-11, 8-15 -> output rows
// 0-3, 4-7, 8-11): 1.5 x the positions of round 3's one pass, on the rare flat / periodic content only.
// RERUN = false: fix_apply_kernel (the work lists, the monitor, the verdict).
// RERUN = true:  fix_rerun_kernel (every tile of the launch as a dense tile, if the verdict asks for it).
template <bool RERUN, bool LDSW = false>
__device__ __forceinline__ void fix_kernel_body(const FixParams &p)
{
    // LDSW: both weight tables of layers 1-2 staged in LDS once per workgroup (exact_layer1_half_lds)
    __shared__ __attribute__((aligned(16))) float s_w1t[LDSW ? 82 * 64 : 4];
    __shared__ __attribute__((aligned(16))) float s_w2t[LDSW ? 65 * 32 : 4];
    if constexpr (LDSW) {
        for (int i = threadIdx.x; i < 81 * 64; i += 256) s_w1t[i] = p.wraw[10177 + i];
        for (int i = 81 * 64 + threadIdx.x; i < 82 * 64; i += 256) s_w1t[i] = 0.f;
        for (int i = threadIdx.x; i < 64 * 32; i += 256) s_w2t[i] = p.wraw[8129 + i];
        for (int i = 64 * 32 + threadIdx.x; i < 65 * 32; i += 256) s_w2t[i] = 0.f;
    }
    __shared__ float Fs[FIX_HALF][33];      // per position: the layer-2 chains between the two halves, then the activations
    // scattered items: the 25-term double sums per (pixel, channel) -- in the luma window's memory, which layers12() is done with
    // by then (with the weights in LDS a workgroup must stay below 160 KB / 3)
    __shared__ __attribute__((aligned(8))) float s_y[FIX_GROUP * 169 > FIX_DWIN_H * FIX_DWIN_W ? FIX_GROUP * 169 : FIX_DWIN_H * FIX_DWIN_W];   // the luma the item's positions read
    static_assert(sizeof(s_y) >= sizeof(double) * FIX_GROUP * 32, "s_tp lives in s_y");
    double (*s_tp)[32] = reinterpret_cast<double (*)[32]>(s_y);
    __shared__ float s_w3[800];
    __shared__ float s_sp[FIX_GROUP][32];   // scattered items: the local scale's per-channel terms a_c * sum_25 F_c
    __shared__ float s_amax[32];            // a_c = max_tap |W3[c][tap]|
    __shared__ unsigned s_changed, s_maxdev, s_maxratio, s_item;
    const int tid = threadIdx.x;
    if constexpr (RERUN) {
        // THE VERDICT, taken by every workgroup of this launch from the same word (fix_apply_kernel is complete: kernels of a
        // stream run in order): the largest |v_mfma - v_reference| the fix-up met on this launch's flagged pixels against
        // delta / 2.  (uniform) The margin held: nothing to do.
        if (!(__uint_as_float(*fix_word(p.counters, FIX_MAX_RATIO)) > p.rerun_above)) return;
        if (blockIdx.x == 0 && tid == 0) atomicAdd(&p.totals[FIX_N_RERUN], 1ull);
    }
    const int q = tid & (FIX_HALF - 1);
    const int hh = __builtin_amdgcn_readfirstlane(tid >> 7);
    for (int i = tid; i < 800; i += 256) s_w3[i] = p.wraw[7329 + i];
    if constexpr (!RERUN) {      // a_c from the table just staged (not 25 global loads per thread in front of the first item)
        __syncthreads();
        if (tid < 32) {
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 25; ++t) a = fmaxf(a, fabsf(s_w3[tid * 25 + t]));
            s_amax[tid] = a;
        }
    }
    if (tid == 0) { s_changed = 0; s_maxdev = 0; s_maxratio = 0; }
    const int W = p.width, H = p.height;
    const int tiles_x = (W + FIX_TILE - 1) / FIX_TILE, bands = (p.row_end - p.row_begin + FIX_TILE - 1) / FIX_TILE;
    const float b3 = p.wraw[7328];
    const cfloat_p b2 = as_constant(p.wraw) + 5248;
    // (rows beyond row_end + 5 feed no pixel of this launch: a row stripe's caller provides [row_begin - 6, row_end + 6))
    const int y_hi = min(H - 1, p.row_end + 5);

    // ---- layers 1-2 of the 128 positions of an item (this lane: position q, channel half hh) -> Fs[q][0..31] ----
    // `win`: the luma under tap (0, 0) of the lane's position in s_y, window rows PITCH apart, border already replicated.
    auto layers12 = [&](const float *win, auto pitch) {
        f32x2 acc[16];
        if constexpr (LDSW) exact_layer1_half_lds<decltype(pitch)::value>(win, s_w1t, p.wraw, hh, acc);
        else exact_layer1_half<decltype(pitch)::value>(win, p.wraw, hh, acc);
        auto own = [&](int i) { return (i & 1) ? acc[i >> 1].y : acc[i >> 1].x; };
        auto theirs = [&](int i) { return Fs[q][i]; };
        f32x2 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = f32x2{0.f, 0.f};
        // input channels 0-31: the hh = 0 lane's activations
        if (hh == 0) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { Fs[q][2 * k] = acc[k].x; Fs[q][2 * k + 1] = acc[k].y; }
        }
        __syncthreads();
        if constexpr (LDSW) {
            if (hh == 0) exact_layer2_quarter_lds(own, s_w2t, 0, 0, r);
            else exact_layer2_quarter_lds(theirs, s_w2t, 1, 0, r);
        } else {
            if (hh == 0) exact_layer2_quarter(own, p.wraw, 0, 0, r);
            else exact_layer2_quarter(theirs, p.wraw, 1, 0, r);
        }
        __syncthreads();                     // the hh = 1 lanes are done reading
        // input channels 32-63: the hh = 1 lane's
        if (hh == 1) {
#pragma unroll
            for (int k = 0; k < 16; ++k) { Fs[q][2 * k] = acc[k].x; Fs[q][2 * k + 1] = acc[k].y; }
        }
        __syncthreads();
        if constexpr (LDSW) {
            if (hh == 1) exact_layer2_quarter_lds(own, s_w2t, 1, 32, r);
            else exact_layer2_quarter_lds(theirs, s_w2t, 0, 32, r);
        } else {
            if (hh == 1) exact_layer2_quarter(own, p.wraw, 1, 32, r);
            else exact_layer2_quarter(theirs, p.wraw, 0, 32, r);
        }
        __syncthreads();                     // the hh = 0 lanes are done reading
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v0 = r[k].x + b2[16 * hh + 2 * k], v1 = r[k].y + b2[16 * hh + 2 * k + 1];
            Fs[q][16 * hh + 2 * k] = (v0 < 0) ? 0.f : v0;
            Fs[q][16 * hh + 2 * k + 1] = (v1 < 0) ? 0.f : v1;
        }
        __syncthreads();
    };

    // ---- one dense tile: output pixels [ty0, ty0 + 12) x [tx0, tx0 + 12) of `frame`, three sub-windows ----
    auto do_dense = [&](int frame, int ty0, int tx0) {
        for (int sub = 0; sub < FIX_TILE / FIX_SUB_ROWS; ++sub) {
            const int sy0 = ty0 + sub * FIX_SUB_ROWS;                    // first output row of the sub-window
            if (sy0 >= p.row_end) break;                                 // (uniform) the band's last tiles may be cut by row_end
            __syncthreads();                     // the previous item's layer 3 is done with Fs and s_y
            // The luma the positions read is staged ONCE in LDS, as floats, border replicated: the 16 x 24 window around the
            // sub-window's 8 x 16 positions, element (r, c) = luma at (clamp(sy0 - 6 + r), clamp(tx0 - 6 + c)).  Rows are clamped
            // to the rows the launch's caller provides as well (a row stripe: [row_begin - 6, row_end + 6)): rows beyond
            // row_end + 5 feed only positions below row_end + 1, whose values reach no pixel this launch stores.
            const int wy0 = sy0 - 6, wx0 = tx0 - 6;
            for (int e = tid; e < FIX_DWIN_H * FIX_DWIN_W; e += 256)
                s_y[e] = (float)fix_src_at(p, frame, clampi_e(wy0 + e / FIX_DWIN_W, 0, y_hi), clampi_e(wx0 + e % FIX_DWIN_W, 0, W - 1));
            __syncthreads();
            // the position's FEATURE coordinates are clamped to the image (the layer-3 border replicates the map, :196-210)
            const int py = clampi_e(sy0 - 2 + q / FIX_POS, 0, H - 1), px_ = clampi_e(tx0 - 2 + q % FIX_POS, 0, W - 1);
            layers12(s_y + (py - 4 - wy0) * FIX_DWIN_W + (px_ - 4 - wx0), std::integral_constant<int, FIX_DWIN_W>());
            // ---- layer 3: threads 0..47, one output pixel each ----
            const int ly = tid / FIX_TILE, lx = tid % FIX_TILE;
            const int y = sy0 + ly, x = tx0 + lx;
            if (tid < FIX_SUB_ROWS * FIX_TILE && y < p.row_end && x < W) {
                float temp = 0.f;
                for (int c = 0; c < 32; ++c) {
                    double tp = 0.0;
#pragma unroll
                    for (int m = 0; m < 5; ++m)
#pragma unroll
                        for (int n = 0; n < 5; ++n) {
                            const float pr = s_w3[(c * 5 + m) * 5 + n] * Fs[(ly + m) * FIX_POS + lx + n][c];
                            tp = tp + (double)pr;
                        }
                    temp = (float)((double)temp + tp);
                }
                temp = temp + b3;
                const uint8_t qv = (uint8_t)clampi_e((int)temp, 0, 255);
                uint8_t *d = p.dst + (long)frame * p.dst_frame_pitch + (long)(y - p.dst_row0) * p.dst_stride + x;
                if (*d != qv) { *d = qv; atomicAdd(&s_changed, 1u); }
            }
        }
    };

    // ---- `take` (<= FIX_GROUP) scattered pixels list[0 .. take): pixel codes (frame * H + y) * W + x ----
    auto do_scattered = [&](const unsigned *list, unsigned take) {
        __syncthreads();                         // the previous item's layer 3 is done with Fs and s_y
        const unsigned o = (unsigned)q / 25u, tap = (unsigned)q % 25u;
        const bool active = q < FIX_GROUP * 25 && o < take;
        // (idle lanes recompute the group's first pixel: any other coordinates could lie outside a row stripe's input)
        const unsigned oo = active ? o : 0u;
        const unsigned pix = list[oo];
        const int y = (int)((pix / (unsigned)W) % (unsigned)H), x = (int)(pix % (unsigned)W);
        const int py = clampi_e(y + (int)((active ? tap : 0u) / 5u) - 2, 0, H - 1);     // the layer-3 border replicates FEATURE coordinates (:196-210)
        const int px_ = clampi_e(x + (int)((active ? tap : 0u) % 5u) - 2, 0, W - 1);
        // the 13 x 13 luma window around pixel k at s_y[k * 169], border replicated
        for (unsigned e = tid; e < take * 169u; e += 256) {
            const unsigned k = e / 169u, m = e % 169u;
            const unsigned pq = list[k];
            const unsigned fy = pq / (unsigned)W;                       // frame * H + y
            const int yy = (int)(fy % (unsigned)H) - 6 + (int)(m / 13u), xx = (int)(pq % (unsigned)W) - 6 + (int)(m % 13u);
            s_y[e] = (float)fix_src_at(p, (int)(fy / (unsigned)H), clampi_e(yy, 0, y_hi), clampi_e(xx, 0, W - 1));
        }
        __syncthreads();
        layers12(s_y + (int)oo * 169 + (py - 4 - (y - 6)) * 13 + (px_ - 4 - (x - 6)), std::integral_constant<int, 13>());
        // ---- layer 3: threads 0..159, one (pixel, channel) each; then one thread per pixel ----
        // (the thread index behind an empty asm: left visible, the LDS addresses below are loop invariants the optimiser computes once
        // per kernel and carries through layers12() in registers that kernel does not have -- three of them went to scratch memory)
        unsigned tid3 = (unsigned)tid;
        asm volatile("" : "+v"(tid3));
        const unsigned o3 = tid3 / 32u, c = tid3 % 32u;
        if (o3 < take) {
            double tp = 0.0;
            // ... and the channel's term of the pixel's LOCAL SCALE S1 = sum_c a_c * sum_25 F_c, a_c = max_tap |W3[c][tap]|: what
            // the strip kernel's threshold for this pixel was proportional to (l3_row_is_scale(); here on the reference's map,
            // which differs from the kernel's in the last bits only -- the monitor compares orders of magnitude)
            float fsum = 0.f;
#pragma unroll
            for (int t = 0; t < 25; ++t) {
                const float fv = Fs[o3 * 25 + t][c];
                const float pr = s_w3[c * 25 + t] * fv;
                tp = tp + (double)pr;
                fsum += fv;
            }
            s_tp[o3][c] = tp;
            s_sp[o3][c] = s_amax[c] * fsum;
        }
        __syncthreads();
        if ((unsigned)tid < take) {
            float temp = 0.f;
            for (int c2 = 0; c2 < 32; ++c2) temp = (float)((double)temp + s_tp[tid][c2]);
            temp = temp + b3;
            const uint8_t qv = (uint8_t)clampi_e((int)temp, 0, 255);
            const unsigned pq = list[tid];
            const unsigned fy = pq / (unsigned)W;
            const long oin = (long)((int)(fy % (unsigned)H) - p.dst_row0) * p.dst_stride + (int)(pq % (unsigned)W);
            const long o2 = (long)(fy / (unsigned)H) * p.dst_frame_pitch + oin;
            const long of = (long)(fy / (unsigned)H) * p.flag_frame_pitch + oin;
            // how far the MFMA path's value was from the reference's: v_mfma = rint(v) + (code's distance), rint(v) = the
            // stored byte (+ 1 where v sat just below the integer).  The code resolves the distance to delta / 253, so a v
            // AT an integer can decode to the other side of it: the difference is therefore taken modulo 1 (both
            // values lie within delta << 0.5 of the same integer)
            const uint8_t was = p.dst[o2];
            const float dist = ((float)p.flag[of] - 1.f) * p.code_step - p.delta;
            const float v_mfma = (float)was + (dist < 0.f ? 1.f : 0.f) + dist;
            const float dev = v_mfma - temp, adev = fabsf(dev - rintf(dev));
            atomicMax(&s_maxdev, __float_as_uint(adev));
            // ... and how much of the pixel's OWN threshold that deviation used up: the verdict of fix_rerun_kernel
            float s1 = 0.f;
            for (int c2 = 0; c2 < 32; ++c2) s1 += s_sp[tid][c2];
            // (this unit holds no fused multiply-add, tests/test_abi.py: the threshold's last bit may differ from the strip kernel's)
            atomicMax(&s_maxratio, __float_as_uint(adev * __builtin_amdgcn_rcpf(fminf(p.delta, s1 * p.kl + p.abs_term))));      // (1 ulp: a division would expand into fused steps)
            if (was != qv) { p.dst[o2] = qv; atomicAdd(&s_changed, 1u); }
        }
    };

    // ---- the draw ----
    // The first item of a workgroup is its own index, the later ones are drawn from the shared counter (which therefore counts
    // from gridDim.x): every resident workgroup drawing at once would queue on the one word (~88 returning atomics per us).
    if constexpr (RERUN) {
        const unsigned n_tiles = (unsigned)(tiles_x * bands) * (unsigned)p.n_frames;
        for (bool first_round = true;; first_round = false) {
            __syncthreads();
            if (tid == 0) s_item = first_round ? blockIdx.x : gridDim.x + atomicAdd(fix_word(p.counters, FIX_NEXT_RERUN), 1u);
            __syncthreads();
            const unsigned t = s_item;
            if (t >= n_tiles) break;
            const unsigned trow = t / (unsigned)tiles_x;                     // frame * bands + band
            do_dense((int)(trow / (unsigned)bands), p.row_begin + (int)(trow % (unsigned)bands) * FIX_TILE, (int)(t % (unsigned)tiles_x) * FIX_TILE);
        }
        return;
    } else {
        unsigned nd[FIX_REGIONS], ns[FIX_REGIONS], n_dense = 0, n_scat = 0, n_groups = 0;
#pragma unroll
        for (int r = 0; r < FIX_REGIONS; ++r) {
            ns[r] = p.counters[FIX_REGION0 + r * FIX_WORD_STRIDE];
            nd[r] = p.counters[FIX_REGION0 + r * FIX_WORD_STRIDE + 32];
            n_dense += nd[r];
            n_scat += ns[r];
            n_groups += (ns[r] + FIX_GROUP - 1) / FIX_GROUP;
        }
        const unsigned n_items = n_dense + n_groups;
        const unsigned region_wgs = fix_region_wgs(p.width, p.row_end - p.row_begin, p.n_frames);
        const unsigned scat_cap = region_wgs * (FIX_SEG_TILES * (FIX_DENSE_MIN - 1)), dense_cap = region_wgs * FIX_SEG_TILES;
        for (bool first_round = true;; first_round = false) {
            __syncthreads();                         // s_item's readers of the previous round are done (and the kernel's LDS set-up)
            // (round 6 A/B, not kept -- profiles/r06/fix_apply_ab.txt: a first item spread over the compute units instead of the block's
            // own index, no draw when the items fit the grid, a static assignment b, b + G, ...: all within noise at every size, the
            // static form 10 % slower on a 7680x4320 plane)
            if (tid == 0) s_item = first_round ? blockIdx.x : gridDim.x + atomicAdd(fix_word(p.counters, FIX_NEXT_ITEM), 1u);
            __syncthreads();
            const unsigned item = s_item;
            if (item >= n_items) break;
            const bool dense = item < n_dense;       // uniform
            // the item's region and its index there (uniform)
            unsigned idx = dense ? item : item - n_dense, reg = 0;
#pragma unroll
            for (int r = 0; r < FIX_REGIONS - 1; ++r) {
                const unsigned here = dense ? nd[r] : (ns[r] + FIX_GROUP - 1) / FIX_GROUP;
                if (reg == (unsigned)r && idx >= here) { idx -= here; reg = r + 1; }
            }
            if (dense) {
                const unsigned t = p.dense[reg * dense_cap + idx];
                const unsigned trow = t / (unsigned)tiles_x;                 // frame * bands + band
                do_dense((int)(trow / (unsigned)bands), p.row_begin + (int)(trow % (unsigned)bands) * FIX_TILE, (int)(t % (unsigned)tiles_x) * FIX_TILE);
            } else {
                unsigned n_reg = ns[0];                  // scattered pixels of the item's region
#pragma unroll
                for (int r = 1; r < FIX_REGIONS; ++r) n_reg = reg == (unsigned)r ? ns[r] : n_reg;
                const unsigned first = idx * FIX_GROUP;
                do_scattered(p.scat + reg * scat_cap + first, min((unsigned)FIX_GROUP, n_reg - first));
            }
        }
        __syncthreads();
        if (tid == 0) {
            // (the context's totals are 64-bit words: at 23 k flagged pixels per 3840x2160 frame and 800 frames a second a 32-bit
            // count wraps after four minutes of streaming.  No workgroup waits for a reply here: round 5's first form elected the LAST
            // workgroup with a returning atomic per workgroup, 1,280 of them on one word within a few microseconds when a small plane's
            // workgroups all finish together -- 5 us on a 1920x1080 plane)
            if (s_changed) { atomicAdd(fix_word(p.counters, FIX_N_CHANGED), s_changed); atomicAdd(&p.totals[FIX_N_CHANGED], (unsigned long long)s_changed); }
            if (s_maxdev) { atomicMax(fix_word(p.counters, FIX_MAX_DEV), s_maxdev); atomicMax(&p.totals[FIX_MAX_DEV], (unsigned long long)s_maxdev); }
            if (s_maxratio) { atomicMax(fix_word(p.counters, FIX_MAX_RATIO), s_maxratio); atomicMax(&p.totals[FIX_MAX_RATIO], (unsigned long long)s_maxratio); }
            if (blockIdx.x == 0) {
                if (n_scat) atomicAdd(&p.totals[FIX_N_SCAT], (unsigned long long)n_scat);
                if (n_dense) atomicAdd(&p.totals[FIX_N_DENSE], (unsigned long long)n_dense);
            }
        }
    }
}

__global__ __launch_bounds__(256, 5) void fix_apply_kernel(const FixParams p) { fix_kernel_body<false>(p); }
// ... the weights in LDS: 54 KB and <= 168 registers, three workgroups per CU
__global__ __launch_bounds__(256, 3) void fix_apply_lds_kernel(const FixParams p) { fix_kernel_body<false, true>(p); }
__global__ __launch_bounds__(256, 5) void fix_rerun_kernel(const FixParams p) { fix_kernel_body<true>(p); }

hipError_t launch_fixup(const FixParams &p, int n_cu, bool with_rerun, bool lds_weights, hipStream_t st)
{
    const int rows = p.row_end - p.row_begin;
    const int tiles_x = (p.width + FIX_TILE - 1) / FIX_TILE, bands = (rows + FIX_TILE - 1) / FIX_TILE;
    const int segs = (tiles_x + FIX_SEG_TILES - 1) / FIX_SEG_TILES;
    hipLaunchKernelGGL(fix_collect_kernel, dim3((unsigned)(bands * segs * p.n_frames)), dim3(256), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // every workgroup the GPU holds at once (5 per CU: <= 102 VGPRs, 26 KB of LDS each; with the weights in LDS 3 per CU: 55 KB,
    // <= 168 VGPRs); they draw items from FIX_NEXT_ITEM
    if (lds_weights) hipLaunchKernelGGL(fix_apply_lds_kernel, dim3((unsigned)(3 * n_cu)), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(fix_apply_kernel, dim3((unsigned)(5 * n_cu)), dim3(256), 0, st, p);
    e = hipGetLastError();
    if (e != hipSuccess || !with_rerun) return e;
    hipLaunchKernelGGL(fix_rerun_kernel, dim3((unsigned)(5 * n_cu)), dim3(256), 0, st, p);
    return hipGetLastError();
}

size_t fixup_list_entries(int width, int rows, int n_frames, size_t *dense_entries)
{
    const size_t wgs = (size_t)fix_region_wgs(width, rows, n_frames) * FIX_REGIONS;
    if (dense_entries) *dense_entries = wgs * FIX_SEG_TILES;
    return wgs * FIX_SEG_TILES * (FIX_DENSE_MIN - 1);     // a tile with more flagged pixels than that is a dense item
}

static inline dim3 px_grid(int w, int h, int n) { return dim3((w + 63) / 64, (h + 3) / 4, n); }

hipError_t launch_conv99_exact(const uint8_t *src, long sstride, float *dst, long dstride, int w, int h,
                               const float *d_kernel81, float bias, hipStream_t st)
{
    hipLaunchKernelGGL(conv99_exact_kernel, px_grid(w, h, 1), dim3(256), 0, st, src, sstride, dst, dstride, w, h,
                       d_kernel81, bias);
    return hipGetLastError();
}

hipError_t launch_conv11_exact(const float *planes, long stride, long pitch, float *dst, long dstride,
                               int w, int h, const float *d_kernel64, float bias, hipStream_t st)
{
    hipLaunchKernelGGL(conv11_exact_kernel, px_grid(w, h, 1), dim3(256), 0, st, planes, stride, pitch, dst,
                       dstride, w, h, d_kernel64, bias);
    return hipGetLastError();
}

hipError_t launch_conv99x11_exact(const uint8_t *src, long sstride, long src_frame_pitch, float *planes,
                                  long stride, long pitch, long frame_pitch, int w, int h, int n_frames,
                                  const float *d_weights, hipStream_t st)
{
    // d_weights = convdata.h-order table (8,129 floats) followed by W2 transposed ([64][32]) and W1 transposed ([81][64])
    hipLaunchKernelGGL(conv99x11_exact_kernel, px_grid(w, h, n_frames), dim3(256), 0, st, src, sstride,
                       src_frame_pitch, planes, stride, pitch, frame_pitch, w, h, d_weights, d_weights + 8129, 0, h, 0, 0);
    return hipGetLastError();
}

hipError_t launch_conv55_exact(const float *planes, long stride, long pitch, long frame_pitch, uint8_t *dst,
                               float *pre, long dstride, long dst_frame_pitch, int w, int h, int n_frames,
                               const float *d_kernel800, float bias, hipStream_t st)
{
    hipLaunchKernelGGL(conv55_exact_kernel, px_grid(w, h, n_frames), dim3(256), 0, st, planes, stride, pitch,
                       frame_pitch, dst, pre, dstride, dst_frame_pitch, w, h, d_kernel800, bias, 0, h, 0, h, 0);
    return hipGetLastError();
}

}  // namespace srcnn
