// srcnn_mfma.hip -- the MFMA strip kernels of the SRCNN Y-channel conv path
// for gfx950 (MI355X).  Replaces the arithmetic of the reference's
// Convolution99x11 (src/srcnn.cpp:254-325) and Convolution55 (:189-243).
//
// One kernel template, three modes:
//   MODE_FUSED  u8 luma -> u8 luma; layers 1, 2, 3 chained in registers/LDS,
//               the 64- and 32-channel maps never reach HBM;
//   MODE_L12    u8 luma -> 32 planar f32 maps           (Convolution99x11);
//   MODE_L3     32 planar f32 maps -> u8 luma           (Convolution55).
//
// Work decomposition.  A workgroup of 4 waves owns a COLUMN STRIP of FW = 128
// feature columns and walks down a segment of rows.  Per feature row every
// wave owns one UNIT = 32 consecutive pixels and runs, entirely in registers,
//
//   L1  D1[64ch][32px] = W1aug[64][82] x im2col(Y)[82][32px]     82 MFMA
//       (81 taps + a constant-1 tap that carries the bias), ReLU
//   L2  D2[32ch][32px] = b2 + W2[32][64] x D1                    32 MFMA
//       (the chain starts from the bias: it is the first MFMA's C operand), ReLU
//   L3  T [25tap][32px] = W3t[25(32)][32ch] x D2                 16 MFMA
//
// Layers 1 and 2 are scaled by exact powers of two (srcnn_kernels.h) so that ReLU is the clamp bit of
// a packed multiply by 1.0, two registers per instruction; MODE_L12 stores the unscaled map.
//
// with v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate; each instruction is
// bit-for-bit a 2-term fmaf chain).  The weights are the A operand (channel /
// tap on the accumulator ROW), the pixels are the B operand (pixel on the
// LANE), so each layer's accumulator registers are directly the next layer's B
// operands: register r of lane-half h holds accumulator row (r&3)+8(r>>2)+4h,
// and the host packs the weight rows (row_chan()) so that this is channel
// 2r+h -- the k-steps then visit the channels in ascending order, which is the
// reference's summation order (src/srcnn.cpp:312-315).
//
// Layer 3 has a single output channel, so instead of a [P x 800] x [800 x 1]
// product the MFMA computes, per FEATURE pixel, the 25 tap-partials
// T[tap] = sum_c W3[c][tap] * F[c]; the output pixel is then the shifted sum
// out(y,x) = b3 + sum_{m,n} T[5m+n](y+m-2, x+n-2) with the reference's
// replicate border applied to the feature coordinates (src/srcnn.cpp:196-210).
// A wave owns the same pixel columns on every row, so the sum over the 5 tap
// ROWS is kept as register chains (12 adds per feature row); only the 5
// finished per-tap-column values of a pixel cross lanes, through a small
// double-buffered LDS tile, for the 5-term horizontal sum.
//
// Nothing is computed twice.  The 5x5 layer couples a strip to its neighbours and a work item to
// the items above and below it; instead of recomputing halo feature columns / rows, MODE_FUSED
// hands the few boundary sums over: "row seams" between the work items of a strip and "column
// seams" between strips (srcnn_kernels.h), finished by srcnn_seam_kernel / srcnn_cseam_kernel below
// with the same additions in the same order (bit-identical to the halo-recompute form).
//
// HBM traffic of MODE_FUSED is ~1.1 B read + 1 B written per pixel plus the seam scratch (25 MB per
// 3840x2160 plane, written and read once); the kernel is bound by the f32 MFMA pipe
// (130 MFMA x 64 cycles per 32 pixels per SIMD).
//
// LDS per workgroup: Y ring 2x16x136 f32 (17.0 KiB) + F tiles (18 KiB; the production kernel uses a ring of four rows
// of 6x128 f32 in it) = 35 KiB; 249 VGPRs -> two workgroups per CU, i.e. two waves per
// SIMD: while one wave waits (LDS, barrier, a dependent chain's result) the other keeps the matrix
// pipe busy.  Vector instructions are NOT hidden that way (profiles/r02/ablation.txt): every one
// costs ~6 cycles of matrix-pipe time wherever it sits, so the row loop keeps them few -- ReLU packed,
// bias as accumulator init, row addresses in scalar registers (saddr form), stores exec-masked, and a FAST row body
// (see the row loop) that is unrolled over four rows and finishes two output rows at once on the two lane halves:
// 51 vector instructions per wave-row next to the 130 MFMAs.
// MODE_L3 (HBM-bound) needs only the F tiles and runs four workgroups per CU.
#include "srcnn_kernels.h"

#include <type_traits>

// -DSRCNN_SAFE_HAZARDS=1: the FALLBACK instantiation of the strip kernels, in which every MFMA -> vector-ALU and vector-ALU -> MFMA
// operand dependency of the row body is visible to the compiler's hazard recogniser, which then pads the wait states the ISA
// manual asks for: the first MFMAs of the layer-2 / layer-3 chains are the builtin, ReLU is a plain max (no inline asm anywhere
// between them).  The fast form relies on the hardware interlocking those dependencies and is ~3 % faster.  This file is
// compiled TWICE into the library (srcnn_cpp_amd/build.py): as it is -- kernels in namespace srcnn::fast, launch_strip() and
// the seam launchers -- and with the define -- the same strip kernels in namespace srcnn::safe behind launch_strip_safe().
// srcnn_create() runs the row body's exact instruction sequences with and without wait states once per device
// (srcnn_probe.hip) and the context launches the safe kernels if they ever disagree (srcnn_kernel_variant()).  Both forms
// produce the same bytes (tests/test_gpu_hardening.py::test_safe_hazard_kernels_give_the_same_bytes).
#ifndef SRCNN_SAFE_HAZARDS
#define SRCNN_SAFE_HAZARDS 0
#endif
#if SRCNN_SAFE_HAZARDS
#define SRCNN_KNS safe
#else
#define SRCNN_KNS fast
#endif

namespace srcnn {
namespace SRCNN_KNS {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

// The first MFMA of a chain whose B operand a packed-multiply (inline asm) has just produced, written as inline asm
// too: the compiler's hazard recogniser does not count inline-asm statements as wait states, sees the previous chain's
// last MFMA "right before" this one and pads with s_nop 13 / s_nop 9 (56 / 40 idle cycles per wave-row) although
// 16 / 8 vector instructions lie between them.
__device__ __forceinline__ f32x16 mfma_first(float a, float b, const f32x16 &c)
{
#if SRCNN_SAFE_HAZARDS
    return MFMA(a, b, c);
#endif
    f32x16 d;
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f32x16 mfma_first0(float a, float b)      // ... with a zero accumulator
{
#if SRCNN_SAFE_HAZARDS
    const f32x16 z = {0};
    return MFMA(a, b, z);
#endif
    f32x16 d;
    asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "v"(b));
    return d;
}


__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// A UNIFORM pointer the optimiser cannot look through.  "scalar base + the lane's 32-bit offset" is one global instruction
// (saddr form) only while the 64-bit base stays a scalar; left visible, base + offset is re-associated into a 64-bit vector
// add (v_lshl_add_u64) per access.
// ... and the lane's 32-bit offset, kept 32 bits wide up to the access (hoisted out of the row loop as a zero-extended
// 64-bit pair it no longer matches the saddr form either).
__device__ __forceinline__ unsigned lane_off(unsigned x)
{
    asm volatile("" : "+v"(x));
    return x;
}
template <typename T>
__device__ __forceinline__ __attribute__((address_space(1))) T *scalar_base(T *p)
{
    // (a pointer to GLOBAL memory: behind the asm the optimiser can no longer infer that from the kernel argument, and a
    // generic pointer would turn the access into a flat_ instruction)
    __attribute__((address_space(1))) T *g = (__attribute__((address_space(1))) T *)p;
    asm volatile("" : "+s"(g));
    return g;
}

// A pointer every lane holds the same value of, moved to scalar registers explicitly (two v_readfirstlane_b32) where the
// compiler's divergence analysis cannot see that it is uniform.
template <typename T>
__device__ __forceinline__ T *uniform_ptr(T *p)
{
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}

// Row barrier.  The waves of a workgroup only exchange data through LDS, so the barrier has to
// order LDS traffic only: __syncthreads() would also wait for every outstanding global store and
// load (s_waitcnt vmcnt(0)) on every row, which stalls MODE_L12 behind its 16 plane stores per row.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ReLU as ONE v_max_f32 (fmaxf() adds a canonicalising v_max first).  For finite x,
// max(x, 0) equals the reference's (x < 0) ? 0 : x  (src/srcnn.cpp:304,319) up to the sign of zero.
__device__ __forceinline__ float relu(float x)
{
    float r;
    asm("v_max_f32 %0, 0, %1" : "=v"(r) : "v"(x));
    return r;
}

// Column-seam export slot e of a strip, for one output row: the sum, in this order, of up to four F-tile
// entries (plane n, strip column c).  Slots 0-4 belong to the right edge (0, 1: the own terms of the pixels at
// columns 126 and 127, i.e. their 5-term sums cut off where the right neighbour's columns begin; 2-4: single
// values the right neighbour's first two pixels start with), slots 5-14 to the left edge (5-7, 8-11: the own
// terms of the pixels at columns 0 and 1, added one by one behind the left neighbour's; 12-14: single values
// that finish the left neighbour's last two pixels).  See srcnn_cseam_kernel.
__device__ const signed char CSEAM_TERMS[15][9] = {
    {4, 0, 124, 1, 125, 2, 126, 3, 127}, {3, 0, 125, 1, 126, 2, 127, 0, 0}, {1, 0, 126, 0, 0, 0, 0, 0, 0},
    {1, 0, 127, 0, 0, 0, 0, 0, 0},       {1, 1, 127, 0, 0, 0, 0, 0, 0},     {1, 2, 0, 0, 0, 0, 0, 0, 0},
    {1, 3, 1, 0, 0, 0, 0, 0, 0},         {1, 4, 2, 0, 0, 0, 0, 0, 0},       {1, 1, 0, 0, 0, 0, 0, 0, 0},
    {1, 2, 1, 0, 0, 0, 0, 0, 0},         {1, 3, 2, 0, 0, 0, 0, 0, 0},       {1, 4, 3, 0, 0, 0, 0, 0, 0},
    {1, 3, 0, 0, 0, 0, 0, 0, 0},         {1, 4, 0, 0, 0, 0, 0, 0, 0},       {1, 4, 1, 0, 0, 0, 0, 0, 0}};

// Export slot of lane e (e < 15 exports): term count and the tile offsets n * FW + c of its up to four terms,
// fetched once per kernel; cseam_export() then costs four LDS reads, three selects/adds and one store per row.
struct CseamLane {
    int cnt, off[4];
};
__device__ __forceinline__ CseamLane cseam_lane(int e)
{
    CseamLane cl = {0, {0, 0, 0, 0}};
    if (e < 15) {
        const signed char *t = CSEAM_TERMS[e];
        cl.cnt = t[0];
#pragma unroll
        for (int i = 0; i < 4; ++i) cl.off[i] = i < t[0] ? t[1 + 2 * i] * FW + t[2 + 2 * i] : 0;
    }
    return cl;
}
// Export the values of one finished F-tile row (planes at tile[n * FW + c]); all lanes of one wave call it.
__device__ __forceinline__ void cseam_export(const float *tile, float *dst, int e, const CseamLane &cl)
{
    const float a = tile[cl.off[0]], b = tile[cl.off[1]], c = tile[cl.off[2]], d = tile[cl.off[3]];
    float v = a;
    v = cl.cnt > 1 ? v + b : v;
    v = cl.cnt > 2 ? v + c : v;
    v = cl.cnt > 3 ? v + d : v;
    if (cl.cnt > 0) dst[(unsigned)e] = v;
}

// (The packed multiply below reads accumulator registers a few cycles behind the MFMA that writes them, inside inline asm the
// compiler's hazard recogniser does not look into: the hardware interlocks that read -- tools/mfma_interlock_probe.hip,
// profiles/r02/mfma_interlock_probe.txt: a v_add_f32 issued directly behind a 64-cycle MFMA returns the finished result.)
// ReLU of TWO registers in one instruction.  The host scales layers 1 and 2 by exact powers of two so that every
// activation is <= 1 for any 8-bit input (srcnn_kernels.h); v_pk_mul_f32 by 1.0 with the clamp bit then returns
// min(max(x, 0), 1) = max(x, 0), the reference's (x < 0) ? 0 : x (src/srcnn.cpp:304,319).
__device__ __forceinline__ void relu_pairs(f32x16 &a, const f32x2 ones)
{
#if SRCNN_SAFE_HAZARDS
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = __builtin_fmaxf(a[q], 0.f);       // activations are <= 1 by the layers' scaling: max(x, 0) is the clamp
    return;
#endif
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        f32x2 pr = {a[2 * q], a[2 * q + 1]};
        asm("v_pk_mul_f32 %0, %0, %1 clamp" : "+v"(pr) : "s"(ones));
        a[2 * q] = pr.x;
        a[2 * q + 1] = pr.y;
    }
}

// Kernels whose steady-state rows run through the FAST row body (see the row loop).  Convolution55 alone (MODE_L3) is
// bound by its plane loads: the FAST body measured 0-1 % slower there (112 registers instead of 93), so it keeps the general one.
constexpr bool fast_kernel(int mode, bool pre, int diag)
{
    (void)diag;      // (the stamped build keeps the FAST body: its four stamps per wave sit outside the row loop)
    return mode == MODE_FUSED && !pre;
}

// DIAG = 2: four wall-clock stamps per wave -- entry, loop start, loop end, exit -- beside the production code
// (SRCNN_DEBUG_TUNE & 16; tools/diag_light.py).  Stamps go to p.sink, never to an output; the pixels are the production ones.
// (The timing-only ablation kernels of rounds 1-2 -- wrong pixels by construction -- and the per-row stamp build are gone
// from this file; profiles/r02/ablation.txt names the commit that still has them.)
// FIX: SRCNN_MODE_REFBYTES -- a flag byte beside every output byte (fix_code()).
// HALO3: a row stripe whose 6 halo rows either side lie in buffers of their own (StripParams::src_top / src_bot): the Y row
// address is a scalar select per row, nothing else changes -- one launch per stripe of a row-striped plane, no band launches
// and no copy of the stripe next to its halo rows.
// (The body is a function of its own so that two kernels can run it: srcnn_strip_kernel, and srcnn_strip_fold_kernel below, whose
// grid carries the seam blocks of the PREVIOUS launch behind its own work items.)
template <int MODE, bool PRE, int DIAG = 0, bool FIX = false, bool HALO3 = false>
__device__ __forceinline__ void strip_body(const StripParams &p)
{
    static_assert(!FIX || (MODE == MODE_FUSED && !PRE && DIAG == 0), "flags belong to the production fused kernel");
    static_assert(!HALO3 || (MODE == MODE_FUSED && !PRE && DIAG == 0), "halo buffers belong to the production fused kernel");
    if constexpr (FIX) {     // the launch's fix-up counters (FixParams::counters) start at zero; the fix-up kernels run behind this one
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < FIX_COUNTERS; i += 256) p.fix_counters[i] = 0u;
    }
    unsigned long long lt[4] = {0, 0, 0, 0};
    if constexpr (DIAG == 2) lt[0] = __builtin_amdgcn_s_memrealtime();
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *ylds = reinterpret_cast<float *>(smem);   // [2*YR][YP]   (MODE_FUSED, MODE_L12)
    float *fbuf = (MODE == MODE_L3) ? ylds : ylds + 2 * YR * YP;      // [2][3][6][FW]  (MODE_FUSED, MODE_L3)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;      // pixel within the unit  (MFMA column)
    const int half = lane >> 5;   // which of the 2 k-slots this lane feeds

    constexpr int HALO = (MODE == MODE_L12) ? 0 : 2;   // layer-3 radius

    const int W = p.width, H = p.height;
    // XCD-aware work mapping: workgroups with equal blockIdx % 8 share an XCD (and its L2), so hand
    // each XCD a CONTIGUOUS range of work items -- neighbouring strips / segments, which share their
    // 12 halo columns / rows of Y -- instead of every 8th one.  Bijective for any grid size; speed
    // only (the input is 1 B/pixel, so the effect on this MFMA-bound kernel is within noise).
    int bid = blockIdx.x;
    int strip, frame = 0, ys, ye, seam_up = -1, seam_dn = -1;
    if (p.items) {
        // Single-round launch: the host hands every block its own {strip, row range}.  Block ids are
        // the hardware dispatch order -- the first n_cu blocks take wave slot 0 of every CU and win the
        // age-based MFMA arbitration, so they run ~12 % faster than the block that joins them later --
        // and the host sizes the items accordingly (measured dispatch order, profiles/r01; speed only:
        // any placement computes the same plane).
        frame = bid / p.items_per_frame;
        const int *it = p.items + ITEM_INTS * (bid - frame * p.items_per_frame);
        strip = it[0];
        ys = it[1];
        ye = it[2];
        if constexpr (MODE == MODE_FUSED) {
            seam_up = it[3] >= 0 ? it[3] + frame * p.seams_per_frame : -1;
            seam_dn = it[4] >= 0 ? it[4] + frame * p.seams_per_frame : -1;
        }
        if (ys >= ye) return;      // placeholder (no rows)
    } else {
        if (!(p.tune & 8)) {
            const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        }
        strip = bid % p.n_strips;
        bid /= p.n_strips;
        const int seg = bid % p.n_segs;
        frame = bid / p.n_segs;
        ys = p.row_begin + seg * p.seg_rows;
        ye = min(ys + p.seg_rows, p.row_end);
    }

    // column seams (srcnn_kernels.h): the strip outputs all FW columns, its four edge pixels are finished elsewhere
    const bool cs = (MODE != MODE_L12) && p.cseam != nullptr;
    // export slots: lanes 0..14 of wave 0 (and, for the two-rows-at-once form of the FAST body, lanes 32..46 as well)
    const CseamLane cl = cseam_lane((cs && threadIdx.x < 64 && (threadIdx.x & 31) < 15 &&
                                     (threadIdx.x < 32 || fast_kernel(MODE, PRE, DIAG)))
                                        ? (int)(threadIdx.x & 31) : 15);
    const int halo_c = cs ? 0 : HALO;
    const int xs = strip * (FW - 2 * halo_c);   // first output column of the strip
    const int gx0 = xs - halo_c;                // image column of feature column xi = 0
    // wave 0 exports: address of this lane's slot for output row 0 of the plane (null in the other waves; taking
    // turns among the four waves measured no better)
    // (a UNIFORM base: the export then stores through a scalar base + the lane's slot, no vector address arithmetic)
    float *cs_row0 = (cs && wave == 0)
                         ? p.cseam + (((long)frame * p.strips_total + strip) * (p.row_end - p.row_begin) - p.row_begin) * CSEAM_FLOATS
                         : nullptr;
    // A seam (srcnn_kernels.h) replaces the two halo feature rows on that side: the item then computes only its
    // own rows and leaves the two output rows next to the seam to srcnn_seam_kernel.
    const bool top_open = seam_up >= 0, bot_open = seam_dn >= 0;
    const int f_lo = top_open ? ys : max(ys - HALO, 0);
    const int f_hi = bot_open ? ye : min(ye + HALO, H);   // feature rows [f_lo, f_hi) are computed
    const int out_lo = top_open ? ys + 2 : ys, out_hi = bot_open ? ye - 2 : ye;   // output rows finished here

    // ---- weight fragments -> registers (A operands, one VGPR per k-step) ----
    const float *wf = p.wfrag + lane;
    float w1f[2][41], w2f[32], w3f[16];
    f32x16 b2v = {0};
    if constexpr (MODE != MODE_L3) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int s = 0; s < 41; ++s) w1f[t][s] = wf[(t * 41 + s) * 64];
        // MODE_FUSED keeps layer 2 scaled (ReLU by clamp); MODE_L12 stores the reference's unscaled map
        constexpr int L2 = (MODE == MODE_L12) ? FRAG_L2U : FRAG_L2, B2 = (MODE == MODE_L12) ? FRAG_B2U : FRAG_B2;
#pragma unroll
        for (int q = 0; q < 32; ++q) w2f[q] = wf[(L2 + q) * 64];
#pragma unroll
        for (int q = 0; q < 16; ++q) b2v[q] = wf[(B2 + q) * 64];
    }
    if constexpr (MODE != MODE_L12) {
        constexpr int L3 = (MODE == MODE_L3) ? FRAG_L3U : FRAG_L3;
#pragma unroll
        for (int q = 0; q < 16; ++q) w3f[q] = wf[(L3 + q) * 64];
    }
    const f32x2 ones = {1.0f, 1.0f};

    // ---- layer-1 input: rolling window of Y rows as f32 in LDS --------------
    // Row r lives in slot r&15 AND in slot (r&15)+16, so the 9-row window that
    // starts at slot (f-4)&15 is always contiguous and every tap is a
    // compile-time LDS offset.  Column c of the ring is image column
    // clamp(gx0-4+c): the reference's replicate border (src/srcnn.cpp:266-280).
    const uint8_t *srcf = nullptr;
    int ycol = 0;
    if constexpr (MODE != MODE_L3) {
        srcf = p.src + (long)frame * p.src_frame_pitch;
        ycol = clampi(gx0 - 4 + tid, 0, W - 1);
    }
    // Y rows beyond f_hi + 3 feed no feature row of this block.  The row loop's prefetch stops there, so that a launch on a
    // row stripe reads nothing outside the rows its caller must provide, [row_begin - 6, row_end + 6)
    // (include/srcnn_amd.h, srcnn_forward_y_rows_dev): the ring slots behind y_last hold copies of that row, which
    // nothing consumes.
    const int y_last = min(H - 1, f_hi + 3);
    // start of image row r (uniform, already clamped to the rows the launch may read)
    auto y_row = [&](int r) -> const uint8_t * {
        if constexpr (HALO3) {
            if (r < p.src_row0) return p.src_top + (long)(r - (p.src_row0 - 6)) * p.halo_stride;
            if (r >= p.src_row1) return p.src_bot + (long)(r - p.src_row1) * p.halo_stride;
        }
        return srcf + (long)(r - p.src_row0) * p.src_stride;
    };
    auto load_y = [&](int r) -> uint8_t {
        const uint8_t *row = y_row(clampi(r, 0, y_last));         // uniform: scalar base + the lane's column
        return row[(unsigned)ycol];
    };
    auto stage_y = [&](int r, uint8_t v) {
        const int slot = r & (YR - 1);
        const float fv = (float)v;
        ylds[slot * YP + tid] = fv;
        ylds[(slot + YR) * YP + tid] = fv;
    };
    if constexpr (MODE != MODE_L3) {
        if (tid < YP) {
            uint8_t v[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) v[q] = load_y(f_lo - 4 + q);
#pragma unroll
            for (int q = 0; q < 9; ++q) stage_y(f_lo - 4 + q, v[q]);
        }
        __syncthreads();
    }
    // The weight fragments must have ARRIVED before the row loop: otherwise the compiler leaves the waits for them
    // (s_waitcnt vmcnt(35) ... vmcnt(0)) at their first uses inside the layer-1 MFMA stream, where they run again in every
    // row and make the wave wait there for its just-issued Y load and its last stores.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (MODE != MODE_L3) {
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int k = 0; k < 41; ++k) asm volatile("" : "+v"(w1f[t][k]));
#pragma unroll
        for (int q = 0; q < 32; ++q) asm volatile("" : "+v"(w2f[q]));
        asm volatile("" : "+v"(b2v));
    }
    if constexpr (MODE != MODE_L12) {
#pragma unroll
        for (int q = 0; q < 16; ++q) asm volatile("" : "+v"(w3f[q]));
    }

    if constexpr (DIAG == 2) lt[1] = __builtin_amdgcn_s_memrealtime();

    const int xi = 32 * wave + j;      // this lane's feature column in the strip
    const int gx = gx0 + xi;           // ... and in the image

    // (MODE_L3 takes its layer-3 input straight from the 32 planes: load_planes_at() below)

    // ---- layer 3, after the MFMA: vertical sums in registers, horizontal sums through LDS -----
    // out(y,x) = b3 + sum_n F_n(y, clamp(x+n-2)),   F_n(y,x') = sum_m T[5m+n](clamp(y+m-2), x').
    // The host packs W3 so that accumulator register 5s+m of lane-half 0 holds tap (m, n=s) and of
    // lane-half 1 tap (m, n=3+s)  (s = 0..2; slot 2 of half 1 is unused).  A wave owns the same 32
    // pixel columns on every row, so the sum over m (one tap row per feature row, m ascending) is a
    // register chain: R[k][s] = taps m=0..k of the output row that is k+1 rows behind completion.
    // Only the 5 finished values F_n per pixel cross lanes, through a double-buffered LDS tile;
    // that 5-term shifted sum of the previous row runs inside the layer-1 MFMA stream of this row.
    // FASTK kernels (the production fused kernel) run their steady-state rows through a FAST row body, unrolled over four
    // rows with everything that depends on f & 3 a compile-time constant: the F-tile slot (LDS offsets become immediates), one
    // in-place chain update without the image-top / image-bottom / seam-export cases, and the horizontal sums, stores and
    // column-seam exports of TWO finished rows at once every second row -- lane-half 0 takes output row f - 3, lane-half 1
    // row f - 4 (in the general body both halves compute the same row).  The rows at either end of a work item (the first six or
    // seven, which export to the seam above or see the image top, and at most one row behind the last pair, or the image's
    // bottom row) go through the general body.  Same operations in the same order on every pixel: bit-identical.
    constexpr bool FASTK = fast_kernel(MODE, PRE, DIAG);
    static_assert(!FIX || FASTK, "the flag threshold reads plane 5 through the FAST kernels' column offsets");
    constexpr int FSLOT = 6 * FW;                 // floats per F-tile slot
    int xn[5] = {0, 0, 0, 0, 0};
    bool px_ok = false;
    if constexpr (MODE != MODE_L12) {
#pragma unroll
        for (int n = 0; n < 5; ++n) {
            xn[n] = clampi(clampi(gx + n - 2, 0, W - 1) - gx0, 0, FW - 1);
            // FASTK: the offset of F_n inside a slot, lane-half 1 one slot further (its row of the pair)
            if constexpr (FASTK) xn[n] += n * FW + half * FSLOT;
        }
        px_ok = cs ? (gx < W) && (xi >= 2 || strip == 0) && (xi < FW - 2 || strip == p.strips_total - 1)
                   : (xi >= HALO) && (xi < FW - HALO) && (gx < W);
    }
    CseamLane cl2 = cl;                           // FASTK: export offsets with lane-half 1 one slot further
    if constexpr (FASTK) {
#pragma unroll
        for (int i = 0; i < 4; ++i) cl2.off[i] += half * FSLOT;
    }
    float R[4][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    // Row offsets that advance by one stride per loop iteration: kept in scalar registers and ADDED to, never
    // re-multiplied (a 64-bit row * stride product per row lands on the vector ALU), so that every global access of the
    // row loop is "scalar base + the lane's 32-bit offset".  o_out: output row f - 3 (the row hp_use() finishes in
    // iteration f); o_src: Y row min(f + 5, y_last); o_pl: plane row f (MODE_L12 store) / min(f + 1, H - 1) (MODE_L3 load).
    long o_out = (long)frame * p.dst_frame_pitch + (long)(f_lo - 3 - p.dst_row0) * p.dst_stride;
    long o_src = (long)(min(f_lo + 5, y_last) - p.src_row0) * p.src_stride;
    long o_pl = (long)frame * p.pl_frame_pitch + (long)(MODE == MODE_L3 ? min(f_lo + 1, H - 1) : f_lo) * p.pl_stride;
    // FIX: the pixel's LOCAL SCALE S1 = sum_n V(clamp(x + n - 2)) from plane 5 of the finished F-tile row (l3_row_is_scale(),
    // srcnn_kernels.h) -- the columns of the 5-term horizontal sum, (5 - n) planes further on -- and its flag threshold
    auto scale_thr = [&](const float *fr) -> float {
        float s1 = fr[xn[0] + 5 * FW];
#pragma unroll
        for (int n = 1; n < 5; ++n) s1 += fr[xn[n] + (5 - n) * FW];
        return fix_threshold(s1, p.fix_delta, p.fix_kl, p.fix_abs);
    };
    auto finalize = [&](long o, float acc, bool ok, const float *fr) {
        const float v = acc + p.b3;
        // (int) truncates toward zero, then clamp: src/srcnn.cpp:238-240.  Lanes that own no output pixel are masked off.
        auto row = scalar_base(p.dst + o);
        if (ok) row[lane_off((unsigned)gx)] = (uint8_t)clampi((int)v, 0, 255);
        if constexpr (FIX) {
            auto frow = scalar_base(p.flag + o);
            const float thr = scale_thr(fr);
            if (ok) frow[lane_off((unsigned)gx)] = fix_code(v, thr, p.fix_delta, p.fix_scale);
        }
        if constexpr (PRE) {
            float *prow = p.pre + o;
            if (ok) prow[(unsigned)gx] = v;
        }
    };
    // F tile.  FASTK: a ring of four slots, feature row g (+ slot for the two extra rows the image's last feature row
    // completes) in slot {0, 3, 2, 1}[(g + slot) & 3], so that an even row and the odd row before it are neighbours.
    // Otherwise [parity of the feature row that completed it][slot][plane n][FW]; slot 0 = output row
    // g-2, slots 1,2 = rows g-1, g, which only the image's last feature row g = H-1 completes.
    auto ftile = [&](int g, int slot) -> float * {
        if constexpr (FASTK) return fbuf + ((4 - ((g + slot) & 3)) & 3) * FSLOT;
        else return fbuf + (((g & 1) * 3 + slot) * 6) * FW;
    };
    // After the layer-3 MFMAs of feature row f: advance the chains, emit the finished F values.
    auto vertical = [&](int f, const f32x16 &t) {
        const int fplane = 3 * half * FW + xi;              // half 0 -> planes 0..2, half 1 -> planes 3..5
        if (top_open && f < ys + SEAM_ROWS) {
            // hand the tap partials the chains of the seam above still need to it (4 times per item):
            // row r = f - ys contributes its tap rows m = r+1 .. 4
            const int r = f - ys;
            float *o = p.seam + ((long)seam_up * SEAM_FLOATS + seam_t_off(r)) * NTHREADS + tid;
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int m = 1; m < 5; ++m)
                    if (m > r) o[(s * (4 - r) + (m - r - 1)) * NTHREADS] = t[5 * s + m];
        }
        if (f > 0) {
            float *fo = ftile(f, 0) + fplane;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                fo[s * FW] = R[3][s] + t[5 * s + 4];          // output row f-2: taps m=0..3 + m=4
                R[3][s] = R[2][s] + t[5 * s + 3];
                R[2][s] = R[1][s] + t[5 * s + 2];
                R[1][s] = R[0][s] + t[5 * s + 1];
                R[0][s] = t[5 * s];
            }
        } else {
            // image top: rows -1, -2 replicate row 0 (src/srcnn.cpp:203), so output rows 0 and 1
            // start with m = 0..2 resp. m = 0..1 all taken from feature row 0
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                R[0][s] = t[5 * s];
                R[1][s] = t[5 * s] + t[5 * s + 1];
                R[2][s] = R[1][s] + t[5 * s + 2];
                R[3][s] = 0.f;
            }
        }
        if (f == H - 1) {
            // image bottom: rows H, H+1 replicate row H-1, which therefore also supplies
            // m = 4 of output row H-2 and m = 3, 4 of output row H-1
            float *f1 = ftile(f, 1) + fplane, *f2 = ftile(f, 2) + fplane;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                f1[s * FW] = R[3][s] + t[5 * s + 4];
                f2[s * FW] = (R[2][s] + t[5 * s + 3]) + t[5 * s + 4];
            }
        }
    };
    // Horizontal 5-term sum of one finished output row, split in a load and a use half so it can sit
    // inside the layer-1 MFMA stream with its LDS latency hidden.  Rows outside the segment and lanes
    // without an output pixel store to the scratch word.
    // (FASTK: the offsets of lane-half 1 point one slot further, so in this one-row form only lane-half 0 counts.)
    float hv[5];
    auto hp_load = [&](int g, int slot) {
        const float *fr = ftile(g, slot);
#pragma unroll
        for (int n = 0; n < 5; ++n) hv[n] = FASTK ? fr[xn[n]] : fr[n * FW + xn[n]];
    };
    auto hp_use = [&](int g, int slot) {
        float acc = hv[0];
#pragma unroll
        for (int n = 1; n < 5; ++n) acc += hv[n];
        const int y = g - 2 + slot;        // slot > 0 only at the image's last feature row
        const bool rows_ok = (y >= out_lo) && (y < out_hi) && (!FASTK || half == 0);
        finalize(slot == 0 ? o_out : o_out + (long)slot * p.dst_stride, acc, px_ok && rows_ok, ftile(g, slot));
        if (cs_row0 && rows_ok) cseam_export(ftile(g, slot), cs_row0 + y * CSEAM_FLOATS, lane, FASTK ? cl2 : cl);
    };
    // FAST body, odd rows: output rows f - 3 (lane-half 0, F slot SLOT) and f - 4 (lane-half 1, slot SLOT + 1) together.
    // Both lie inside [out_lo, out_hi) for every row the FAST body runs on.
    const unsigned st2 = (unsigned)gx + (half ? 0u : (unsigned)p.dst_stride);      // offset from the start of row f - 4
    const unsigned cs2 = (unsigned)(lane & 31) + (half ? 0u : (unsigned)CSEAM_FLOATS);
    const int do_cs = __builtin_amdgcn_readfirstlane((cs && wave == 0) ? 1 : 0);     // a scalar, not a lane mask
    auto hp2_load = [&](int slot) {
        const float *fr = fbuf + slot * FSLOT;
#pragma unroll
        for (int n = 0; n < 5; ++n) hv[n] = fr[xn[n]];
    };
    float v2 = 0.f;                               // FIX: the pair's values between hp2_use() and hp2_flag()
    auto hp2_use = [&](int f, int slot) {
        float acc = hv[0];
#pragma unroll
        for (int n = 1; n < 5; ++n) acc += hv[n];
        const float v = acc + p.b3;
        auto row = scalar_base(p.dst + (o_out - p.dst_stride));        // uniform: output row f - 4
        if (px_ok) row[lane_off(st2)] = (uint8_t)clampi((int)v, 0, 255);
        if constexpr (FIX) {
            // the flag needs the pixel's local scale: its five plane-5 values are read HERE, into the registers the F values
            // have just left, and used six k-steps further on (hp2_flag()) -- LDS latency hidden like the F values' own
            v2 = v;
            const float *fr = fbuf + slot * FSLOT;
#pragma unroll
            for (int n = 0; n < 5; ++n) hv[n] = fr[xn[n] + (5 - n) * FW];
        }
        if (do_cs) {
            const float *tile = fbuf + slot * FSLOT;
            const float a = tile[cl2.off[0]], b = tile[cl2.off[1]], c = tile[cl2.off[2]], d = tile[cl2.off[3]];
            float e = a;
            e = cl2.cnt > 1 ? e + b : e;
            e = cl2.cnt > 2 ? e + c : e;
            e = cl2.cnt > 3 ? e + d : e;
            auto dst = scalar_base(cs_row0 + (f - 4) * CSEAM_FLOATS);     // uniform: the export row of output row f - 4
            if (cl2.cnt > 0) dst[cs2] = e;
        }
    };

    auto hp2_flag = [&]() {
        float s1 = hv[0];
#pragma unroll
        for (int n = 1; n < 5; ++n) s1 += hv[n];
        const float thr = fix_threshold(s1, p.fix_delta, p.fix_kl, p.fix_abs);
        auto frow = scalar_base(p.flag + (o_out - p.dst_stride));
        if (px_ok) frow[lane_off(st2)] = fix_code(v2, thr, p.fix_delta, p.fix_scale);
    };

    // Row loop.  Iteration f computes feature row f (layers 1-3, 130 MFMA per wave) and, INSIDE that
    // MFMA stream, finishes the output row that feature row f-1 completed; a drain step behind the loop
    // finishes the last row(s).  One barrier per row.
    // MODE_L3 is bound by the 128 B/pixel it reads: the 16 plane loads of row f+1 are issued before
    // row f is consumed, so a whole row of HBM latency hides behind the MFMAs and the row barrier.
    f32x16 d2n = {0};
    const unsigned pl_lane = (MODE == MODE_L12) ? (unsigned)(half * p.pl_pitch + gx)
                                                : (unsigned)(half * p.pl_pitch + clampi(gx, 0, W - 1));   // host: pl_pitch < 2^29
    auto load_planes_at = [&](long o) {
        const float *q = p.planes_in + o;
        const unsigned lo = lane_off(pl_lane * 4u);       // BYTE offset, 32 bits (host: pl_pitch < 2^29)
#pragma unroll
        for (int r = 0; r < 16; ++r)                      // scalar base per plane pair
            // (the planes are read once: non-temporal, they do not displace anything in L2 / the Infinity Cache;
            // together with the non-temporal stores of MODE_L12: unfused 3840x2160 1.11-1.14 -> 1.05 ms)
            d2n[r] = __builtin_nontemporal_load((const __attribute__((address_space(1))) float *)((const __attribute__((address_space(1))) char *)scalar_base(q + (long)(2 * r) * p.pl_pitch) + lo));
    };
    if constexpr (MODE == MODE_L3) load_planes_at((long)frame * p.pl_frame_pitch + (long)f_lo * p.pl_stride);
    // The drain step (the output rows the last feature row completes) sits behind the loop, not in an extra iteration
    // of it: fewer values merge at the loop head, so fewer loop-carried register copies.
    auto drain = [&](int g) {
        const int nslots = (g == H - 1) ? 3 : 1;
        for (int slot = 0; slot < nslots; ++slot) {
            hp_load(g, slot);
            hp_use(g, slot);
        }
    };
    // One row.  PH < 0: the general body; PH = 0..3: the FAST body for a row with f & 3 == PH (FASTK kernels only).
    auto row = [&](int f, auto ph_tag) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr bool FAST = PH >= 0;
        const int g = f - 1;
        const bool hp = FAST ? (PH & 1) != 0 : (MODE != MODE_L12) && (g >= f_lo);     // g < H-1 inside the loop
        constexpr int SLOT_W = (4 - PH) & 3;              // FAST: the slot feature row f writes
        constexpr int SLOT_R = (4 - ((PH + 3) & 3)) & 3;  // FAST, odd rows: the slot of feature row f - 1 (row f - 2: the next one)

        f32x16 d2;
        if constexpr (MODE != MODE_L3) {
            // prefetch the Y row the NEXT feature row needs
            // (kept as the raw byte until after the MFMA stream: converting it
            // here would make the compiler wait for the load right away)
            unsigned ynext;
            asm volatile("" : "=v"(ynext));      // (undefined in the lanes that load nothing: no instruction)
            if constexpr (HALO3) {
                const uint8_t *yr = uniform_ptr(y_row(min(f + 5, y_last)));      // a few scalar instructions per row
                if (tid < YP) ynext = scalar_base(yr)[lane_off((unsigned)ycol)];
            } else {
                if (tid < YP) ynext = scalar_base(srcf + o_src)[lane_off((unsigned)ycol)];       // Y row min(f + 5, y_last)
                if (f + 5 < y_last) o_src += p.src_stride;
            }

            // ---------------- layer 1: 82 MFMA ------------------------------
            const float *yb = ylds + ((f - 4) & (YR - 1)) * YP + xi;
            const float *ybN = yb + half;                    // taps 2s | 2s+1 in one row
            const float *ybW = yb + (half ? YP - 8 : 0);     // tap 2s = (ki,8), tap 2s+1 = (ki+1,0)
            auto ldb = [&](int s) -> float {
                const int ki = (2 * s) / 9, kj = (2 * s) % 9;
                if (s == 40) return half ? 1.0f : yb[8 * YP + 8];   // tap 80 | bias tap
                if (kj == 8) return ybW[ki * YP + kj];
                return ybN[ki * YP + kj];
            };
            f32x16 a0 = {0}, a1 = {0};
            auto layer1 = [&](auto with_pb) {
                constexpr bool PB = decltype(with_pb)::value;
                constexpr int PF = 3;                         // B operands read PF k-steps ahead
                float bq[41];
#pragma unroll
                for (int s = 0; s < PF; ++s) bq[s] = ldb(s);
#pragma unroll
                for (int s = 0; s < 41; ++s) {
                    if (s + PF < 41) bq[s + PF] = ldb(s + PF);
                    if constexpr (PB && MODE != MODE_L12) {
                        if constexpr (FAST) {
                            if (s == 2) hp2_load(SLOT_R);
                            if (s == 8) hp2_use(f, SLOT_R);
                            if constexpr (FIX) {
                                if (s == 14) hp2_flag();
                            }
                        } else {
                            if (s == 2) hp_load(g, 0);
                            if (s == 8) hp_use(g, 0);
                        }
                    }
                    a0 = MFMA(w1f[0][s], bq[s], a0);
                    a1 = MFMA(w1f[1][s], bq[s], a1);
                    // keep this k-step's LDS traffic / horizontal-sum slice where it is
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if constexpr (FAST) {
                layer1(std::integral_constant<bool, (PH & 1) != 0>{});
            } else {
                if (hp) layer1(std::true_type{});
                else layer1(std::false_type{});
            }
            // ReLU in bulk BEFORE the dependent layer-2 chain: a VALU instruction between two
            // dependent MFMAs breaks their back-to-back issue (~64 -> ~81 cycles per MFMA,
            // tools/mfma_probe.hip), 32 of them up front cost ~140 cycles once.
            relu_pairs(a0, ones);
            relu_pairs(a1, ones);
            __builtin_amdgcn_sched_barrier(0);

            // ---------------- layer 2: 32 MFMA, one back-to-back chain that starts from the bias ------------
            d2 = mfma_first(w2f[0], a0[0], b2v);
#pragma unroll
            for (int r = 1; r < 16; ++r) d2 = MFMA(w2f[r], a0[r], d2);
#pragma unroll
            for (int r = 0; r < 16; ++r) d2 = MFMA(w2f[16 + r], a1[r], d2);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE == MODE_L12) {
#pragma unroll
                for (int r = 0; r < 16; ++r) d2[r] = relu(d2[r]);
            } else {
                relu_pairs(d2, ones);
            }
            __builtin_amdgcn_sched_barrier(0);

            asm volatile("" : "+v"(ynext));
            if (tid < YP) stage_y(f + 5, (uint8_t)ynext);

            if constexpr (MODE == MODE_L12) {
                // register r / half h = channel 2r+h of pixel gx: 128-B runs per plane
                if (gx < W) {
                    float *o = p.planes_out + o_pl;      // uniform: plane row f
                    const unsigned lo = lane_off(pl_lane * 4u);       // BYTE offset, 32 bits (host: pl_pitch < 2^29)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        // (written once, 1 GB per 3840x2160 frame: non-temporal)
                        __builtin_nontemporal_store(d2[r], (__attribute__((address_space(1))) float *)((__attribute__((address_space(1))) char *)scalar_base(o + (long)(2 * r) * p.pl_pitch) + lo));
                }
            }
        } else {
            d2 = d2n;
            load_planes_at(o_pl);             // plane row min(f + 1, H - 1)
            if constexpr (FAST) {
                if constexpr ((PH & 1) != 0) {
                    hp2_load(SLOT_R);
                    hp2_use(f, SLOT_R);
                }
            } else if (hp) {
                hp_load(g, 0);
                hp_use(g, 0);
            }
        }

        if constexpr (MODE != MODE_L12) {
            // ---------------- layer 3 tap partials: 16 MFMA ------------------
            f32x16 t = {0};
            if constexpr (MODE == MODE_FUSED) {
                t = mfma_first0(w3f[0], d2[0]);
#pragma unroll
                for (int r = 1; r < 16; ++r) t = MFMA(w3f[r], d2[r], t);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) t = MFMA(w3f[r], d2[r], t);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (FAST) {
                float *fo = fbuf + SLOT_W * FSLOT + 3 * half * FW + xi;
#pragma unroll
                for (int s = 0; s < 3; ++s) {
                    fo[s * FW] = R[3][s] + t[5 * s + 4];
#ifndef SRCNN_ABL_NO_CHAIN_ADDS      // timing experiment only (wrong pixels): the upper bound of what in-place layer-3 accumulation could save
                    R[3][s] = R[2][s] + t[5 * s + 3];
                    R[2][s] = R[1][s] + t[5 * s + 2];
                    R[1][s] = R[0][s] + t[5 * s + 1];
                    R[0][s] = t[5 * s];
#endif
                    // pin the update HERE: left alone, the optimiser sinks the adds into the next row (their first use)
                    // and keeps the 16 registers of t alive across the barrier
                    asm volatile("" : "+v"(R[0][s]), "+v"(R[1][s]), "+v"(R[2][s]), "+v"(R[3][s]));
                }
            } else {
                vertical(f, t);
            }
        }

        // (nothing moves across the row barrier: the unrolled FAST rows would otherwise trade instructions -- and live registers)
        if constexpr (FASTK) __builtin_amdgcn_sched_barrier(0);
        lds_barrier();
        if constexpr (FASTK) __builtin_amdgcn_sched_barrier(0);
        o_out += p.dst_stride;
        if constexpr (MODE == MODE_L3) {
            if (f + 1 < H - 1) o_pl += p.pl_stride;
        } else {
            o_pl += p.pl_stride;
        }
    };
    {
        int f = f_lo;
        if constexpr (FASTK) {
            // FAST rows: [f_a, f_b), both EVEN (the FAST body finishes output rows in pairs), f_a >= f_lo + 6 (behind the
            // seam-export rows and the image top, and late enough for output rows f - 4 to belong to this item), f_b <= f_hi and
            // <= H - 1 (not the image's last row).  A pair in front of and a pair behind the four-row loop bring the region to
            // multiples of four (576x576, 12-row items: +1 %; 3840x2160: +0.1 %).
            const int f_a = (f_lo + 6 + 1) & ~1;
            const int f_b = min(f_hi, H - 1) & ~1;
            for (; f < f_hi && f < f_a; ++f) row(f, std::integral_constant<int, -1>{});
            if ((f & 2) && f + 2 <= f_b) {
                row(f, std::integral_constant<int, 2>{});
                row(f + 1, std::integral_constant<int, 3>{});
                f += 2;
            }
            for (; f + 4 <= f_b; f += 4) {
                row(f, std::integral_constant<int, 0>{});
                row(f + 1, std::integral_constant<int, 1>{});
                row(f + 2, std::integral_constant<int, 2>{});
                row(f + 3, std::integral_constant<int, 3>{});
            }
            if (f + 2 <= f_b) {
                row(f, std::integral_constant<int, 0>{});
                row(f + 1, std::integral_constant<int, 1>{});
                f += 2;
            }
        }
        for (; f < f_hi; ++f) row(f, std::integral_constant<int, -1>{});
    }
    if constexpr (MODE != MODE_L12) {
        if (f_hi > f_lo) drain(f_hi - 1);
    }
    if constexpr (DIAG == 2) lt[2] = __builtin_amdgcn_s_memrealtime();
    if (bot_open) {
        // hand the vertical chains (taps of this item's last rows) to the seam below
        float *o = p.seam + (long)seam_dn * SEAM_FLOATS * NTHREADS + tid;
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int s = 0; s < 3; ++s) o[(k * 3 + s) * NTHREADS] = R[k][s];
    }
    if constexpr (DIAG == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        lt[3] = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(p.sink + 256) + ((long)blockIdx.x * NWAVES + wave) * 8;
            o[0] = lt[0];
            o[1] = lt[1];
            o[2] = lt[2];
            o[3] = lt[3];
            o[4] = (unsigned long long)(f_hi - f_lo);
            o[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4) |
                   ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20) << 32);
            o[6] = (unsigned long long)strip | ((unsigned long long)ys << 32);
        }
    }
}

template <int MODE, bool PRE, int DIAG = 0, bool FIX = false, bool HALO3 = false>
__global__ __launch_bounds__(NTHREADS, MODE == MODE_L3 ? 4 : 2) void srcnn_strip_kernel(const StripParams p)
{
    strip_body<MODE, PRE, DIAG, FIX, HALO3>(p);
}

}  // namespace SRCNN_KNS
using namespace SRCNN_KNS;

#if !SRCNN_SAFE_HAZARDS       // the seam kernels hold no MFMA: one copy, in the fast translation unit

// Finishes the four output rows around every seam: replays the four chain steps the lower item's first rows
// would have contributed to the upper item's chains (same adds in the same order as vertical() / hp_use() of
// srcnn_strip_kernel), then the horizontal 5-term sum, bias, truncate, clamp (src/srcnn.cpp:235-240).
// One workgroup per seam, thread <-> (wave, lane) as in the strip kernel.
// Export slot e of a finished F-tile row (planes at tile[n * FW + c]) as cseam_export() computes it: the same terms in the same order.
__device__ __forceinline__ float cseam_value(const float *tile, int e)
{
    const signed char *t = CSEAM_TERMS[e];
    float v = tile[t[1] * FW + t[2]];
    for (int i = 1; i < t[0]; ++i) v += tile[t[1 + 2 * i] * FW + t[2 + 2 * i]];
    return v;
}

// The four pixels around a strip boundary from the left strip's exports es[0..4] and the right strip's et[5..14]
// (srcnn_cseam_kernel's arithmetic, shared with the merged form of the seam kernel).
template <bool PRE, bool FIX>
__device__ __forceinline__ void cseam_pixels(const StripParams &p, const float *es, const float *et, int frame, int y, int xt)
{
    float acc[4];
    acc[0] = es[0] + et[13];
    acc[1] = (es[1] + et[12]) + et[14];
    acc[2] = (((es[2] + es[4]) + et[5]) + et[6]) + et[7];
    acc[3] = (((es[3] + et[8]) + et[9]) + et[10]) + et[11];
    const long o = (long)frame * p.dst_frame_pitch + (long)(y - p.dst_row0) * p.dst_stride + xt - 2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (xt - 2 + k >= p.width) break;
        const float val = acc[k] + p.b3;
        p.dst[o + k] = (uint8_t)clampi((int)val, 0, 255);
        // (the four pixels around a strip boundary keep the global threshold: their local scale would need the plane-5 values of
        // both strips, which the column-seam exports do not carry -- 3 % of the pixels)
        if constexpr (FIX) p.flag[o + k] = fix_code(val, p.fix_delta, p.fix_delta, p.fix_scale);
        if constexpr (PRE) p.pre[o + k] = val;
    }
}

// MERGED: the plan keeps the seam windows of neighbouring strips apart, so this block also finishes the column-seam pixels
// either side of its strip on its four rows (the neighbours' values there are complete exports of the strip kernel) and
// exports nothing itself.
template <bool PRE, bool MERGED, bool FIX>
__device__ __forceinline__ void seam_block(const StripParams &p, const int *__restrict__ seams, int blk)
{
    __shared__ float ft[SEAM_ROWS][6][FW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 31, half = lane >> 5;
    constexpr int HALO = 2;
    const int frame = blk / p.seams_per_frame, sid = blk - frame * p.seams_per_frame;
    const int strip = seams[2 * sid], b = seams[2 * sid + 1];
    const int W = p.width;
    const bool cs = p.cseam != nullptr;
    const int halo_c = cs ? 0 : HALO;
    const int gx0 = strip * (FW - 2 * halo_c) - halo_c, xi = 32 * wave + j, gx = gx0 + xi;
    const float *sc = p.seam + (long)blk * SEAM_FLOATS * NTHREADS + tid;
    float R[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int s = 0; s < 3; ++s) R[k][s] = sc[(k * 3 + s) * NTHREADS];
#pragma unroll
    for (int r = 0; r < SEAM_ROWS; ++r) {
        // tap rows m = r+1 .. 4 of the lower item's row r (the others feed output rows below the seam's four)
        float t[3][5];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int m = r + 1; m < 5; ++m) t[s][m] = sc[(seam_t_off(r) + s * (4 - r) + (m - r - 1)) * NTHREADS];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            ft[r][3 * half + s][xi] = R[3][s] + t[s][4];
            if (r < 3) R[3][s] = R[2][s] + t[s][3];
            if (r < 2) R[2][s] = R[1][s] + t[s][2];
            if (r < 1) R[1][s] = R[0][s] + t[s][1];
        }
    }
    __syncthreads();
    if constexpr (MERGED) {
        // threads 0..7: (side, row) -- side 0 = the boundary with the left neighbour, side 1 = with the right one
        if (tid < 2 * SEAM_ROWS) {
            const int side = tid / SEAM_ROWS, r = tid % SEAM_ROWS, y = b - 2 + r;
            const int rows = p.row_end - p.row_begin;
            const int nb = side ? strip + 1 : strip - 1;
            if (nb >= 0 && nb < p.strips_total && y >= p.row_begin && y < p.row_end) {
                const float *theirs = p.cseam + (((long)frame * p.strips_total + nb) * rows + (y - p.row_begin)) * CSEAM_FLOATS;
                float mine[15];
                if (side) {
#pragma unroll
                    for (int e = 0; e < 5; ++e) mine[e] = cseam_value(&ft[r][0][0], e);
                    cseam_pixels<PRE, FIX>(p, mine, theirs, frame, y, (strip + 1) * FW);
                } else {
#pragma unroll
                    for (int e = 5; e < 15; ++e) mine[e] = cseam_value(&ft[r][0][0], e);
                    cseam_pixels<PRE, FIX>(p, theirs, mine, frame, y, strip * FW);
                }
            }
        }
    } else if (cs && wave == 0) {
        const CseamLane cl = cseam_lane(lane);
        for (int r = 0; r < SEAM_ROWS; ++r)
            cseam_export(&ft[r][0][0], p.cseam + (((long)frame * p.strips_total + strip) * (p.row_end - p.row_begin) + (b - 2 + r - p.row_begin)) * CSEAM_FLOATS, lane, cl);
    }
    const bool px_ok = cs ? (gx < W) && (xi >= 2 || strip == 0) && (xi < FW - 2 || strip == p.strips_total - 1)
                          : (xi >= HALO) && (xi < FW - HALO) && (gx < W);
    if (!px_ok) return;
    int xn[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) xn[n] = clampi(clampi(gx + n - 2, 0, W - 1) - gx0, 0, FW - 1);
#pragma unroll
    for (int r = 0; r < SEAM_ROWS; ++r) {
        float acc = ft[r][0][xn[0]];
#pragma unroll
        for (int n = 1; n < 5; ++n) acc += ft[r][n][xn[n]];
        const float v = acc + p.b3;
        const long o = (long)frame * p.dst_frame_pitch + (long)(b - 2 + r - p.dst_row0) * p.dst_stride + gx;
        p.dst[o] = (uint8_t)clampi((int)v, 0, 255);
        if constexpr (FIX) {
            // the local scale as the strip kernel has it: plane 5 = the chains of the scale rows, replayed above like the taps'
            float s1 = ft[r][5][xn[0]];
#pragma unroll
            for (int n = 1; n < 5; ++n) s1 += ft[r][5][xn[n]];
            p.flag[o] = fix_code(v, fix_threshold(s1, p.fix_delta, p.fix_kl, p.fix_abs), p.fix_delta, p.fix_scale);
        }
        if constexpr (PRE) p.pre[o] = v;
    }
}

template <bool PRE, bool FIX>
__global__ __launch_bounds__(NTHREADS) void srcnn_seam_kernel(const StripParams p, const int *__restrict__ seams)
{
    seam_block<PRE, false, FIX>(p, seams, (int)blockIdx.x);
}

// The four output pixels around every strip boundary, every row of the launch: columns xT-2, xT-1 of the left
// strip S (its pixels 126, 127) and xT, xT+1 of the right strip T (its pixels 0, 1), from the two strips' exports
// (CSEAM_TERMS), added in the order of hp_use(): F0 + F1 + F2 + F3 + F4, then the bias, truncate, clamp.
// `winmap` (merged launch only): rows inside a seam window of either strip belong to that seam's block.
template <bool PRE, bool FIX>
__device__ __forceinline__ void cseam_block(const StripParams &p, long blk_in_frame, int frame, const unsigned char *__restrict__ winmap)
{
    const int rows = p.row_end - p.row_begin;
    const long idx = blk_in_frame * 256 + threadIdx.x;
    const int v = (int)(idx / rows), yrel = (int)(idx - (long)v * rows);
    if (v >= p.strips_total - 1) return;
    if (winmap && (winmap[(long)v * rows + yrel] | winmap[(long)(v + 1) * rows + yrel])) return;
    const float *es = p.cseam + (((long)frame * p.strips_total + v) * rows + yrel) * CSEAM_FLOATS;
    const float *et = es + (long)rows * CSEAM_FLOATS;
    cseam_pixels<PRE, FIX>(p, es, et, frame, p.row_begin + yrel, (v + 1) * FW);
}

template <bool PRE, bool FIX>
__global__ __launch_bounds__(256) void srcnn_cseam_kernel(const StripParams p)
{
    cseam_block<PRE, FIX>(p, (long)blockIdx.x, (int)blockIdx.y, nullptr);
}

// Row seams and column seams in one launch: blocks [0, n_seams) finish the row seams (and the column-seam pixels of their
// rows), the others the column seams of all remaining rows; no block depends on another.
template <bool PRE, bool FIX>
__global__ __launch_bounds__(NTHREADS) void srcnn_seams_merged_kernel(const StripParams p, const int *__restrict__ seams, int n_seams,
                                                                     const unsigned char *__restrict__ winmap, int cblocks_per_frame)
{
    static_assert(NTHREADS == 256, "both roles use 256 threads");
    if ((int)blockIdx.x < n_seams) {
        seam_block<PRE, true, FIX>(p, seams, (int)blockIdx.x);
    } else {
        const int q = (int)blockIdx.x - n_seams;
        cseam_block<PRE, FIX>(p, (long)(q % cblocks_per_frame), q / cblocks_per_frame, winmap);
    }
}

size_t strip_lds_bytes(int mode);

// SEAM DEFERRAL (srcnn_set_seam_deferral).  A stream of launches queued back to back pays one seam launch per step: ~8 us of
// kernel and a launch boundary, 1 % of a 3840x2160 step, 3 % of a 1920x1080 one, 2 % of a 540-row stripe of a 7680x4320 plane.
// In this kernel the seam blocks of launch k ride BEHIND the work items of launch k + 1: they are dispatched as the first strip
// blocks finish and run on the CUs that would otherwise idle until the slowest one is done (13 us on average).  No block waits
// for another one: the seam scratch they read was completed by launch k, a finished kernel, and launch k + 1 writes the other of
// two scratch sets.  Float32 MFMA mode, single planes with explicit work items and a separated seam plan only (run_strip()).
template <bool HALO3>
__global__ __launch_bounds__(NTHREADS, 2) void srcnn_strip_fold_kernel(const StripParams p, const FoldParams f)
{
    if ((int)blockIdx.x >= f.first_block) {         // (uniform per block)
        const int q = (int)blockIdx.x - f.first_block;
        if (q >= f.n_seams) cseam_block<false, false>(f.prev, (long)(q - f.n_seams), 0, f.winmap);
        else if (f.winmap) seam_block<false, true, false>(f.prev, f.seams, q);      // row seams + the column-seam pixels of their rows
        else seam_block<false, false, false>(f.prev, f.seams, q);                   // a plan without column seams (narrow planes)
        return;
    }
    fast::strip_body<MODE_FUSED, false, 0, false, HALO3>(p);
}

hipError_t launch_strip_fold(const StripParams &p, const FoldParams &f, hipStream_t stream, size_t lds_pad)
{
    if (p.pre || p.flag || f.prev.pre || f.prev.flag || !p.items) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(f.first_block + f.n_seams + f.cblocks)), block(NTHREADS);
    const size_t lds = strip_lds_bytes(MODE_FUSED) + lds_pad;
    if (p.src_top || p.src_bot) hipLaunchKernelGGL((srcnn_strip_fold_kernel<true>), grid, block, lds, stream, p, f);
    else hipLaunchKernelGGL((srcnn_strip_fold_kernel<false>), grid, block, lds, stream, p, f);
    return hipGetLastError();
}

hipError_t launch_cseams(const StripParams &p, int n_frames, hipStream_t stream)
{
    const long n = (long)(p.strips_total - 1) * (p.row_end - p.row_begin);
    if (n <= 0) return hipSuccess;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)n_frames);
    if (p.flag) hipLaunchKernelGGL((srcnn_cseam_kernel<false, true>), grid, dim3(256), 0, stream, p);
    else if (p.pre) hipLaunchKernelGGL((srcnn_cseam_kernel<true, false>), grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((srcnn_cseam_kernel<false, false>), grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_seams_merged(const StripParams &p, int n_seams /* of all frames */, const int *d_seams, const unsigned char *d_winmap,
                               int n_frames, hipStream_t stream)
{
    const long n = (long)(p.strips_total - 1) * (p.row_end - p.row_begin);
    const int cb = (int)((n + 255) / 256);
    const dim3 grid((unsigned)(n_seams + cb * n_frames));
    if (p.flag) hipLaunchKernelGGL((srcnn_seams_merged_kernel<false, true>), grid, dim3(NTHREADS), 0, stream, p, d_seams, n_seams, d_winmap, cb);
    else if (p.pre) hipLaunchKernelGGL((srcnn_seams_merged_kernel<true, false>), grid, dim3(NTHREADS), 0, stream, p, d_seams, n_seams, d_winmap, cb);
    else hipLaunchKernelGGL((srcnn_seams_merged_kernel<false, false>), grid, dim3(NTHREADS), 0, stream, p, d_seams, n_seams, d_winmap, cb);
    return hipGetLastError();
}

hipError_t launch_seams(const StripParams &p, int n_seams /* of all frames */, const int *d_seams, hipStream_t stream)
{
    if (p.flag) hipLaunchKernelGGL((srcnn_seam_kernel<false, true>), dim3(n_seams), dim3(NTHREADS), 0, stream, p, d_seams);
    else if (p.pre) hipLaunchKernelGGL((srcnn_seam_kernel<true, false>), dim3(n_seams), dim3(NTHREADS), 0, stream, p, d_seams);
    else hipLaunchKernelGGL((srcnn_seam_kernel<false, false>), dim3(n_seams), dim3(NTHREADS), 0, stream, p, d_seams);
    return hipGetLastError();
}

size_t strip_lds_bytes(int mode)
{
    // Y ring (layers 1-2) + F tiles (layer 3); Convolution55 alone needs only the tiles
    return sizeof(float) * ((mode == MODE_L3 ? 0 : 2 * YR * YP) + 2 * 3 * 6 * FW);
}

#endif  // !SRCNN_SAFE_HAZARDS

#if SRCNN_SAFE_HAZARDS
#define launch_strip launch_strip_safe
// (LDS of a strip workgroup: the same figure as strip_lds_bytes() of the fast translation unit)
static size_t strip_lds_bytes_safe(int mode) { return sizeof(float) * ((mode == MODE_L3 ? 0 : 2 * YR * YP) + 2 * 3 * 6 * FW); }
#define strip_lds_bytes strip_lds_bytes_safe
#endif

hipError_t launch_strip(int mode, const StripParams &p, int n_frames, hipStream_t stream, size_t lds_pad)
{
    // with explicit items, n_segs carries the number of items per strip-set: n_strips * n_segs = item count
    const dim3 grid((unsigned)((long)p.n_strips * p.n_segs * n_frames));
    const dim3 block(NTHREADS);
    const size_t lds = strip_lds_bytes(mode) + lds_pad;
    const bool pre = p.pre != nullptr;
    switch (mode) {
    case MODE_FUSED:
        if (p.src_top || p.src_bot) {
            if (pre || (p.tune & 16) || n_frames != 1) return hipErrorInvalidValue;
            if (p.flag) hipLaunchKernelGGL((srcnn_strip_kernel<MODE_FUSED, false, 0, true, true>), grid, block, lds, stream, p);
            else hipLaunchKernelGGL((srcnn_strip_kernel<MODE_FUSED, false, 0, false, true>), grid, block, lds, stream, p);
        }
        else if (p.flag) {
            if (pre) return hipErrorInvalidValue;       // (the API runs pre-clamp requests of that mode on the exact kernels)
            hipLaunchKernelGGL((srcnn_strip_kernel<MODE_FUSED, false, 0, true>), grid, block, lds, stream, p);
        }
        else if (p.tune & 16) hipLaunchKernelGGL((srcnn_strip_kernel<MODE_FUSED, false, 2>), grid, block, lds, stream, p);
        else if (pre) hipLaunchKernelGGL((srcnn_strip_kernel<MODE_FUSED, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((srcnn_strip_kernel<MODE_FUSED, false>), grid, block, lds, stream, p);
        break;
    case MODE_L12:
        hipLaunchKernelGGL((srcnn_strip_kernel<MODE_L12, false>), grid, block, lds, stream, p);
        break;
    case MODE_L3:
        if (pre) hipLaunchKernelGGL((srcnn_strip_kernel<MODE_L3, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((srcnn_strip_kernel<MODE_L3, false>), grid, block, lds, stream, p);
        break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace srcnn
