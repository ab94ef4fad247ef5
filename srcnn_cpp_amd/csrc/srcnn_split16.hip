// srcnn_split16.hip -- opt-in SPLIT-F16 variant of the fused strip kernel (SRCNN_MODE_SPLIT16).
//
// a lower-precision-MFMA mode, outside the float32 north star and
// never the default.  Same path as srcnn_mfma.hip MODE_FUSED -- Convolution99x11 + Convolution55
// (src/srcnn.cpp:254-325, :189-243), u8 luma in, u8 luma out, same strip / row-segment walk, same
// layer-3 tap-partial scheme -- but the three contractions run on v_mfma_f32_32x32x16_f16
// (16x the f32 MFMA rate) with every float32 operand SPLIT into two f16 numbers so that the
// result stays at float32-level accuracy:
//
//   w = w_hi + w_lo   (host, round-to-nearest; |w - w_hi - w_lo| <= 2^-22 |w|)
//   a = a_hi + a_lo   (kernel: a_hi = max(rtz_f16(a), 0), a_lo = clamp(f16(a - a_hi), 0, 1): v_cvt_pkrtz_f16_f32,
//                      v_pk_max_f16, v_fma_mixlo/hi_f16 with the clamp bit -- ReLU comes for free: a - a_hi is
//                      exact in f32, and for a < 0 both parts come out 0)
//   w*a ~= w_hi*a_hi + w_lo*a_hi + w_hi*a_lo          (the dropped w_lo*a_lo is <= 2^-22 |w a|)
//
// f16 x f16 products are exact in the MFMA's f32 accumulation, so the only differences from the
// f32 path are the 2^-22 truncations above and the summation order.  Layer 1 needs only two
// products: its input is an 8-bit integer, exact in f16.  Power-of-two scales (exact) keep every f16
// factor in the normal range and every activation below 1024 (needed by the clamped split steps):
//   Y staged as y * 2^-14, W1 as w * 2^11, b1 as b/8 (its B operand is 1.0)  -> layer-1 map / 8
//   W2 as w * 2^14 -> accumulator = 2^11 * pre-bias; fma(acc, 2^-15, b2/16)  -> layer-2 map / 16
//   W3 as w * 2^14 -> tap partials * 2^10, removed by the final fma(acc, 2^-10, b3).
// The host checks at srcnn_set_weights that the weights allow this (rigorous bounds on the maps for
// 8-bit input: < 8192 and < 16384 here 2065 and 8786) and refuses the mode otherwise.
//
//   L1  D1[64][32px] : K = 96 slots = 81 taps + bias slot + padding, x {hi, lo} x 2 tiles   24 MFMA
//   L2  D2[32][32px] : K = 64, x {hi*hi, lo*hi, hi*lo}                                      12 MFMA
//   L3  T [25][32px] : K = 32, x 3                                                            6 MFMA
//
// = 42 MFMA x 32 cycles per 32 pixels (vs 130 x 64 in the f32 kernel).  The accumulator layout of
// 32x32x16 equals that of 32x32x2, so a layer's accumulators again feed the next layer's B operand
// in place: k-block b takes registers 8(b&1)..+7 of tile b>>1, converted pairwise to packed f16.
//
// Layer-1 B operand: the Y ring is kept in LDS as f16, TWICE -- copy 0 as is, copy 1 shifted by
// one column -- so that a lane reads two adjacent taps with one aligned ds_read_b32 whatever the
// parity of its column.  Lane-half 0 feeds window rows 0-3 and taps 0-3 of row 4, lane-half 1 rows
// 5-8 and taps 4-8 of row 4 (l1s_tap() in srcnn_kernels.h): both halves then use the same
// immediate offsets from a per-lane base.
//
// Parity: tolerance-checked against the oracle (tests/test_gpu_split16.py); there is no bitwise CPU
// model of this mode (the MFMA's internal f16 summation order is not documented).
#include "srcnn_kernels.h"

#include <type_traits>

namespace srcnn {

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)

constexpr float S16_UNSCALE_L2 = 3.0517578125e-05f;   // 2^-15
constexpr float S16_UNSCALE_L3 = 9.765625e-04f;       // 2^-10
__device__ __forceinline__ int clampi16(int v, int lo, int hi) { return min(max(v, lo), hi); }

__device__ __forceinline__ void lds_barrier16() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Schedule.  ONE workgroup per CU = one wave per SIMD with all 512 registers of a lane: the 36 weight
// fragments (144 registers) live in the accumulation-register half (they are only ever MFMA A operands),
// everything the vector ALU touches in the 256 architectural VGPRs (built with -amdgpu-mfma-vgpr-form so
// that MFMA results land there, srcnn_cpp_amd/build.py).  With a single wave on the SIMD nothing else
// hides a wave's vector work, so the row loop is software-pipelined inside the wave: while the dependent
// chain of row f runs (split -> layer 2 -> rescale/split -> layer 3), the 24 layer-1 MFMAs of row f+1,
// which depend on nothing in that chain, are issued between its steps.  Program order is the schedule:
// each of the 42 MFMAs of a row is followed by the vector work that fits in its shadow (24 issue cycles;
// two 8-cycle v_fma_mix plus two 4-cycle instructions, all independent of each other: tools/f16_probe.hip)
// and sched_barrier(0) keeps the compiler from regrouping.  Two accumulator / B-operand / tap-partial
// register sets alternate between "being converted" and "being accumulated" (loop unrolled by two), Y rows
// are staged two ahead so that the B operands of row f+2 are read before the barrier that ends row f, the
// vertical tap sums of row f are folded in at the start of row f+1 (branch-free), and the finished output
// row leaves two rows behind.  An earlier form with two workgroups per CU and no pipelining measured 5-8 %
// slower (profiles/r01/split16_ablation.txt).
constexpr int SP_RS = 264;                      // ring row pitch in halfs: every thread stages its own column
constexpr int SP_CS = 2 * YR * SP_RS + 32;
constexpr int SP_RING_BYTES = 2 * SP_CS * 2;

// A pointer every lane holds the same value of, moved to scalar registers explicitly.
template <typename T>
__device__ __forceinline__ T *uniform_ptr16(T *p)
{
    const unsigned long long a = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return (T *)(((unsigned long long)hi << 32) | lo);
}

// FIX: SRCNN_MODE_REFBYTES16 -- a flag byte beside every output byte (fix_code(), srcnn_kernels.h) for the exact fix-up kernels.
template <bool PRE, bool DIAG = false, bool FIX = false>
__global__ __launch_bounds__(NTHREADS, 1) void srcnn_split16_kernel(const StripParams p)
{
    if constexpr (FIX) {     // the launch's fix-up counters start at zero; the fix-up kernels run behind this one
        if (blockIdx.x == 0)
            for (int i = threadIdx.x; i < FIX_COUNTERS; i += 256) p.fix_counters[i] = 0u;
    }
    extern __shared__ __attribute__((aligned(16))) char smem[];
    _Float16 *ring = reinterpret_cast<_Float16 *>(smem);                  // [2 copies][2*YR][SP_RS]
    float *fbuf = reinterpret_cast<float *>(smem + SP_RING_BYTES);        // [2][3][6][FW]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31;
    const int half = lane >> 5;
    constexpr int HALO = 2;
    constexpr int OWM = FW - 2 * HALO;
    const int W = p.width, H = p.height;

    int bid = blockIdx.x;
    int strip, frame = 0, ys, ye;
    if (p.items) {
        const int *it = p.items + ITEM_INTS * bid;
        strip = it[0];
        ys = it[1];
        ye = it[2];
    } else {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        strip = bid % p.n_strips;
        bid /= p.n_strips;
        const int seg = bid % p.n_segs;
        frame = bid / p.n_segs;
        ys = p.row_begin + seg * p.seg_rows;
        ye = min(ys + p.seg_rows, p.row_end);
    }
    const int xs = strip * OWM;
    const int gx0 = xs - HALO;
    const int f_lo = max(ys - HALO, 0);
    const int f_hi = min(ye + HALO, H);

    const u32x4 *wf = reinterpret_cast<const u32x4 *>(p.wfrag16) + lane;
    u32x4 w1[2][2][6], w2[2][4], w3[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int b = 0; b < 6; ++b) w1[t][s][b] = wf[((t * 2 + s) * 6 + b) * 64];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int b = 0; b < 4; ++b) w2[s][b] = wf[(S16_FRAG_L2 + s * 4 + b) * 64];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int b = 0; b < 2; ++b) w3[s][b] = wf[(S16_FRAG_L3 + s * 2 + b) * 64];
    float b2v[16];
    {
        const float *bt = reinterpret_cast<const float *>(p.wfrag16 + S16_NFRAG * 64 * 4) + half * 16;
#pragma unroll
        for (int r = 0; r < 16; ++r) b2v[r] = bt[r];
    }
    // The weight fragments are only ever MFMA A operands: keep them in the accumulation-register half of
    // the wave's 512 registers, so the 256 architectural VGPRs are free for everything the vector ALU touches.
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int b = 0; b < 6; ++b) asm volatile("" : "+a"(w1[t][s][b]));
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int b = 0; b < 4; ++b) asm volatile("" : "+a"(w2[s][b]));
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int b = 0; b < 2; ++b) asm volatile("" : "+a"(w3[s][b]));

    // Y ring: as in srcnn_split16_kernel, but every thread stages its own column (columns >= FW+10
    // are never read) so staging needs no predicate, and rows are staged two ahead.
    const uint8_t *srcf = p.src + (long)frame * p.src_frame_pitch;
    const int ycol = clampi16(gx0 - 4 + tid, 0, W - 1);
    // (rows beyond f_hi + 3 feed no feature row of this block: the read stays inside the rows the caller must provide,
    // [row_begin - 6, row_end + 6) for a row stripe -- include/srcnn_amd.h, srcnn_forward_y_rows_dev)
    const int y_last = min(H - 1, f_hi + 3);
    auto load_y = [&](int r) -> uint8_t {
        const int rr = clampi16(r, 0, y_last) - p.src_row0;
        return srcf[(long)rr * p.src_stride + ycol];
    };
    const int c1col = tid == 0 ? SP_RS - 1 : tid - 1;                     // copy 1 holds column c at element c-1
    auto stage_y = [&](int r, uint8_t v) {
        const int slot = r & (YR - 1);
        const _Float16 hv = (_Float16)((float)v * 6.103515625e-05f);
        _Float16 *c0 = ring + slot * SP_RS + tid;
        c0[0] = hv;
        c0[YR * SP_RS] = hv;
        _Float16 *c1 = ring + SP_CS + slot * SP_RS + c1col;
        c1[0] = hv;
        c1[YR * SP_RS] = hv;
    };
    {
        uint8_t v[11];
#pragma unroll
        for (int q = 0; q < 11; ++q) v[q] = load_y(f_lo - 4 + q);
#pragma unroll
        for (int q = 0; q < 11; ++q) stage_y(f_lo - 4 + q, v[q]);
    }
    __syncthreads();

    const int xi = 32 * wave + j;
    const int gx = gx0 + xi;
    int xn[5];
#pragma unroll
    for (int n = 0; n < 5; ++n) xn[n] = clampi16(clampi16(gx + n - 2, 0, W - 1) - gx0, 0, FW - 1);
    const bool px_ok = (xi >= HALO) && (xi < FW - HALO) && (gx < W);
    float R[4][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
    float hs[5] = {0.f, 0.f, 0.f, 0.f, 0.f};      // FIX: plane 5 of the finished F-tile row at the pixel's five columns (the local scale)
    auto finalize = [&](int y, float acc, bool ok) {
        const float v = __builtin_fmaf(acc, S16_UNSCALE_L3, p.b3);
        const long o = (long)frame * p.dst_frame_pitch + (long)(y - p.dst_row0) * p.dst_stride + gx;
        uint8_t *d8 = ok ? p.dst + o : reinterpret_cast<uint8_t *>(p.sink) + lane;
        *d8 = (uint8_t)clampi16((int)v, 0, 255);                          // src/srcnn.cpp:238-240
        if constexpr (FIX) {
            // scalar row base + the lane's column (one saddr-form store): a second selected 64-bit address per lane is what made
            // this instantiation spill 48 registers to scratch and run at half the speed of the plain one
            const long orow = (long)frame * p.dst_frame_pitch + (long)(y - p.dst_row0) * p.dst_stride;      // uniform
            __attribute__((address_space(1))) uint8_t *frow = (__attribute__((address_space(1))) uint8_t *)uniform_ptr16(p.flag + orow);
            asm volatile("" : "+s"(frow));
            unsigned col = (unsigned)gx;
            asm volatile("" : "+v"(col));
            // the pixel's own threshold (srcnn_kernels.h, l3_row_is_scale()): the scale rows of the layer-3 fragments come out of the
            // same MFMAs as the taps, x 2^10 like them (fix_kl carries the 2^-10: run_strip())
            const float s1 = (((hs[0] + hs[1]) + hs[2]) + hs[3]) + hs[4];
            if (ok) frow[col] = fix_code(v, fix_threshold(s1, p.fix_delta, p.fix_kl, p.fix_abs), p.fix_delta, p.fix_scale);
        }
        if constexpr (PRE) {
            float *dp = ok ? p.pre + o : p.sink + 64 + lane;
            *dp = v;
        }
    };
    auto ftile = [&](int g, int slot) -> float * { return fbuf + (((g & 1) * 3 + slot) * 6) * FW; };
    // Branch-free: the row loop carries no control flow around the chains.  The image's FIRST feature row
    // (rows -1, -2 replicate row 0, src/srcnn.cpp:203: output rows 0 and 1 start with taps m = 0..2 resp.
    // m = 0..1 all taken from row 0) is folded in with a 0/1 factor: top * x is exact, so
    // fma(top, t0, t1) == t0 + t1 and fma(-top, t3, t3) == 0 bit for bit.
    auto vertical = [&](int f, const f32x16 &t) {
        float *fo = ftile(f, 0) + 3 * half * FW + xi;
        const float top = f == 0 ? 1.f : 0.f;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            fo[s * FW] = R[3][s] + t[5 * s + 4];            // output row f-2: taps m=0..3 + m=4
            const float t01 = t[5 * s] + t[5 * s + 1];
            R[3][s] = __builtin_fmaf(-top, t[5 * s + 3], R[2][s] + t[5 * s + 3]);
            R[2][s] = __builtin_fmaf(top, t01, R[1][s] + t[5 * s + 2]);
            R[1][s] = __builtin_fmaf(top, t[5 * s], R[0][s] + t[5 * s + 1]);
            R[0][s] = t[5 * s];
        }
    };
    // The image's LAST feature row (rows H, H+1 replicate row H-1) also supplies m = 4 of output row H-2
    // and m = 3, 4 of output row H-1; it is always the last row of its work item, i.e. part of the drain.
    auto vertical_bottom = [&](int f, const f32x16 &t) {
        float *f1 = ftile(f, 1) + 3 * half * FW + xi, *f2 = ftile(f, 2) + 3 * half * FW + xi;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            f1[s * FW] = R[3][s] + t[5 * s + 4];
            f2[s * FW] = (R[2][s] + t[5 * s + 3]) + t[5 * s + 4];
        }
    };
    float hv[5];
    auto hp_load = [&](int g, int slot) {
        const float *fr = ftile(g, slot);
#pragma unroll
        for (int n = 0; n < 5; ++n) hv[n] = fr[n * FW + xn[n]];
        if constexpr (FIX) {
#pragma unroll
            for (int n = 0; n < 5; ++n) hs[n] = fr[5 * FW + xn[n]];
        }
    };
    auto hp_use = [&](int g, int slot, bool on) {
        float acc = hv[0];
#pragma unroll
        for (int n = 1; n < 5; ++n) acc += hv[n];
        const int y = g - 2 + slot;
        finalize(y, acc, on && px_ok && (y >= ys) && (y < ye));
    };

    const int ring_lane = (xi & 1) * SP_CS + (xi & ~1);
    // B operand of feature row f: 23 dwords (l1s_tap()); part 0..3 = window rows rr of the lane-half, part 4 = row 4
    auto read_b_part = [&](int f, unsigned (&bq)[24], int part) {
        const unsigned *yb = reinterpret_cast<const unsigned *>(ring + ring_lane + ((f - 4) & (YR - 1)) * SP_RS);
        if (part < 4) {
            const unsigned *ybH = yb + half * (5 * SP_RS / 2) + part * (SP_RS / 2);
#pragma unroll
            for (int q = 0; q < 5; ++q) bq[part * 5 + q] = ybH[q];
        } else {
            const unsigned *ybX = yb + 4 * (SP_RS / 2) + 2 * half;
#pragma unroll
            for (int q = 0; q < 3; ++q) bq[20 + q] = ybX[q];
            bq[23] = 0x00003C00u;
        }
    };
    auto read_b = [&](int f, unsigned (&bq)[24]) {
#pragma unroll
        for (int part = 0; part < 5; ++part) read_b_part(f, bq, part);
    };
#define BV(b) ((u32x4){bq[4 * (b)], bq[4 * (b) + 1], bq[4 * (b) + 2], bq[4 * (b) + 3]})
#define PIN() __builtin_amdgcn_sched_barrier(0)
    // ReLU + split steps on TWO pairs (four values x[0..3] -> h[0..1], l[0..1]).  Costs inside the shadow of a
    // 32-cycle f16 MFMA (tools/f16_probe.hip): v_cvt_pkrtz / v_pk_max / v_fma_f32 4 cycles, v_fma_mixlo/hi_f16 8;
    // the MFMA itself holds the issue port 8 cycles, so a gap hides 24 cycles of INDEPENDENT vector work: two mix
    // plus two simple instructions, or four simple ones.  A dependent instruction right behind its producer
    // stalls, so every step works on values produced at least one gap earlier.  `dep` is the accumulator of the
    // MFMA issued just before the step: naming it as an (unused) input ties the step behind that MFMA in
    // program order -- the compiler otherwise sinks an MFMA whose result is not needed yet below the
    // following steps, leaving one gap overfull and the next one empty.
    auto cvt2 = [&](const float *x, unsigned *h, const f32x16 &dep) {
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2 ; after %3" : "=v"(h[0]) : "v"(x[0]), "v"(x[1]), "v"(dep));
        asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(h[1]) : "v"(x[2]), "v"(x[3]));
    };
    auto max2 = [&](unsigned *h, const f32x16 &dep) {
        asm volatile("v_pk_max_f16 %0, %0, 0 ; after %1" : "+v"(h[0]) : "v"(dep));
        asm volatile("v_pk_max_f16 %0, %0, 0" : "+v"(h[1]));
    };
    auto lo2 = [&](const float *x, const unsigned *h, unsigned *l) {
        asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(l[0]) : "v"(h[0]), "v"(x[0]));
        asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0] clamp" : "=v"(l[1]) : "v"(h[1]), "v"(x[2]));
    };
    auto hi2 = [&](const float *x, const unsigned *h, unsigned *l) {
        asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "+v"(l[0]) : "v"(h[0]), "v"(x[1]));
        asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp" : "+v"(l[1]) : "v"(h[1]), "v"(x[3]));
    };
    auto after = [&](const f32x16 &dep) { asm volatile("; after %0" ::"v"(dep)); };

    // prologue: layer 1 of the first row, not overlapped
    f32x16 accA0 = {0}, accA1 = {0}, accB0, accB1;
    unsigned bqA[24], bqB[24];
    read_b(f_lo, bqA);
    {
        unsigned (&bq)[24] = bqA;
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            accA0 = MFMA16(w1[0][0][b], BV(b), accA0);
            accA1 = MFMA16(w1[1][0][b], BV(b), accA1);
            accA0 = MFMA16(w1[0][1][b], BV(b), accA0);
            accA1 = MFMA16(w1[1][1][b], BV(b), accA1);
        }
    }
    PIN();
    read_b(f_lo + 1, bqB);

    unsigned long long dgp[6] = {0, 0, 0, 0, 0, 0};
    auto stampp = [&]() -> unsigned long long {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
        return t;
    };
    // One row: consume (c0, c1) = layer-1 result of row f, produce (n0, n1) = layer 1 of row f+1 from the
    // B operands bq (read during the previous row); read the B operands of row f+2 into bqn.
    // The layer-3 tap partials t of row f are folded into the vertical chains at the START of row f+1
    // (tp = partials of row f-1), so their F-tile writes are long complete at the barrier that ends the
    // row; the horizontal sum / output therefore lags two rows (g = f-2).
    auto row = [&](int f, f32x16 &c0, f32x16 &c1, f32x16 &n0, f32x16 &n1, unsigned (&bq)[24], unsigned (&bqn)[24],
                   const f32x16 &tp, f32x16 &t) {
        const int g = f - 2;
        const bool hp = g >= f_lo;
        unsigned ynext = load_y(f + 7);
        unsigned h1[16], l1[16], h2[8], l2[8];
        float x1[32], e[16];
        f32x16 d2 = {0};
        t = (f32x16){0};
        n0 = (f32x16){0};
        n1 = (f32x16){0};
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            x1[i] = c0[i];
            x1[16 + i] = c1[i];
        }
        // rescale + bias of half a quad of the layer-2 result (4 fma)
        auto s2e = [&](int i0) {
#pragma unroll
            for (int i = i0; i < i0 + 4; ++i) e[i] = __builtin_fmaf(d2[i], S16_UNSCALE_L2, b2v[i]);
        };
        auto bh1 = [&](int b) -> u32x4 { return (u32x4){h1[4 * b], h1[4 * b + 1], h1[4 * b + 2], h1[4 * b + 3]}; };
        auto bl1 = [&](int b) -> u32x4 { return (u32x4){l1[4 * b], l1[4 * b + 1], l1[4 * b + 2], l1[4 * b + 3]}; };
        auto bh2 = [&](int b) -> u32x4 { return (u32x4){h2[4 * b], h2[4 * b + 1], h2[4 * b + 2], h2[4 * b + 3]}; };
        auto bl2 = [&](int b) -> u32x4 { return (u32x4){l2[4 * b], l2[4 * b + 1], l2[4 * b + 2], l2[4 * b + 3]}; };
        unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0;
        if constexpr (DIAG) q0 = stampp();
        PIN();
        // Values are converted in QUADS (4 pairs = the 8 values of one k-block operand).  In every gap the two
        // simple steps of quad q (cvt or max of two pairs) share the gap with two mix instructions of quad q-1:
        //   SIMPLE(q, j):  j = 0,1: cvt of pairs 0-1 / 2-3;  j = 2,3: max of pairs 0-1 / 2-3
        //   MIX(q, j):     j = 0,1: mixlo of pairs 0-1 / 2-3;  j = 2,3: mixhi of pairs 0-1 / 2-3
        // so the hi part of quad q is complete 4 gaps after its first step, the lo part 8 gaps after.
#define SIMPLE1(q, j, dep)                                                                         \
    do {                                                                                           \
        if ((j) < 2) cvt2(x1 + 8 * (q) + 4 * (j), h1 + 4 * (q) + 2 * (j), dep);                    \
        else max2(h1 + 4 * (q) + 2 * ((j) - 2), dep);                                              \
    } while (0)
#define MIX1(q, j)                                                                                 \
    do {                                                                                           \
        if ((j) < 2) lo2(x1 + 8 * (q) + 4 * (j), h1 + 4 * (q) + 2 * (j), l1 + 4 * (q) + 2 * (j));  \
        else hi2(x1 + 8 * (q) + 4 * ((j) - 2), h1 + 4 * (q) + 2 * ((j) - 2), l1 + 4 * (q) + 2 * ((j) - 2)); \
    } while (0)
#define SIMPLE2(q, j, dep)                                                                         \
    do {                                                                                           \
        if ((j) < 2) cvt2(e + 8 * (q) + 4 * (j), h2 + 4 * (q) + 2 * (j), dep);                     \
        else max2(h2 + 4 * (q) + 2 * ((j) - 2), dep);                                              \
    } while (0)
#define MIX2(q, j)                                                                                 \
    do {                                                                                           \
        if ((j) < 2) lo2(e + 8 * (q) + 4 * (j), h2 + 4 * (q) + 2 * (j), l2 + 4 * (q) + 2 * (j));   \
        else hi2(e + 8 * (q) + 4 * ((j) - 2), h2 + 4 * (q) + 2 * ((j) - 2), l2 + 4 * (q) + 2 * ((j) - 2)); \
    } while (0)
        // g1-g4: layer 1 of the next row | simple steps of quad 0; F tile reads of the previous row
        n0 = MFMA16(w1[0][0][0], BV(0), n0); SIMPLE1(0, 0, n0); hp_load(g, 0); PIN();
        vertical(f - 1, tp);        // tp == 0 before the first row of the item: adds nothing
        PIN();
        n1 = MFMA16(w1[1][0][0], BV(0), n1); SIMPLE1(0, 1, n1); PIN();
        n0 = MFMA16(w1[0][1][0], BV(0), n0); SIMPLE1(0, 2, n0); PIN();
        n1 = MFMA16(w1[1][1][0], BV(0), n1); SIMPLE1(0, 3, n1); PIN();
        // g5-g8: layer 2 k-block 0 (hi operand) | simple quad 1, mix quad 0
        d2 = MFMA16(w2[0][0], bh1(0), d2); SIMPLE1(1, 0, d2); MIX1(0, 0); PIN();
        d2 = MFMA16(w2[1][0], bh1(0), d2); SIMPLE1(1, 1, d2); MIX1(0, 1); PIN();
        n0 = MFMA16(w1[0][0][1], BV(1), n0); SIMPLE1(1, 2, n0); MIX1(0, 2); PIN();
        n1 = MFMA16(w1[1][0][1], BV(1), n1); SIMPLE1(1, 3, n1); MIX1(0, 3); PIN();
        // g9-g12 | simple quad 2, mix quad 1
        d2 = MFMA16(w2[0][0], bl1(0), d2); SIMPLE1(2, 0, d2); MIX1(1, 0); PIN();
        d2 = MFMA16(w2[0][1], bh1(1), d2); SIMPLE1(2, 1, d2); MIX1(1, 1); PIN();
        d2 = MFMA16(w2[1][1], bh1(1), d2); SIMPLE1(2, 2, d2); MIX1(1, 2); PIN();
        n0 = MFMA16(w1[0][1][1], BV(1), n0); SIMPLE1(2, 3, n0); MIX1(1, 3); PIN();
        // g13-g16 | simple quad 3, mix quad 2
        d2 = MFMA16(w2[0][1], bl1(1), d2); SIMPLE1(3, 0, d2); MIX1(2, 0); PIN();
        d2 = MFMA16(w2[0][2], bh1(2), d2); SIMPLE1(3, 1, d2); MIX1(2, 1); PIN();
        d2 = MFMA16(w2[1][2], bh1(2), d2); SIMPLE1(3, 2, d2); MIX1(2, 2); PIN();
        n1 = MFMA16(w1[1][1][1], BV(1), n1); SIMPLE1(3, 3, n1); MIX1(2, 3); PIN();
        if constexpr (DIAG) { q1 = stampp(); PIN(); }
        // g17-g21 | mix quad 3; B operands of row f+2 (rows up to f+6 are staged and published)
        d2 = MFMA16(w2[0][2], bl1(2), d2); after(d2); MIX1(3, 0); PIN();
        d2 = MFMA16(w2[0][3], bh1(3), d2); after(d2); MIX1(3, 1); PIN();
        d2 = MFMA16(w2[1][3], bh1(3), d2); after(d2); MIX1(3, 2); PIN();
        n0 = MFMA16(w1[0][0][2], BV(2), n0); after(n0); MIX1(3, 3); PIN();
        d2 = MFMA16(w2[0][3], bl1(3), d2); after(d2); PIN();
        // g22-g23: layer 1 while the layer-2 result completes | Y staging, output of the previous row
        n1 = MFMA16(w1[1][0][2], BV(2), n1); after(n1); PIN();
        n0 = MFMA16(w1[0][1][2], BV(2), n0); after(n0); PIN();
        if constexpr (DIAG) { q2 = stampp(); PIN(); }
        // g24-g29: layer 1 | rescale + bias of the layer-2 result, simple steps of its quad 0
        n1 = MFMA16(w1[1][1][2], BV(2), n1); after(n1); s2e(0); PIN();
        n0 = MFMA16(w1[0][0][3], BV(3), n0); after(n0); s2e(4); PIN();
        n1 = MFMA16(w1[1][0][3], BV(3), n1); SIMPLE2(0, 0, n1); s2e(8); PIN();
        n0 = MFMA16(w1[0][1][3], BV(3), n0); SIMPLE2(0, 1, n0); s2e(12); PIN();
        n1 = MFMA16(w1[1][1][3], BV(3), n1); SIMPLE2(0, 2, n1); PIN();
        n0 = MFMA16(w1[0][0][4], BV(4), n0); SIMPLE2(0, 3, n0); PIN();
        // g30-g33: layer 3 k-block 0 (hi operand) | simple quad 1, mix quad 0
        t = MFMA16(w3[0][0], bh2(0), t); SIMPLE2(1, 0, t); MIX2(0, 0); PIN();
        t = MFMA16(w3[1][0], bh2(0), t); SIMPLE2(1, 1, t); MIX2(0, 1); PIN();
        n1 = MFMA16(w1[1][0][4], BV(4), n1); SIMPLE2(1, 2, n1); MIX2(0, 2); PIN();
        n0 = MFMA16(w1[0][1][4], BV(4), n0); SIMPLE2(1, 3, n0); MIX2(0, 3); PIN();
        // g34-g38 | mix quad 1
        t = MFMA16(w3[0][0], bl2(0), t); after(t); MIX2(1, 0); read_b_part(f + 2, bqn, 0); PIN();
        t = MFMA16(w3[0][1], bh2(1), t); after(t); MIX2(1, 1); read_b_part(f + 2, bqn, 1); PIN();
        t = MFMA16(w3[1][1], bh2(1), t); after(t); MIX2(1, 2); read_b_part(f + 2, bqn, 2); PIN();
        n1 = MFMA16(w1[1][1][4], BV(4), n1); after(n1); MIX2(1, 3); read_b_part(f + 2, bqn, 3); PIN();
        t = MFMA16(w3[0][1], bl2(1), t); after(t); read_b_part(f + 2, bqn, 4); PIN();
        if constexpr (DIAG) { q3 = stampp(); PIN(); }
        // g39-g42: the last layer-1 MFMAs; t is complete when they are through
        n0 = MFMA16(w1[0][0][5], BV(5), n0); after(n0); stage_y(f + 7, (uint8_t)ynext); PIN();
        n1 = MFMA16(w1[1][0][5], BV(5), n1); after(n1); hp_use(g, 0, hp); PIN();
        n0 = MFMA16(w1[0][1][5], BV(5), n0); PIN();
        n1 = MFMA16(w1[1][1][5], BV(5), n1); PIN();
#undef SIMPLE1
#undef MIX1
#undef SIMPLE2
#undef MIX2
        if constexpr (DIAG) { q4 = stampp(); PIN(); }
        lds_barrier16();
        if constexpr (DIAG) {
            q5 = stampp();
            dgp[0] += q1 - q0; dgp[1] += q2 - q1; dgp[2] += q3 - q2; dgp[3] += q4 - q3; dgp[4] += q5 - q4;
        }
    };

    unsigned long long dg_t0 = 0, dg_r0 = 0;
    if constexpr (DIAG) {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dg_t0)::"memory");
        dg_r0 = __builtin_amdgcn_s_memrealtime();
    }
    f32x16 tA = {0}, tB = {0};
    int f = f_lo;
    bool last_in_a = false;            // which of tA / tB holds the partials of the last row
    for (; f + 1 < f_hi; f += 2) {
        row(f, accA0, accA1, accB0, accB1, bqB, bqA, tB, tA);
        row(f + 1, accB0, accB1, accA0, accA1, bqA, bqB, tA, tB);
    }
    if (f < f_hi) {
        row(f, accA0, accA1, accB0, accB1, bqB, bqA, tB, tA);
        last_in_a = true;
        ++f;
    }
    // drain (f == f_hi): the output row that feature row f-2 completed, then the last feature row itself
    if (f - 2 >= f_lo) {
        hp_load(f - 2, 0);
        hp_use(f - 2, 0, true);
    }
    if (f - 1 >= f_lo) {
        // (a select of VALUES: selecting between references to tA and tB made the FIX instantiation keep both in scratch memory,
        // with 44 scratch accesses inside the row loop -- 503 us per 3840x2160 plane instead of 272)
        if constexpr (FIX) {
            f32x16 tl;
#pragma unroll
            for (int q = 0; q < 16; ++q) tl[q] = last_in_a ? tA[q] : tB[q];
            vertical(f - 1, tl);
            if (f - 1 == H - 1) vertical_bottom(f - 1, tl);
        } else {                           // (the other instantiations: their listings stay what rounds 1-4 measured)
            vertical(f - 1, last_in_a ? tA : tB);
            if (f - 1 == H - 1) vertical_bottom(f - 1, last_in_a ? tA : tB);
        }
        __syncthreads();
        const int g = f - 1;
        const int nslots = (g == H - 1) ? 3 : 1;
        for (int slot = 0; slot < nslots; ++slot) {
            hp_load(g, slot);
            hp_use(g, slot, true);
        }
    }
    if constexpr (DIAG) {
        unsigned long long t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
        if (lane == 0) {
            unsigned long long *o = reinterpret_cast<unsigned long long *>(p.sink + 256) + ((long)blockIdx.x * NWAVES + wave) * 8;
            o[0] = t1 - dg_t0;
            o[1] = r1 - dg_r0;
            o[2] = dgp[0]; o[3] = dgp[1]; o[4] = dgp[2]; o[5] = dgp[3];
            o[6] = (unsigned long long)(f_hi - f_lo) | ((dg_r0 & 0xffffffffull) << 32);
            o[7] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | (dgp[4] << 32);
        }
    }
#undef BV
#undef PIN
}

}  // namespace

size_t split16_lds_bytes() { return (size_t)SP_RING_BYTES + sizeof(float) * 2 * 3 * 6 * FW; }

hipError_t launch_split16(const StripParams &p, int n_frames, hipStream_t stream, size_t lds_pad)
{
    const dim3 grid((unsigned)((long)p.n_strips * p.n_segs * n_frames));
    const dim3 block(NTHREADS);
    const size_t lds = split16_lds_bytes() + lds_pad;
    if (p.flag) {
        if (p.pre) return hipErrorInvalidValue;     // (the API runs pre-clamp requests of that mode on the exact kernels)
        hipLaunchKernelGGL((srcnn_split16_kernel<false, false, true>), grid, block, lds, stream, p);
    }
    else if (p.tune & 2) hipLaunchKernelGGL((srcnn_split16_kernel<false, true>), grid, block, lds, stream, p);
    else if (p.pre) hipLaunchKernelGGL((srcnn_split16_kernel<true>), grid, block, lds, stream, p);
    else hipLaunchKernelGGL((srcnn_split16_kernel<false>), grid, block, lds, stream, p);
    return hipGetLastError();
}

}  // namespace srcnn
