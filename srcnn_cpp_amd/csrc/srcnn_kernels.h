// Internal interface between the C-ABI layer (srcnn_api.cpp and its sibling units, see srcnn_ctx.h) and the HIP
// kernels.  Not installed; the public boundary is include/srcnn_amd.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace srcnn {

// ---- geometry of the strip kernels (srcnn_mfma.hip) ------------------------
constexpr int FW = 128;            // feature columns per strip (4 waves x 32 lanes-columns)
constexpr int YP = FW + 8;         // Y ring pitch: strip + 4-pixel layer-1 border each side
constexpr int YR = 16;             // Y ring rows (each stored twice, see kernel)
constexpr int NWAVES = FW / 32;    // one 32-pixel unit per wave per feature row
constexpr int NTHREADS = NWAVES * 64;

enum StripMode { MODE_FUSED = 0, MODE_L12 = 1, MODE_L3 = 2 };

// Packed MFMA A-operand fragments, [NFRAG][64 lanes] floats (see pack_fragments()).
// The layers are SCALED by exact powers of two so that every activation lies in [-1, 1] for any 8-bit input
// (layer 1 by 2^-e1, layer 2 by 2^-e2; pack_fragments() derives e1, e2 from the weights): ReLU is then the clamp
// bit of a packed multiply by 1.0 -- one instruction per TWO registers -- and the scaling changes no result bit
// (a power-of-two factor commutes with every rounding; the layer-3 weights carry 2^e2 back).
constexpr int NFRAG_L1 = 82;       // 2 channel tiles x 41 k-steps (81 taps + bias tap), x 2^-e1
constexpr int NFRAG_L2 = 32;       // 2 x 16 k-steps over the 64 layer-1 channels, x 2^(e1-e2)        (MODE_FUSED)
constexpr int NFRAG_L3 = 16;       // 16 k-steps over the 32 layer-2 channels, rows = 25 taps (+7 zero), x 2^e2 (MODE_FUSED)
constexpr int NFRAG_B2 = 16;       // layer-2 bias laid out like the accumulator (the MFMA chain starts from it), x 2^-e2
constexpr int NFRAG_L2U = 32;      // layer 2 with UNSCALED output, x 2^e1: MODE_L12 stores the reference's map
constexpr int NFRAG_B2U = 16;      // ... and its unscaled bias
constexpr int NFRAG_L3U = 16;      // layer 3 on the unscaled map from HBM (MODE_L3)
constexpr int FRAG_L2 = NFRAG_L1, FRAG_L3 = FRAG_L2 + NFRAG_L2, FRAG_B2 = FRAG_L3 + NFRAG_L3,
              FRAG_L2U = FRAG_B2 + NFRAG_B2, FRAG_B2U = FRAG_L2U + NFRAG_L2U, FRAG_L3U = FRAG_B2U + NFRAG_B2U;
constexpr int NFRAG = FRAG_L3U + NFRAG_L3U;

// MFMA 32x32 accumulator row held by register r on lane-half h, and the
// channel we ASSIGN to accumulator row i so that register r / half h holds
// channel 2r+h (then the next layer's k-steps walk channels in ascending
// order, the reference's summation order).
__host__ __device__ constexpr int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__host__ __device__ constexpr int row_chan(int i) { return 2 * ((i & 3) + 4 * (i >> 3)) + ((i >> 2) & 1); }

// Layer-3 tap held by accumulator ROW i: register r = (i&3)+4(i>>3) of lane-half h = (i>>2)&1 holds
// tap (m, n) with r = 5s+m, n = s (half 0) or 3+s (half 1); -1 for the 7 unused rows.  Returns 5m+n.
__host__ __device__ constexpr int l3_row_tap(int i)
{
    const int r = (i & 3) + 4 * (i >> 3), h = (i >> 2) & 1, s = r / 5, m = r % 5, n = h ? 3 + s : s;
    return (r < 15 && n < 5) ? 5 * m + n : -1;
}
// Five of the 7 unused rows -- registers 10..14 of lane-half 1, which the vertical chains and the F tile treat as a sixth tap
// column (plane 5) anyway -- carry the LOCAL SCALE of SRCNN_MODE_REFBYTES' per-pixel flag threshold (round 6): weight
// a_c = max_tap |W3[c][tap]| in every one of them, so that plane 5 of a finished F-tile row holds, at no extra instruction,
// V(y, x) = sum over the 5 feature rows y-2..y+2 of U(., x), U = sum_c a_c * F_c, and the pixel's scale is the 5-term
// horizontal sum S1(y, x) = sum_n V(y, clamp(x + n - 2)).  Nothing else reads those rows: every output byte is as before.
__host__ __device__ constexpr bool l3_row_is_scale(int i)
{
    const int r = (i & 3) + 4 * (i >> 3), h = (i >> 2) & 1;
    return h == 1 && r >= 10 && r < 15;
}

// ---- split-f16 variant (srcnn_split16.hip): A-operand fragments of v_mfma_f32_32x32x16_f16 ----
// [S16_NFRAG][64 lanes][8 halfs]: L1 (tile t, part s = hi|lo, k-block b) at (2t+s)*6+b, L2 (s, b) at
// S16_FRAG_L2 + 4s+b, L3 (s, b) at S16_FRAG_L3 + 2s+b; then float b2[2 lane-halves][16 registers].
constexpr int S16_FRAG_L2 = 24, S16_FRAG_L3 = 32, S16_NFRAG = 36;
constexpr size_t S16_TABLE_BYTES = (size_t)S16_NFRAG * 64 * 16 + 2 * 16 * sizeof(float);
// Layer-1 K slot (k-block b, lane-half h, element e) -> tap 9*i+jj of the 9x9 window, 81 = the bias
// slot (B operand 1.0), -1 = padding (zero weight).  Register n = 4b + e/2 of the lane holds the f16
// pair (e even, e odd): n < 20 -> window row n/5 + 5h, columns 2(n%5), 2(n%5)+1; n = 20..22 -> row 4,
// columns 4h + 2(n-20) + (0,1), of which half 0 owns 0..3 and half 1 owns 4..8; n = 23 -> bias.
__host__ __device__ constexpr int l1s_tap(int b, int h, int e)
{
    const int n = 4 * b + e / 2, u = e & 1;
    if (n < 20) return (2 * (n % 5) + u < 9) ? 9 * (n / 5 + 5 * h) + 2 * (n % 5) + u : -1;
    if (n < 23) {
        const int jj = 4 * h + 2 * (n - 20) + u;
        return (h == 0 ? jj <= 3 : (jj >= 4 && jj <= 8)) ? 36 + jj : -1;
    }
    return (u == 0 && h == 0) ? 81 : -1;
}

struct StripParams {
    // layer-1 input (MODE_FUSED, MODE_L12)
    const uint8_t *src;
    long src_stride, src_frame_pitch;
    int src_row0;
    // Row stripes with their halo rows in SEPARATE buffers (srcnn_forward_y_rows_halo_dev, single frame): src holds image
    // rows [src_row0, src_row1), src_top rows [src_row0 - 6, src_row0), src_bot rows [src_row1, src_row1 + 6), both with
    // row stride halo_stride.  Either may be null when the launch reads no row on that side.  Both null: src holds every row.
    const uint8_t *src_top, *src_bot;
    long halo_stride;
    int src_row1;
    // layer-3 input (MODE_L3) / layer-2 output (MODE_L12): 32 planes in one allocation
    const float *planes_in;
    float *planes_out;
    long pl_stride, pl_pitch, pl_frame_pitch;
    // output (MODE_FUSED, MODE_L3)
    uint8_t *dst;
    float *pre;                     // optional pre-clamp f32, same strides as dst
    long dst_stride, dst_frame_pitch;
    int dst_row0;
    const float *wfrag;
    const uint32_t *wfrag16;        // split-f16 fragments (SRCNN_MODE_SPLIT16), else unused
    float *sink;                    // >= 128 floats of scratch for predicated-off stores
    float b3;
    int width, height;              // full image (border-replication domain)
    int row_begin, row_end;         // output rows produced by this launch
    int seg_rows, n_strips, n_segs; // workgroup decomposition
    const int *items;               // optional explicit work items, ITEM_INTS ints per block: {strip, row_begin,
                                    // row_end, seam above, seam below} (see plan_items()); nullptr = regular grid
    int items_per_frame;            // a batch repeats the plane's items frame after frame: block b = item b % items_per_frame
    int seams_per_frame;            // of frame b / items_per_frame, whose seam ids follow those of the frames before it
    float *seam;                    // seam scratch, SEAM_FLOATS * NTHREADS floats per seam (MODE_FUSED, items only)
    float *cseam;                   // column-seam scratch [frame][strip][row][CSEAM_FLOATS]; non-null = strips without column halo
    int strips_total;               // number of strips of the plane (n_strips is 1 in an items launch)
    int tune;                       // experiment switches (SRCNN_DEBUG_TUNE), 0 in production
    // SRCNN_MODE_REFBYTES (srcnn_exact.hip, "the reference's bytes"): a flag byte beside every output byte, same offsets as dst
    uint8_t *flag;                  // null = no flags
    float fix_delta, fix_scale;     // flag where |v - rint(v)| <= threshold <= fix_delta; code = 1 + (v - rint(v) + delta) * scale, scale = 253 / (2 delta)
    // the per-pixel threshold (float32 MFMA kernel): min(fix_delta, fix_kl * S1 + fix_abs), S1 = the pixel's local scale
    // (l3_row_is_scale()); fix_kl = 0, fix_abs = fix_delta: the one global threshold of rounds 3-5
    float fix_kl, fix_abs;
    unsigned *fix_counters;         // FixParams::counters: the strip kernel's first block zeroes them for the launch
};

// ---- SRCNN_MODE_REFBYTES: fix-up of the pixels whose MFMA value lies within delta of a truncation boundary (srcnn_exact.hip)
// Per-launch counter words (zeroed by the strip kernel), FIX_WORD_STRIDE words = 256 B apart so that atomics on different counters
// do not queue on one line, and the context's running totals (FIX_TOTALS 64-bit words of another buffer).  The work lists are filled in FIX_REGIONS regions, each
// with its own pair of reservation words: a returning atomic on ONE word sustains ~88 per microsecond chip-wide, and
// fix_collect_kernel makes one reservation per workgroup -- 900 of them on a 3840x2160 plane (one word: 16.3 us for the kernel,
// eight: 8.0).
enum { FIX_N_SCAT = 0, FIX_N_DENSE = 1, FIX_N_CHANGED = 2, FIX_MAX_DEV = 3, FIX_N_RERUN = 4,
       FIX_MAX_RATIO = 5,    // the largest |v_mfma - v_reference| / (the pixel's own threshold): what the verdict is taken from
       FIX_TOTALS = 6,
       FIX_WORD_STRIDE = 64,
       FIX_NEXT_ITEM = 6,    // item draw of fix_apply_kernel
       FIX_NEXT_RERUN = 7,   // tile draw of fix_rerun_kernel
       FIX_REGIONS = 8, FIX_REGION0 = 8 * FIX_WORD_STRIDE,      // region r: [FIX_REGION0 + r * FIX_WORD_STRIDE] scattered pixels, [+ 32] dense tiles
       FIX_COUNTERS = FIX_REGION0 + FIX_REGIONS * FIX_WORD_STRIDE };
struct FixParams {
    const uint8_t *src;             // the launch's Y input, as the strip kernel reads it
    long src_stride;
    int src_row0;
    const uint8_t *src_top, *src_bot;   // halo rows in separate buffers (StripParams::src_top), or null
    long halo_stride;
    int src_row1;
    uint8_t *dst;                   // the launch's output plane (already written by the strip and seam kernels)
    const uint8_t *flag;            // flag plane, same offsets as dst
    long dst_stride;
    int dst_row0;
    int width, height, row_begin, row_end;
    const float *wraw;              // b1|W1|b2|W2|b3|W3 in convdata.h order, then W2 transposed [64][32], then W1 transposed [81][64]
    unsigned *counters;             // FIX_COUNTERS words, zeroed by the strip kernel; counter k at counters[k * FIX_WORD_STRIDE]
    unsigned long long *totals;     // FIX_TOTALS 64-bit words, accumulated over every launch of the context (srcnn_fixup_stats)
    unsigned *scat, *dense;         // work lists in FIX_REGIONS equal regions: pixel (frame * height + y) * width + x; tile index
    float delta, code_step;         // code_step = 2 delta / 253
    float kl, abs_term;             // the strip kernel's per-pixel threshold min(delta, kl * S1 + abs_term) (StripParams::fix_kl, fix_abs)
    // The monitor ACTS, on the device: fix_rerun_kernel, queued behind fix_apply_kernel unconditionally, compares the launch's largest
    // |v_mfma - v_reference| / threshold(pixel) with rerun_above (1/2; negative = always: the test hook) and recomputes every pixel
    // of the launch in the reference's arithmetic when it is exceeded.
    float rerun_above;
    // one fix-up for the planes of several single-frame strip launches (srcnn_forward_y_dev): frame k of the batch lies at
    // src + k * src_frame_pitch / dst + k * dst_frame_pitch / flag + k * flag_frame_pitch, same rows in every frame
    int n_frames;
    long src_frame_pitch, dst_frame_pitch, flag_frame_pitch;
};
// SRCNN_MODE_REFBYTES: the flag byte stored beside an output byte (srcnn_kernels.h, srcnn_exact.hip): 0, or 1 + the position of
// v - rint(v) in [-delta, +delta] on a 253-step scale, for the values a rounding difference of the MFMA path could carry across
// a truncation boundary: |v - rint(v)| <= thr and 0.5 < v < 255.5 (the store truncates toward zero and clamps: (-1, 1) -> 0,
// >= 255 -> 255, so there is no boundary at 0 nor above 255).  The range test is ONE unsigned compare on the float's bits.
// thr <= delta is the pixel's own threshold (fix_threshold()); the CODE stays on the scale of the global delta, so the
// fix-up kernels decode every pixel the same way.
__device__ __forceinline__ float fix_threshold(float s1, float delta, float kl, float abs_term)
{
    return __builtin_fminf(delta, __builtin_fmaf(s1, kl, abs_term));
}
__device__ __forceinline__ uint8_t fix_code(float v, float thr, float delta, float scale)
{
    // (device code only; the host translation units include this header for the parameter structs)
    const float dist = v - __builtin_rintf(v);
    const bool live = (__builtin_fabsf(dist) <= thr) & ((__builtin_bit_cast(unsigned, v) - 0x3f000000u) < (0x437f8000u - 0x3f000000u));
    const unsigned code = (unsigned)((dist + delta) * scale + 1.5f);
    return live ? (uint8_t)code : (uint8_t)0;
}

// luma of image row yy, column xx of frame `frame`, wherever the launch keeps that row
__device__ __forceinline__ uint8_t fix_src_at(const FixParams &p, int frame, int yy, int xx)
{
    if (p.src_top && yy < p.src_row0) return p.src_top[(long)(yy - (p.src_row0 - 6)) * p.halo_stride + xx];
    if (p.src_bot && yy >= p.src_row1) return p.src_bot[(long)(yy - p.src_row1) * p.halo_stride + xx];
    return p.src[(long)frame * p.src_frame_pitch + (long)(yy - p.src_row0) * p.src_stride + xx];
}

// fix_collect_kernel, fix_apply_kernel, and behind them fix_rerun_kernel when `with_rerun` (srcnn_set_fixup_strict, the default)
// lds_weights: fix_apply_lds_kernel (both weight tables of layers 1-2 in LDS, 3 workgroups per CU) instead of the scalar-load form
hipError_t launch_fixup(const FixParams &p, int n_cu, bool with_rerun, bool lds_weights, hipStream_t st);
constexpr int FIX_BATCH_FRAMES = 16;      // frames per fix-up launch at most (pixel codes stay below 2^32 up to 16 x 16384 x 16384)
size_t fixup_list_entries(int width, int rows, int n_frames, size_t *dense_entries);

// A SEAM is the boundary between two vertically adjacent work items of a strip.  Instead of recomputing the
// two feature rows either side of it (4 rows per item), the item above hands over its 12 vertical-chain
// registers and the item below the tap partials of its first 4 rows that those chains still need (tap rows
// m > r of row r: 12 + 9 + 6 + 3 values); srcnn_seam_kernel replays those chain steps and finishes the 4 output
// rows around the seam -- same operations in the same order, so the result is bit-identical to the
// halo-recompute form.
constexpr int ITEM_INTS = 5;
constexpr int SEAM_R = 12, SEAM_ROWS = 4;
__host__ __device__ constexpr int seam_t_off(int r) { return SEAM_R + 3 * (r * 4 - r * (r - 1) / 2); }   // 12, 24, 33, 39
constexpr int SEAM_FLOATS = SEAM_R + 3 * (4 + 3 + 2 + 1);     // 42 per thread

size_t strip_lds_bytes(int mode);
hipError_t launch_seams(const StripParams &p, int n_seams, const int *d_seams, hipStream_t stream);

// COLUMN seams: with them a strip produces all FW = 128 columns it computes features for (no 2-column halo each
// side, 30 strips instead of 31 at 3840).  The four output pixels around a strip boundary need layer-3 column
// sums F_n from both strips: every strip exports, per output row, CSEAM_FLOATS values of its two edge columns
// (cseam_terms(): partial 5-term sums in the original order, or single F_n values) and srcnn_cseam_kernel
// completes those pixels -- same additions in the same order, bit-identical.
constexpr int CSEAM_FLOATS = 16;      // 15 used
hipError_t launch_cseams(const StripParams &p, int n_frames, hipStream_t stream);
// both in one launch; `winmap` [strips_total][rows] marks the rows inside a seam window of a strip (srcnn_plan.cpp, plan_items_balanced())
hipError_t launch_seams_merged(const StripParams &p, int n_seams, const int *d_seams, const unsigned char *d_winmap, int n_frames,
                               hipStream_t stream);
hipError_t launch_strip(int mode, const StripParams &p, int n_frames, hipStream_t stream, size_t lds_pad = 0);
// A fused float32 single-plane launch that also carries the seam blocks of the launch BEFORE it (seam deferral, srcnn_mfma.hip):
// blocks [0, first_block) are p's work items, the n_seams + cblocks behind them finish `prev`'s row and column seams.
struct FoldParams {
    StripParams prev;
    const int *seams;               // prev's seam table {strip, row} per seam
    const unsigned char *winmap;    // prev's rows inside a seam window, per strip; null: prev has row seams only (cblocks = 0)
    int n_seams, cblocks, first_block;
};
hipError_t launch_strip_fold(const StripParams &p, const FoldParams &f, hipStream_t stream, size_t lds_pad = 0);
// the same kernels with every MFMA <-> vector-ALU hazard visible to the compiler (srcnn_mfma.hip built with -DSRCNN_SAFE_HAZARDS=1)
hipError_t launch_strip_safe(int mode, const StripParams &p, int n_frames, hipStream_t stream, size_t lds_pad = 0);
// Does this device interlock the inline-asm MFMA -> packed multiply -> MFMA sequences of the fast row body?  Runs them with and
// without wait states (srcnn_probe.hip); returns the number of results that differ (0 = interlocked), negative on a HIP error.
long interlock_probe_mismatches(int device, hipStream_t stream);

size_t split16_lds_bytes();
hipError_t launch_split16(const StripParams &p, int n_frames, hipStream_t stream, size_t lds_pad = 0);

// ---- exact (vector-ALU, reference arithmetic) kernels (srcnn_exact.hip) ----
hipError_t launch_conv99_exact(const uint8_t *src, long sstride, float *dst, long dstride,
                               int w, int h, const float *d_kernel81, float bias, hipStream_t st);
hipError_t launch_conv11_exact(const float *planes, long stride, long pitch, float *dst, long dstride,
                               int w, int h, const float *d_kernel64, float bias, hipStream_t st);
hipError_t launch_conv99x11_exact(const uint8_t *src, long sstride, long src_frame_pitch,
                                  float *planes, long stride, long pitch, long frame_pitch,
                                  int w, int h, int n_frames, const float *d_weights, hipStream_t st);
hipError_t launch_conv55_exact(const float *planes, long stride, long pitch, long frame_pitch,
                               uint8_t *dst, float *pre, long dstride, long dst_frame_pitch,
                               int w, int h, int n_frames, const float *d_kernel800, float bias,
                               hipStream_t st);

// ---- pipeline steps around the conv path (srcnn_pipeline.hip) ---------------
hipError_t launch_copy_rows(uint8_t *dst, long dstride, const uint8_t *src, long sstride, int width, int rows, hipStream_t st);
hipError_t launch_bgr2ycrcb(const uint8_t *bgr, long stride, int w, int h, uint8_t *planes, long pstride,
                            long ppitch, hipStream_t st);
hipError_t launch_ycrcb2bgr(const uint8_t *y, long ystride, const uint8_t *crcb, long pstride, long ppitch, int w,
                            int h, uint8_t *bgr, long stride, hipStream_t st);
hipError_t launch_resize_cubic(const uint8_t *src, long sstride, long spitch, int sw, int sh, uint8_t *dst,
                               long dstride, long dpitch, int dw, int dh, int n_planes, const int *xofs,
                               const short *alpha, const int *yofs, const short *beta, hipStream_t st);

// the pipeline step in two launches: BGR -> up-sampled Y, and BGR + the conv path's Y -> up-sampled BGR
bool fused_pipeline_ok(int sw, int sh, int dw, int dh, const void *y_hi, long ystride, const void *out, long ostride);
hipError_t launch_bgr_to_y_resized(const uint8_t *bgr, long stride, int sw, int sh, uint8_t *dst, long dstride, int dw, int dh,
                                   const int *xofs, const short *alpha, const int *yofs, const short *beta, hipStream_t st);
hipError_t launch_resize_merge(const uint8_t *bgr, long stride, int sw, int sh, const uint8_t *ysr, long ystride, uint8_t *out,
                               long ostride, int dw, int dh, const int *xofs, const short *alpha, const int *yofs,
                               const short *beta, hipStream_t st);

}  // namespace srcnn
