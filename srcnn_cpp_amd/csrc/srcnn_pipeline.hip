// srcnn_pipeline.hip -- the steps either side of the conv path in the
// reference's pipeline driver, as HBM-bound
// byte kernels:
//   cvtColor BGR->YCrCb + split     src/srcnn.cpp:509,540
//   resize INTER_CUBIC (3 planes)   src/srcnn.cpp:568-583
//   merge + cvtColor YCrCb->BGR     src/srcnn.cpp:639,657
// The arithmetic is OpenCV 4.x's 8-bit integer definition (fixed-point colour
// coefficients, 11-bit cubic coefficients, (sum + 2^21) >> 22), restated in
// oracle/opencv_steps.c; it is integer work, so the kernels are bit-exact
// against that restatement.  One byte in, one byte out per element: no reuse,
// no LDS, consecutive lanes on consecutive bytes.
#include "srcnn_kernels.h"

// No FMA contraction anywhere in this file (also given on the command line, srcnn_cpp_amd/build.py): the vertical pass of
// the resize reproduces OpenCV's separately rounded float32 products and sums.
#pragma clang fp contract(off)

namespace srcnn {

__device__ __forceinline__ int descale14(int x) { return (x + (1 << 13)) >> 14; }
__device__ __forceinline__ uint8_t sat8(int v) { return (uint8_t)min(max(v, 0), 255); }

__global__ __launch_bounds__(256) void bgr2ycrcb_kernel(const uint8_t *__restrict__ bgr, long stride, int w, int h,
                                                        uint8_t *__restrict__ planes, long pstride, long ppitch)
{
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= w || r >= h) return;
    const uint8_t *p = bgr + (long)r * stride + 3L * c;
    const int B = p[0], G = p[1], R = p[2];
    const int Y = descale14(B * 1868 + G * 9617 + R * 4899);
    const long o = (long)r * pstride + c;
    planes[o] = sat8(Y);
    planes[ppitch + o] = sat8(descale14((R - Y) * 11682 + (128 << 14)));
    planes[2 * ppitch + o] = sat8(descale14((B - Y) * 9241 + (128 << 14)));
}

__global__ __launch_bounds__(256) void ycrcb2bgr_kernel(const uint8_t *__restrict__ yp, long ystride,
                                                        const uint8_t *__restrict__ crcb, long pstride, long ppitch,
                                                        int w, int h, uint8_t *__restrict__ bgr, long stride)
{
    const int c = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (c >= w || r >= h) return;
    const int Y = yp[(long)r * ystride + c];
    const long o = (long)r * pstride + c;
    const int Cr = crcb[o] - 128, Cb = crcb[ppitch + o] - 128;
    uint8_t *p = bgr + (long)r * stride + 3L * c;
    p[0] = sat8(Y + descale14(Cb * 29049));
    p[1] = sat8(Y + descale14(Cb * -5636 + Cr * -11698));
    p[2] = sat8(Y + descale14(Cr * 22987));
}

// Vertical pass of cv::resize(INTER_CUBIC), 8-bit (OpenCV 4.x modules/imgproc/src/resize.cpp, x86 baseline build).
// h0..h3 are the int sums of the horizontal pass for the four source rows, b0..b3 the 11-bit row coefficients.
// Columns below dw - dw % 8 go through the SIMD functor VResizeCubicVec_32s8u, which works in float32:
//   r = h3*(b3*2^-22);  r = h2*(b2*2^-22) + r;  r = h1*(b1*2^-22) + r;  r = h0*(b0*2^-22) + r
// with every product and every sum rounded on its own (the SSE baseline's v_muladd is mul + add: no contraction
// here either), then round-to-nearest-even and saturate; the remaining dw % 8 columns take the scalar fixed-point
// cast (sum + 2^21) >> 22.  This is the variant that reproduces the reference's published picture bit for bit
// (oracle/opencv_steps.c, tests/test_pipeline_oracle.py).
__device__ __forceinline__ unsigned vresize_px(int h0, int h1, int h2, int h3, int b0, int b1, int b2, int b3, bool simd)
{
    if (simd) {
        const float sc = 1.0f / (2048.0f * 2048.0f);
        float r = __fmul_rn((float)h3, (float)b3 * sc);
        r = __fadd_rn(__fmul_rn((float)h2, (float)b2 * sc), r);
        r = __fadd_rn(__fmul_rn((float)h1, (float)b1 * sc), r);
        r = __fadd_rn(__fmul_rn((float)h0, (float)b0 * sc), r);
        return sat8(__float2int_rn(r));
    }
    return sat8((h0 * b0 + h1 * b1 + h2 * b2 + h3 * b3 + (1 << 21)) >> 22);
}

// dst(dy,dx) = vresize_px( sum_kx alpha[dx][kx] * src[clamp(yofs[dy]-1+ky)][clamp(xofs[dx]-1+kx)], beta[dy][ky] );
// blockIdx.z = plane
__global__ __launch_bounds__(256) void resize_cubic_kernel(const uint8_t *__restrict__ src, long sstride, long spitch,
                                                           int sw, int sh, uint8_t *__restrict__ dst, long dstride,
                                                           long dpitch, int dw, int dh,
                                                           const int *__restrict__ xofs, const short *__restrict__ alpha,
                                                           const int *__restrict__ yofs, const short *__restrict__ beta)
{
    const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
    if (dx >= dw || dy >= dh) return;
    const uint8_t *s = src + (long)blockIdx.z * spitch;
    const int x0 = xofs[dx] - 1, y0 = yofs[dy] - 1;
    int a[4], xs[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        a[k] = alpha[4 * dx + k];
        xs[k] = min(max(x0 + k, 0), sw - 1);
    }
    int hs[4];
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const uint8_t *row = s + (long)min(max(y0 + ky, 0), sh - 1) * sstride;
        int t = 0;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) t += row[xs[kx]] * a[kx];
        hs[ky] = t;
    }
    const short *b = beta + 4 * dy;
    dst[(long)blockIdx.z * dpitch + (long)dy * dstride + dx] =
        (uint8_t)vresize_px(hs[0], hs[1], hs[2], hs[3], b[0], b[1], b[2], b[3], dx < dw - dw % 8);
}

// Tiled variant for up-scaling (and mild down-scaling): a workgroup produces RT output rows x 256
// columns.  The horizontal pass of every source row the tile touches is done ONCE into LDS (as the
// int sums OpenCV's HResize produces), the vertical pass then reads LDS: ~4 byte loads per output
// pixel instead of 16.  Same integer arithmetic, bit-identical results.
constexpr int RT = 8;        // output rows per workgroup
constexpr int RMAX = 16;     // source rows a tile may span (host checks before choosing this kernel)
constexpr int SMAX = 288;    // source columns a 256-wide tile may span

__global__ __launch_bounds__(256) void resize_cubic_tiled_kernel(const uint8_t *__restrict__ src, long sstride,
                                                                 long spitch, int sw, int sh,
                                                                 uint8_t *__restrict__ dst, long dstride, long dpitch,
                                                                 int dw, int dh, const int *__restrict__ xofs,
                                                                 const short *__restrict__ alpha,
                                                                 const int *__restrict__ yofs,
                                                                 const short *__restrict__ beta)
{
    __shared__ int hbuf[RMAX][256];
    __shared__ uint8_t sbuf[RMAX][SMAX];   // the source rows of the tile (replicate border applied), bytes
    const int dx0 = blockIdx.x * 256, dx = dx0 + threadIdx.x;
    const int dy0 = blockIdx.y * RT, dy1 = min(dy0 + RT, dh);
    const uint8_t *s = src + (long)blockIdx.z * spitch;
    const int r_lo = yofs[dy0] - 1, r_hi = yofs[dy1 - 1] + 2;     // unclamped source rows of this tile
    const int c_lo = xofs[dx0] - 1, c_hi = xofs[min(dx0 + 255, dw - 1)] + 2;   // ... and columns
    const int ncol = c_hi - c_lo + 1, nrow = r_hi - r_lo + 1;
    // stage: consecutive lanes fetch consecutive source bytes of a row
    for (int e = threadIdx.x; e < nrow * ncol; e += 256) {
        const int rr = e / ncol, cc = e - rr * ncol;
        sbuf[rr][cc] = s[(long)min(max(r_lo + rr, 0), sh - 1) * sstride + min(max(c_lo + cc, 0), sw - 1)];
    }
    __syncthreads();
    const int dxc = min(dx, dw - 1);
    const int x0 = xofs[dxc] - 1 - c_lo;
    int a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = alpha[4 * dxc + k];
    for (int rr = 0; rr < nrow; ++rr) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) t += sbuf[rr][x0 + k] * a[k];
        hbuf[rr][threadIdx.x] = t;
    }
    // every thread only reads back its own hbuf column: no second barrier needed
    if (dx >= dw) return;
    for (int dy = dy0; dy < dy1; ++dy) {
        const int j = yofs[dy] - 1 - r_lo;
        const short *b = beta + 4 * dy;
        dst[(long)blockIdx.z * dpitch + (long)dy * dstride + dx] =
            (uint8_t)vresize_px(hbuf[j][threadIdx.x], hbuf[j + 1][threadIdx.x], hbuf[j + 2][threadIdx.x],
                                hbuf[j + 3][threadIdx.x], b[0], b[1], b[2], b[3], dx < dw - dw % 8);
    }
}

// vertical pass of output row dy, columns dx .. dx+3 (thread tx of the tile), packed one byte per pixel.
// hbuf holds the horizontal sums as floats (exact); see vresize_px() for the two arithmetic variants.
__device__ __forceinline__ unsigned vpass4(const float (*hbuf)[256], int tx, int j, const short *__restrict__ b, int dx, int dw)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) h[k] = *reinterpret_cast<const f32x4 *>(&hbuf[j + k][4 * tx]);
    unsigned r = 0;
    if (dx + 3 < dw - dw % 8) {            // all four columns in the SIMD functor's range: float32, every product and sum rounded
        const float sc = 1.0f / (2048.0f * 2048.0f);
        const float b0 = (float)b[0] * sc, b1 = (float)b[1] * sc, b2 = (float)b[2] * sc, b3 = (float)b[3] * sc;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float v = __fmul_rn(h[3][c], b3);
            v = __fadd_rn(__fmul_rn(h[2][c], b2), v);
            v = __fadd_rn(__fmul_rn(h[1][c], b1), v);
            v = __fadd_rn(__fmul_rn(h[0][c], b0), v);
            r |= (unsigned)sat8(__float2int_rn(v)) << (8 * c);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            r |= vresize_px((int)h[0][c], (int)h[1][c], (int)h[2][c], (int)h[3][c], b[0], b[1], b[2], b[3], dx + c < dw - dw % 8) << (8 * c);
    }
    return r;
}

// Second tiling: a workgroup produces a 256 x 32 output tile and every thread FOUR adjacent pixels of 8 rows, so
// the vertical pass reads its taps as one 16-byte LDS word per row and stores one dword per row: a quarter of the
// store and LDS instructions of the kernel above (which remains the fallback for small scales / odd strides).
// Same arithmetic, bit-identical results.
#ifndef SRCNN_RT4
#define SRCNN_RT4 32
#endif
constexpr int RT4 = SRCNN_RT4;          // output rows per workgroup (four row groups of RPT rows)
constexpr int RPT = RT4 / 4;            // rows per thread
constexpr int RMAX4 = RT4 == 32 ? 28 : 20;     // source rows such a tile may span (host checks)

__global__ __launch_bounds__(256) void resize_cubic_tiled4_kernel(const uint8_t *__restrict__ src, long sstride,
                                                                  long spitch, int sw, int sh,
                                                                  uint8_t *__restrict__ dst, long dstride, long dpitch,
                                                                  int dw, int dh, const int *__restrict__ xofs,
                                                                  const short *__restrict__ alpha,
                                                                  const int *__restrict__ yofs,
                                                                  const short *__restrict__ beta)
{
    __shared__ __attribute__((aligned(16))) float hbuf[RMAX4][256];
    __shared__ uint8_t sbuf[RMAX4][SMAX];
    const int tid = threadIdx.x;
    const int dx0 = blockIdx.x * 256;
    const int dy0 = blockIdx.y * RT4, dy1 = min(dy0 + RT4, dh);
    const uint8_t *s = src + (long)blockIdx.z * spitch;
    const int r_lo = yofs[dy0] - 1, r_hi = yofs[dy1 - 1] + 2;
    const int c_lo = xofs[dx0] - 1, c_hi = xofs[min(dx0 + 255, dw - 1)] + 2;
    const int ncol = c_hi - c_lo + 1, nrow = r_hi - r_lo + 1;
    for (int e = tid; e < nrow * ncol; e += 256) {
        const int rr = e / ncol, cc = e - rr * ncol;
        sbuf[rr][cc] = s[(long)min(max(r_lo + rr, 0), sh - 1) * sstride + min(max(c_lo + cc, 0), sw - 1)];
    }
    __syncthreads();
    {   // horizontal pass: thread = output column, all source rows of the tile
        const int dxc = min(dx0 + tid, dw - 1);
        const int x0 = xofs[dxc] - 1 - c_lo;
        int a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) a[k] = alpha[4 * dxc + k];
        for (int rr = 0; rr < nrow; ++rr) {
            int t = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) t += sbuf[rr][x0 + k] * a[k];
            hbuf[rr][tid] = (float)t;      // exact (|t| < 2^24)
        }
    }
    __syncthreads();
    // vertical pass: thread (tx, ty) = columns 4tx..4tx+3 of rows dy0 + 8ty .. +7
    const int tx = tid & 63, ty = tid >> 6;
    const int dx = dx0 + 4 * tx;
    if (dx >= dw) return;
    for (int r = 0; r < RPT; ++r) {
        const int dy = dy0 + RPT * ty + r;
        if (dy >= dy1) break;
        const unsigned v = vpass4(hbuf, tx, yofs[dy] - 1 - r_lo, beta + 4 * dy, dx, dw);
        uint8_t *o = dst + (long)blockIdx.z * dpitch + (long)dy * dstride + dx;
        if (dx + 3 < dw) *reinterpret_cast<unsigned *>(o) = v;
        else
            for (int c = 0; c < 4 && dx + c < dw; ++c) o[c] = (uint8_t)(v >> (8 * c));
    }
}

// ---- the whole pipeline step in two launches (srcnn_process_bgr[_dev]) ------------------------------------------
// cvtColor + split + 3 x resize + (conv path) + merge + cvtColor as the reference runs them (src/srcnn.cpp:509-657) touch
// every plane twice more than needed: the low-resolution Y / Cr / Cb planes only feed the resize, the resized Cr / Cb only
// the final conversion.  bgr_to_y_resized_kernel converts while it stages its source tile (BGR in, up-sampled Y out);
// resize_merge_kernel does the same for Cr and Cb and turns them, with the conv path's Y, straight into BGR.  The
// arithmetic per value is that of the three separate kernels above -- bit-identical -- in the tiled4 layout
// (256 x 32 output tile per workgroup, four adjacent pixels of eight rows per thread).
__device__ __forceinline__ int bgr_to_comp(const uint8_t *px, int comp)
{
    const int B = px[0], G = px[1], R = px[2];
    const int Y = descale14(B * 1868 + G * 9617 + R * 4899);
    if (comp == 0) return sat8(Y);
    if (comp == 1) return sat8(descale14((R - Y) * 11682 + (128 << 14)));
    return sat8(descale14((B - Y) * 9241 + (128 << 14)));
}

struct TileGeom {
    int dx0, dy0, dy1, r_lo, c_lo, ncol, nrow;
};

// stage component `comp` of the BGR source tile into sbuf, horizontal pass into hbuf (both barriers included)
__device__ __forceinline__ void stage_and_hpass(const uint8_t *__restrict__ bgr, long stride, int sw, int sh, int dw, int comp,
                                                const TileGeom &g, const int *__restrict__ xofs,
                                                const short *__restrict__ alpha, uint8_t (*sbuf)[SMAX], float (*hbuf)[256])
{
    const int tid = threadIdx.x;
    for (int e = tid; e < g.nrow * g.ncol; e += 256) {
        const int rr = e / g.ncol, cc = e - rr * g.ncol;
        sbuf[rr][cc] = (uint8_t)bgr_to_comp(bgr + (long)min(max(g.r_lo + rr, 0), sh - 1) * stride + 3L * min(max(g.c_lo + cc, 0), sw - 1), comp);
    }
    __syncthreads();
    const int dxc = min(g.dx0 + tid, dw - 1);
    const int x0 = xofs[dxc] - 1 - g.c_lo;
    int a[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = alpha[4 * dxc + k];
    for (int rr = 0; rr < g.nrow; ++rr) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) t += sbuf[rr][x0 + k] * a[k];
        hbuf[rr][tid] = (float)t;          // |t| < 2^24: exact; converted once per source row instead of once per output row
    }
    __syncthreads();
}

__device__ __forceinline__ TileGeom tile_geom(int dw, int dh, const int *__restrict__ xofs, const int *__restrict__ yofs)
{
    TileGeom g;
    g.dx0 = blockIdx.x * 256;
    g.dy0 = blockIdx.y * RT4;
    g.dy1 = min(g.dy0 + RT4, dh);
    g.r_lo = yofs[g.dy0] - 1;
    g.c_lo = xofs[g.dx0] - 1;
    g.ncol = xofs[min(g.dx0 + 255, dw - 1)] + 2 - g.c_lo + 1;
    g.nrow = yofs[g.dy1 - 1] + 2 - g.r_lo + 1;
    return g;
}

__global__ __launch_bounds__(256) void bgr_to_y_resized_kernel(const uint8_t *__restrict__ bgr, long stride, int sw, int sh,
                                                               uint8_t *__restrict__ dst, long dstride, int dw, int dh,
                                                               const int *__restrict__ xofs, const short *__restrict__ alpha,
                                                               const int *__restrict__ yofs, const short *__restrict__ beta)
{
    __shared__ __attribute__((aligned(16))) float hbuf[RMAX4][256];
    __shared__ uint8_t sbuf[RMAX4][SMAX];
    const TileGeom g = tile_geom(dw, dh, xofs, yofs);
    stage_and_hpass(bgr, stride, sw, sh, dw, 0, g, xofs, alpha, sbuf, hbuf);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, dx = g.dx0 + 4 * tx;
    if (dx >= dw) return;
    for (int r = 0; r < RPT; ++r) {
        const int dy = g.dy0 + RPT * ty + r;
        if (dy >= g.dy1) break;
        const unsigned v = vpass4(hbuf, tx, yofs[dy] - 1 - g.r_lo, beta + 4 * dy, dx, dw);
        uint8_t *o = dst + (long)dy * dstride + dx;
        if (dx + 3 < dw) *reinterpret_cast<unsigned *>(o) = v;
        else
            for (int c = 0; c < 4 && dx + c < dw; ++c) o[c] = (uint8_t)(v >> (8 * c));
    }
}

__global__ __launch_bounds__(256) void resize_merge_kernel(const uint8_t *__restrict__ bgr, long stride, int sw, int sh,
                                                           const uint8_t *__restrict__ ysr, long ystride,
                                                           uint8_t *__restrict__ out, long ostride, int dw, int dh,
                                                           const int *__restrict__ xofs, const short *__restrict__ alpha,
                                                           const int *__restrict__ yofs, const short *__restrict__ beta)
{
    __shared__ __attribute__((aligned(16))) float hbuf[RMAX4][256];
    __shared__ uint8_t sbuf[RMAX4][SMAX];
    const TileGeom g = tile_geom(dw, dh, xofs, yofs);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6, dx = g.dx0 + 4 * tx;
    unsigned crcb[2][RPT];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        if (pl) __syncthreads();            // every thread has finished reading plane 0's sums
        stage_and_hpass(bgr, stride, sw, sh, dw, 1 + pl, g, xofs, alpha, sbuf, hbuf);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int dy = min(g.dy0 + RPT * ty + r, g.dy1 - 1);
            crcb[pl][r] = vpass4(hbuf, tx, yofs[dy] - 1 - g.r_lo, beta + 4 * dy, min(dx, dw - 1), dw);
        }
    }
    if (dx >= dw) return;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int dy = g.dy0 + RPT * ty + r;
        if (dy >= g.dy1) break;
        const uint8_t *yrow = ysr + (long)dy * ystride + dx;
        uint8_t *o = out + (long)dy * ostride + 3L * dx;
        const bool vec = dx + 3 < dw;
        unsigned yv = 0;
        if (vec) yv = *reinterpret_cast<const unsigned *>(yrow);
        else
            for (int c = 0; c < 4 && dx + c < dw; ++c) yv |= (unsigned)yrow[c] << (8 * c);
        uint8_t px[12];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int Y = (yv >> (8 * c)) & 255, Cr = (int)((crcb[0][r] >> (8 * c)) & 255) - 128,
                      Cb = (int)((crcb[1][r] >> (8 * c)) & 255) - 128;
            px[3 * c + 0] = sat8(Y + descale14(Cb * 29049));
            px[3 * c + 1] = sat8(Y + descale14(Cb * -5636 + Cr * -11698));
            px[3 * c + 2] = sat8(Y + descale14(Cr * 22987));
        }
        if (vec) {
            unsigned *o4 = reinterpret_cast<unsigned *>(o);
#pragma unroll
            for (int q = 0; q < 3; ++q)
                o4[q] = px[4 * q] | (px[4 * q + 1] << 8) | (px[4 * q + 2] << 16) | ((unsigned)px[4 * q + 3] << 24);
        } else {
            for (int c = 0; c < 4 && dx + c < dw; ++c)
                for (int q = 0; q < 3; ++q) o[3 * c + q] = px[3 * c + q];
        }
    }
}

// true when the two fused launches apply (the tiled4 geometry limits, dword-aligned rows)
bool fused_pipeline_ok(int sw, int sh, int dw, int dh, const void *y_hi, long ystride, const void *out, long ostride)
{
    const long cspan = (256L * sw + dw - 1) / dw + 5, span4 = ((long)RT4 * sh + dh - 1) / dh + 4;
    const bool aligned = ((reinterpret_cast<uintptr_t>(y_hi) | (uintptr_t)ystride | reinterpret_cast<uintptr_t>(out) | (uintptr_t)ostride) & 3) == 0;
    return span4 <= RMAX4 && cspan <= SMAX && aligned;
}

hipError_t launch_bgr_to_y_resized(const uint8_t *bgr, long stride, int sw, int sh, uint8_t *dst, long dstride, int dw, int dh,
                                   const int *xofs, const short *alpha, const int *yofs, const short *beta, hipStream_t st)
{
    hipLaunchKernelGGL(bgr_to_y_resized_kernel, dim3((dw + 255) / 256, (dh + RT4 - 1) / RT4), dim3(256), 0, st, bgr, stride, sw,
                       sh, dst, dstride, dw, dh, xofs, alpha, yofs, beta);
    return hipGetLastError();
}

hipError_t launch_resize_merge(const uint8_t *bgr, long stride, int sw, int sh, const uint8_t *ysr, long ystride, uint8_t *out,
                               long ostride, int dw, int dh, const int *xofs, const short *alpha, const int *yofs,
                               const short *beta, hipStream_t st)
{
    hipLaunchKernelGGL(resize_merge_kernel, dim3((dw + 255) / 256, (dh + RT4 - 1) / RT4), dim3(256), 0, st, bgr, stride, sw, sh,
                       ysr, ystride, out, ostride, dw, dh, xofs, alpha, yofs, beta);
    return hipGetLastError();
}

// A few rows of a byte plane, device to device on ONE device, as a kernel: the striped step's band assembly (12 + 12 rows per
// step).  A launch is queued like the kernels around it and never blocks the host; hipMemcpy2DAsync between device buffers was
// seen to, once a process keeps more streams busy than the device has hardware queues (profiles/r03/stripe_overhead.txt).
__global__ __launch_bounds__(256) void copy_rows_kernel(uint8_t *__restrict__ dst, long dstride, const uint8_t *__restrict__ src,
                                                        long sstride, int width)
{
    const int x = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x >= width) return;
    const uint8_t *s = src + (long)blockIdx.y * sstride + x;
    uint8_t *d = dst + (long)blockIdx.y * dstride + x;
    if (x + 4 <= width && (((size_t)s | (size_t)d) & 3) == 0) {
        *reinterpret_cast<unsigned *>(d) = *reinterpret_cast<const unsigned *>(s);
    } else {
        for (int b = 0; b < 4 && x + b < width; ++b) d[b] = s[b];
    }
}

hipError_t launch_copy_rows(uint8_t *dst, long dstride, const uint8_t *src, long sstride, int width, int rows, hipStream_t st)
{
    if (width <= 0 || rows <= 0) return hipSuccess;
    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((width + 1023) / 1024), (unsigned)rows), dim3(256), 0, st, dst, dstride, src,
                       sstride, width);
    return hipGetLastError();
}

hipError_t launch_bgr2ycrcb(const uint8_t *bgr, long stride, int w, int h, uint8_t *planes, long pstride,
                            long ppitch, hipStream_t st)
{
    hipLaunchKernelGGL(bgr2ycrcb_kernel, dim3((w + 255) / 256, h), dim3(256), 0, st, bgr, stride, w, h, planes,
                       pstride, ppitch);
    return hipGetLastError();
}

hipError_t launch_ycrcb2bgr(const uint8_t *y, long ystride, const uint8_t *crcb, long pstride, long ppitch, int w,
                            int h, uint8_t *bgr, long stride, hipStream_t st)
{
    hipLaunchKernelGGL(ycrcb2bgr_kernel, dim3((w + 255) / 256, h), dim3(256), 0, st, y, ystride, crcb, pstride,
                       ppitch, w, h, bgr, stride);
    return hipGetLastError();
}

hipError_t launch_resize_cubic(const uint8_t *src, long sstride, long spitch, int sw, int sh, uint8_t *dst,
                               long dstride, long dpitch, int dw, int dh, int n_planes, const int *xofs,
                               const short *alpha, const int *yofs, const short *beta, hipStream_t st)
{
    // a tile of RT output rows spans at most ceil(RT * sh / dh) + 4 source rows (4-tap support)
    const long span = ((long)RT * sh + dh - 1) / dh + 4;
    const long cspan = (256L * sw + dw - 1) / dw + 5;      // likewise for 256 output columns
    const long span4 = ((long)RT4 * sh + dh - 1) / dh + 4;
    const bool dword_ok = ((reinterpret_cast<uintptr_t>(dst) | (uintptr_t)dstride | (uintptr_t)dpitch) & 3) == 0;
    if (span4 <= RMAX4 && cspan <= SMAX && dword_ok)
        hipLaunchKernelGGL(resize_cubic_tiled4_kernel, dim3((dw + 255) / 256, (dh + RT4 - 1) / RT4, n_planes), dim3(256),
                           0, st, src, sstride, spitch, sw, sh, dst, dstride, dpitch, dw, dh, xofs, alpha, yofs, beta);
    else if (span <= RMAX && cspan <= SMAX)
        hipLaunchKernelGGL(resize_cubic_tiled_kernel, dim3((dw + 255) / 256, (dh + RT - 1) / RT, n_planes), dim3(256),
                           0, st, src, sstride, spitch, sw, sh, dst, dstride, dpitch, dw, dh, xofs, alpha, yofs, beta);
    else
        hipLaunchKernelGGL(resize_cubic_kernel, dim3((dw + 255) / 256, dh, n_planes), dim3(256), 0, st, src, sstride,
                           spitch, sw, sh, dst, dstride, dpitch, dw, dh, xofs, alpha, yofs, beta);
    return hipGetLastError();
}

}  // namespace srcnn
