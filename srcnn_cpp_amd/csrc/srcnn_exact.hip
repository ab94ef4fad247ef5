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
// v_fmac may appear in these kernels' ISA (checked by tests/test_build.py).
#include "srcnn_kernels.h"

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
// One pixel per lane; the weights are read through the scalar cache
// (wave-uniform addresses), the 81 window pixels live in registers.
// weights = b1[64] | W1[64][81] | b2[32] | W2[32][64]  (convdata.h order).
__global__ __launch_bounds__(256) void conv99x11_exact_kernel(const uint8_t *__restrict__ src, long sstride,
                                                              long src_frame_pitch,
                                                              float *__restrict__ planes, long stride,
                                                              long pitch, long frame_pitch, int w, int h,
                                                              const float *__restrict__ weights)
{
    const float *b1 = weights, *w1 = weights + 64, *b2 = w1 + 64 * 81, *w2 = b2 + 32;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = blockIdx.z;
    if (row >= h) return;
    const bool ok = col < w;
    const uint8_t *sf = src + (long)frame * src_frame_pitch;
    float px[81];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint8_t *sr = sf + (long)clampi_e(row + i - 4, 0, h - 1) * sstride;
#pragma unroll
        for (int j = 0; j < 9; ++j) px[i * 9 + j] = (float)sr[clampi_e(col + j - 4, 0, w - 1)];
    }
    // layer-1 activations of this pixel go through LDS (column tid) so the
    // filter loop can stay rolled without dynamic register indexing.
    __shared__ float tl[64][256];
#pragma unroll 2
    for (int k = 0; k < 64; ++k) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 81; ++q) {
            const float pr = w1[k * 81 + q] * px[q];
            a = a + pr;
        }
        a = a + b1[k];
        tl[k][threadIdx.x] = (a < 0) ? 0.f : a;
    }
    float *o = planes + (long)frame * frame_pitch + (long)row * stride + col;
#pragma unroll 1
    for (int k = 0; k < 32; ++k) {
        float r = 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            const float pr = tl[i][threadIdx.x] * w2[k * 64 + i];
            r = r + pr;
        }
        r = r + b2[k];
        r = (r < 0) ? 0.f : r;
        if (ok) o[(long)k * pitch] = r;
    }
}

// ---- Convolution55, exact ---------------------------------------------------
// float product, double 25-term sum per channel, float running sum over the
// channels (src/srcnn.cpp:218-240).
__global__ __launch_bounds__(256) void conv55_exact_kernel(const float *__restrict__ planes, long stride,
                                                           long pitch, long frame_pitch,
                                                           uint8_t *__restrict__ dst, float *__restrict__ pre,
                                                           long dstride, long dst_frame_pitch, int w, int h,
                                                           const float *__restrict__ kernel, float bias)
{
    __shared__ float kw[800];
    for (int q = threadIdx.x; q < 800; q += 256) kw[q] = kernel[q];
    __syncthreads();
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = blockIdx.z;
    if (col >= w || row >= h) return;
    int rr[5], cc[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        rr[q] = clampi_e(row + q - 2, 0, h - 1);
        cc[q] = clampi_e(col + q - 2, 0, w - 1);
    }
    const float *pf = planes + (long)frame * frame_pitch;
    float temp = 0.f;
    for (int i = 0; i < 32; ++i) {
        const float *pl = pf + (long)i * pitch;
        double tp = 0.0;
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const float pr = kw[(i * 5 + m) * 5 + n] * pl[(long)rr[m] * stride + cc[n]];
                tp = tp + (double)pr;
            }
        temp = (float)((double)temp + tp);
    }
    temp = temp + bias;
    const long o = (long)frame * dst_frame_pitch + (long)row * dstride + col;
    if (pre) pre[o] = temp;
    int q = (int)temp;                       // truncation toward zero, src/srcnn.cpp:238
    q = clampi_e(q, 0, 255);
    dst[o] = (uint8_t)q;
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
    hipLaunchKernelGGL(conv99x11_exact_kernel, px_grid(w, h, n_frames), dim3(256), 0, st, src, sstride,
                       src_frame_pitch, planes, stride, pitch, frame_pitch, w, h, d_weights);
    return hipGetLastError();
}

hipError_t launch_conv55_exact(const float *planes, long stride, long pitch, long frame_pitch, uint8_t *dst,
                               float *pre, long dstride, long dst_frame_pitch, int w, int h, int n_frames,
                               const float *d_kernel800, float bias, hipStream_t st)
{
    hipLaunchKernelGGL(conv55_exact_kernel, px_grid(w, h, n_frames), dim3(256), 0, st, planes, stride, pitch,
                       frame_pitch, dst, pre, dstride, dst_frame_pitch, w, h, d_kernel800, bias);
    return hipGetLastError();
}

}  // namespace srcnn
