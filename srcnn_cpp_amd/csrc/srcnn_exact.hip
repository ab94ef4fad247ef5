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
// One pixel per lane, everything in registers: the 81 window pixels, and the 32
// layer-2 accumulators.  Layer 2 sums its inputs in ascending channel order
// (src/srcnn.cpp:312-315), so each layer-1 activation t_i is folded into the 32
// running sums as soon as it exists -- same order, same roundings, no 64-float
// scratch per pixel.  Weights are wave-uniform and come through the scalar
// cache: weights = b1[64] | W1[64][81] | b2[32] | W2[32][64] (convdata.h order),
// w2t = W2 transposed to [64][32] so that one channel's 32 weights are contiguous.
__global__ __launch_bounds__(256) void conv99x11_exact_kernel(const uint8_t *__restrict__ src, long sstride,
                                                              long src_frame_pitch,
                                                              float *__restrict__ planes, long stride,
                                                              long pitch, long frame_pitch, int w, int h,
                                                              const float *__restrict__ weights,
                                                              const float *__restrict__ w2t)
{
    const float *b1 = weights, *w1 = weights + 64, *b2 = w1 + 64 * 81;
    const int col = blockIdx.x * 64 + (threadIdx.x & 63);
    const int row = blockIdx.y * 4 + (threadIdx.x >> 6);
    const int frame = blockIdx.z;
    if (row >= h) return;
    const uint8_t *sf = src + (long)frame * src_frame_pitch;
    float px[81];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const uint8_t *sr = sf + (long)clampi_e(row + i - 4, 0, h - 1) * sstride;
#pragma unroll
        for (int j = 0; j < 9; ++j) px[i * 9 + j] = (float)sr[clampi_e(col + j - 4, 0, w - 1)];
    }
    float r[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) r[k] = 0.f;
#pragma unroll 1
    for (int i = 0; i < 64; ++i) {
        const float *wi = w1 + i * 81;
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 81; ++q) {
            const float pr = wi[q] * px[q];
            a = a + pr;
        }
        a = a + b1[i];
        a = (a < 0) ? 0.f : a;
        const float *w2i = w2t + i * 32;
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const float pr = a * w2i[k];
            r[k] = r[k] + pr;
        }
    }
    if (col >= w) return;
    float *o = planes + (long)frame * frame_pitch + (long)row * stride + col;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        float v = r[k] + b2[k];
        v = (v < 0) ? 0.f : v;
        o[(long)k * pitch] = v;
    }
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
                                                           const float *__restrict__ kernel, float bias)
{
    constexpr int TW = 64, TH = 4, WW = TW + 4, WH = TH + 4, WN = WW * WH;   // 68 x 8 window
    __shared__ float win[2][WN];
    const float *kw = kernel;          // wave-uniform index: the weights come through the scalar cache (from LDS: 2 % slower)
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int col0 = blockIdx.x * TW, row0 = blockIdx.y * TH;
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
        off[q] = (long)clampi_e(row0 + wy - 2, 0, h - 1) * stride + clampi_e(col0 + wx - 2, 0, w - 1);
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
#pragma unroll
        for (int m = 0; m < 5; ++m)
#pragma unroll
            for (int n = 0; n < 5; ++n) {
                const float pr = kw[(i * 5 + m) * 5 + n] * wv[m * WW + n];
                tp = tp + (double)pr;
            }
        temp = (float)((double)temp + tp);
        __syncthreads();
    }
    if (col >= w || row >= h) return;
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
    // d_weights = convdata.h-order table (8,129 floats) followed by W2 transposed ([64][32])
    hipLaunchKernelGGL(conv99x11_exact_kernel, px_grid(w, h, n_frames), dim3(256), 0, st, src, sstride,
                       src_frame_pitch, planes, stride, pitch, frame_pitch, w, h, d_weights, d_weights + 8129);
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
