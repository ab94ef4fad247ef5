// srcnn_probe.hip -- does this device interlock the instruction sequences the FAST strip kernels rely on?
//
// The fast row body (srcnn_mfma.hip, namespace srcnn::fast) hides three operand dependencies from the compiler's hazard
// recogniser inside inline asm -- the price of not padding 56 + 40 idle cycles per wave-row in front of the layer-2 and
// layer-3 MFMA chains:
//   (1) MFMA -> v_pk_mul_f32 ... clamp rewriting the MFMA's result registers in place, all 8 pairs (relu_pairs), directly behind it;
//   (2) that packed multiply -> an asm MFMA that reads the just-rewritten register as its B operand and another accumulator as
//       C (mfma_first: the first layer-2 MFMA), followed by compiler-visible MFMAs on the other rewritten registers;
//   (3) the same again on the second accumulator -> an asm MFMA with a zero accumulator (mfma_first0: the first layer-3 MFMA).
// Measured on MI355X: the hardware interlocks all of them (results are bit-identical with and without 32 wait states between
// every two instructions, at the production occupancy of two 256-thread workgroups per CU).  That is an observation, not a
// documented guarantee, so the library does not assume it: srcnn_create() runs THESE sequences both ways once per device and
// process (a millisecond) and the context falls back to the hazard-safe kernels (namespace srcnn::safe, ~3 % slower, same
// bytes) if a single result differs.  tools/mfma_interlock_probe.hip is the stand-alone form with more geometries.
#include "srcnn_kernels.h"

#include <vector>

namespace srcnn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PROBE_WAIT() do { if constexpr (WAIT) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); } while (0)

template <bool WAIT>
__global__ __launch_bounds__(256) void interlock_probe_kernel(const float *__restrict__ in, float *__restrict__ out, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a = in[t], b = in[t + 4096], w = in[t + 2048] * 0.25f, r = 0.f;
    const f32x2 ones = {1.0f, 1.0f};
    f32x16 cinit;
    for (int q = 0; q < 16; ++q) cinit[q] = in[(t + 64 * q) & 4095] * 0.125f;
    for (int k = 0; k < iters; ++k) {
        f32x16 acc, acc2, acc3;
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
        PROBE_WAIT();
#pragma unroll
        for (int q = 0; q < 8; ++q) {                                   // (1) relu_pairs
            f32x2 pr = {acc[2 * q], acc[2 * q + 1]};
            asm volatile("v_pk_mul_f32 %0, %0, %1 clamp" : "+v"(pr) : "s"(ones));
            acc[2 * q] = pr.x;
            acc[2 * q + 1] = pr.y;
        }
        PROBE_WAIT();
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %3" : "=&v"(acc2) : "v"(w), "v"(acc[0]), "v"(cinit));     // (2) mfma_first
        PROBE_WAIT();
#pragma unroll
        for (int q = 1; q < 16; ++q) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, acc[q], acc2, 0, 0, 0);
        PROBE_WAIT();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            f32x2 pr = {acc2[2 * q], acc2[2 * q + 1]};
            asm volatile("v_pk_mul_f32 %0, %0, %1 clamp" : "+v"(pr) : "s"(ones));
            acc2[2 * q] = pr.x;
            acc2[2 * q + 1] = pr.y;
        }
        PROBE_WAIT();
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(acc3) : "v"(w), "v"(acc2[0]));                  // (3) mfma_first0
        PROBE_WAIT();
#pragma unroll
        for (int q = 1; q < 16; ++q) acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, acc2[q], acc3, 0, 0, 0);
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += acc3[q] + acc2[q] * 0.5f + acc[q] * 0.25f;
        r = r * 0.5f + s;
        a = a * 0.5f + 0.25f * w;
        b = b * 0.75f + 0.1f;
    }
    out[t] = r;
}

// `st`: a stream of the caller's on `device` (srcnn_create passes the context's own non-blocking stream: launching on the legacy
// null stream would synchronise with every blocking stream of the host application).
long interlock_probe_mismatches(int device, hipStream_t st)
{
    // 512 workgroups of 256 threads: two on every CU at once, the strip kernels' occupancy
    constexpr int kThreads = 256, kBlocks = 512, n = kThreads * kBlocks;
    int prev = -1;
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (hipSetDevice(device) != hipSuccess) return -1;
    std::vector<float> h((size_t)n + 4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 250.f - 2.f;
    float *d_in = nullptr, *d_a = nullptr, *d_b = nullptr;
    long bad = -1;
    if (hipMalloc(&d_in, h.size() * 4) == hipSuccess && hipMalloc(&d_a, (size_t)n * 4) == hipSuccess &&
        hipMalloc(&d_b, (size_t)n * 4) == hipSuccess &&
        hipMemcpyAsync(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice, st) == hipSuccess) {
        bad = 0;
        std::vector<float> ra((size_t)n), rb((size_t)n);
        for (int rep = 0; rep < 2 && bad >= 0; ++rep) {
            hipLaunchKernelGGL(interlock_probe_kernel<false>, dim3(kBlocks), dim3(kThreads), 0, st, d_in, d_a, 24);
            hipLaunchKernelGGL(interlock_probe_kernel<true>, dim3(kBlocks), dim3(kThreads), 0, st, d_in, d_b, 24);
            if (hipGetLastError() != hipSuccess || hipMemcpyAsync(ra.data(), d_a, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipMemcpyAsync(rb.data(), d_b, (size_t)n * 4, hipMemcpyDeviceToHost, st) != hipSuccess ||
                hipStreamSynchronize(st) != hipSuccess) {
                bad = -1;
                break;
            }
            long nonzero = 0;
            for (int i = 0; i < n; ++i) {
                bad += ra[(size_t)i] != rb[(size_t)i];
                nonzero += rb[(size_t)i] != 0.f;
            }
            if (nonzero < n / 2) bad += n;          // a probe that computes nothing proves nothing
        }
    }
    (void)hipStreamSynchronize(st);
    if (d_in) (void)hipFree(d_in);
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (prev >= 0 && prev != device) (void)hipSetDevice(prev);
    return bad;
}

}  // namespace srcnn
