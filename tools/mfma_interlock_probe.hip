// mfma_interlock_probe.hip -- does the hardware interlock a vector instruction that reads an f32 MFMA result right behind it?
// The strip kernels' packed-ReLU inline asm reads accumulator registers a few cycles after the MFMA that writes them was
// issued, with no s_nop in between (the compiler's hazard recogniser does not look inside inline asm).  Here an asm MFMA is
// followed IMMEDIATELY (next instruction, 4 cycles later; the MFMA needs 64) by an asm v_add_f32 of its first result
// register, 4 chained times per thread, for one and for eight waves per workgroup; every result is compared with the value
// computed with the wait the ISA manual asks for.  Prints the number of mismatches (0 = interlocked).
// Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_interlock_probe.hip -o build/mfma_interlock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <bool WAIT>
__global__ void probe(const float *__restrict__ in, float *__restrict__ out, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float a = in[t], b = in[t + 4096], r = 0.f;
    for (int k = 0; k < iters; ++k) {
        f32x16 acc;
        asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
        if constexpr (WAIT) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // 32 wait states: far more than the manual's 18
        float s;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(s) : "v"(acc[0]), "v"(r));
        float u;
        asm volatile("v_add_f32 %0, %1, %2" : "=v"(u) : "v"(acc[15]), "v"(s));
        r = u;
        a = a * 0.5f + 0.25f;
        b = b + 1.0f;
    }
    out[t] = r;
}

// The EXACT instruction sequences of srcnn_strip_kernel's row body (srcnn_mfma.hip: relu_pairs, mfma_first, mfma_first0), whose
// operand dependencies are hidden from the compiler's hazard recogniser inside inline asm:
//   (1) MFMA -> v_pk_mul_f32 ... clamp rewriting the MFMA's result registers in place, all 8 pairs (relu_pairs), directly behind it;
//   (2) that packed multiply -> an asm MFMA that reads the just-rewritten register as its B operand and another accumulator as
//       C (mfma_first: the first layer-2 MFMA), followed by compiler-visible MFMAs on the other rewritten registers;
//   (3) the same again on the second accumulator -> an asm MFMA with a zero accumulator (mfma_first0: the first layer-3 MFMA).
// WAIT inserts 32 wait states between all of them -- far more than any hazard table asks for.  Launched with 256-thread workgroups,
// two per CU on every CU (the production occupancy), and 64 / 512 threads for comparison.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define PROBE_WAIT() do { if constexpr (WAIT) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); } while (0)

template <bool WAIT>
__global__ __launch_bounds__(512) void probe_sequences(const float *__restrict__ in, float *__restrict__ out, int iters)
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

int main()
{
    const int n = 256 * 512;
    std::vector<float> h(n + 4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 250.f - 2.f;
    float *d_in, *d_a, *d_b;
    if (hipMalloc(&d_in, h.size() * 4) || hipMalloc(&d_a, n * 4) || hipMalloc(&d_b, n * 4)) return 1;
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    long total_bad = 0;
    for (int threads : {64, 256, 512}) {
        hipLaunchKernelGGL(probe<false>, dim3(n / threads), dim3(threads), 0, 0, d_in, d_a, 200);
        hipLaunchKernelGGL(probe<true>, dim3(n / threads), dim3(threads), 0, 0, d_in, d_b, 200);
        std::vector<float> ra(n), rb(n);
        hipMemcpy(ra.data(), d_a, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(rb.data(), d_b, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; ++i) bad += (ra[i] != rb[i]);
        std::printf("%3d threads per workgroup: %ld of %d results differ between 'no wait' and 'waited'  (sample %.6g / %.6g)\n", threads, bad, n, ra[12345], rb[12345]);
        total_bad += bad;
    }
    // the kernel's own sequences; 256 threads x 512 workgroups = two workgroups on every CU at once
    for (int threads : {256, 64, 512}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipLaunchKernelGGL(probe_sequences<false>, dim3(n / threads), dim3(threads), 0, 0, d_in, d_a, 100);
            hipLaunchKernelGGL(probe_sequences<true>, dim3(n / threads), dim3(threads), 0, 0, d_in, d_b, 100);
            std::vector<float> ra(n), rb(n);
            hipMemcpy(ra.data(), d_a, n * 4, hipMemcpyDeviceToHost);
            hipMemcpy(rb.data(), d_b, n * 4, hipMemcpyDeviceToHost);
            long bad = 0, nonzero = 0;
            for (int i = 0; i < n; ++i) { bad += (ra[i] != rb[i]); nonzero += (rb[i] != 0.f); }
            std::printf("kernel sequences, %3d threads per workgroup, pass %d: %ld of %d results differ (%ld non-zero; sample %.6g / %.6g)\n",
                        threads, rep, bad, n, nonzero, ra[777], rb[777]);
            total_bad += bad + (nonzero < n / 2);
        }
    }
    std::printf("TOTAL mismatches: %ld\n", total_bad);
    return total_bad ? 2 : 0;
}
