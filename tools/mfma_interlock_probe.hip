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

int main()
{
    const int n = 256 * 512;
    std::vector<float> h(n + 4096);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 250.f - 2.f;
    float *d_in, *d_a, *d_b;
    if (hipMalloc(&d_in, h.size() * 4) || hipMalloc(&d_a, n * 4) || hipMalloc(&d_b, n * 4)) return 1;
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int threads : {64, 256, 512}) {
        hipLaunchKernelGGL(probe<false>, dim3(n / threads), dim3(threads), 0, 0, d_in, d_a, 200);
        hipLaunchKernelGGL(probe<true>, dim3(n / threads), dim3(threads), 0, 0, d_in, d_b, 200);
        std::vector<float> ra(n), rb(n);
        hipMemcpy(ra.data(), d_a, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(rb.data(), d_b, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; ++i) bad += (ra[i] != rb[i]);
        std::printf("%3d threads per workgroup: %ld of %d results differ between 'no wait' and 'waited'  (sample %.6g / %.6g)\n", threads, bad, n, ra[12345], rb[12345]);
    }
    return 0;
}
