// Micro-probe: issue cost of v_mfma_f32_32x32x2_f32 in the patterns the SRCNN
// kernel uses (dependent chain, two interleaved chains, chain fed by VALU results).
// Build: hipcc --offload-arch=gfx950 -O3 -o build/mfma_probe tools/mfma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__device__ __forceinline__ unsigned long long stamp()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
template <int VARIANT>
__global__ __launch_bounds__(256, 2) void probe(float *out, unsigned long long *cyc, int iters)
{
    __shared__ float lds[512];
    for (int q = threadIdx.x; q < 512; q += blockDim.x) lds[q] = 0.001f * q;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float a = 0.001f * lane, b = 0.002f * (lane + 1);
    f32x16 c0 = {0}, c1 = {0};
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = b + r;
    unsigned sacc = 0;
    const unsigned long long t0 = stamp();
    for (int it = 0; it < iters; ++it) {
        if constexpr (VARIANT == 0) {          // one dependent chain, 32 MFMA
#pragma unroll
            for (int r = 0; r < 32; ++r) c0 = MFMA(a, b, c0);
        } else if constexpr (VARIANT == 1) {   // two interleaved chains, 32 MFMA
#pragma unroll
            for (int r = 0; r < 16; ++r) { c0 = MFMA(a, b, c0); c1 = MFMA(b, a, c1); }
        } else if constexpr (VARIANT == 2) {   // dependent chain, B operand = fresh VALU result (ReLU)
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                float x = fmaxf(v[r & 15], 0.f);
                c0 = MFMA(a, x, c0);
                v[r & 15] = x + 1.0f;
            }
        } else if constexpr (VARIANT == 3) {   // dependent chain whose B operand is the OTHER chain's accumulator
#pragma unroll
            for (int r = 0; r < 16; ++r) c0 = MFMA(a, b, c0);
#pragma unroll
            for (int r = 0; r < 16; ++r) c1 = MFMA(a, fmaxf(c0[r], 0.f), c1);
        }
        else if constexpr (VARIANT == 4) {   // B from VALU, computed 2 k-steps AHEAD of its MFMA
            float x[34];
            x[0] = fmaxf(v[0], 0.f); x[1] = fmaxf(v[1], 0.f);
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                x[r + 2] = fmaxf(v[(r + 2) & 15], 0.f);
                c0 = MFMA(a, x[r], c0);
                v[r & 15] = x[r] + 1.0f;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (VARIANT == 5) {   // dependent chain + 6 independent VALU per MFMA
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                c0 = MFMA(a, b, c0);
#pragma unroll
                for (int q = 0; q < 6; ++q) v[(r + q) & 15] = v[(r + q) & 15] * 1.0001f + 0.5f;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (VARIANT == 6) {   // as 4 with a single-instruction ReLU
            float x[34];
            asm volatile("v_max_f32 %0, 0, %1" : "=v"(x[0]) : "v"(v[0]));
            asm volatile("v_max_f32 %0, 0, %1" : "=v"(x[1]) : "v"(v[1]));
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                asm volatile("v_max_f32 %0, 0, %1" : "=v"(x[r + 2]) : "v"(v[(r + 2) & 15]));
                c0 = MFMA(a, x[r], c0);
                v[r & 15] = x[r] + 1.0f;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (VARIANT == 7) {   // as 6 but only 1 step ahead
            float x[34];
            asm volatile("v_max_f32 %0, 0, %1" : "=v"(x[0]) : "v"(v[0]));
#pragma unroll
            for (int r = 0; r < 32; ++r) {
                asm volatile("v_max_f32 %0, 0, %1" : "=v"(x[r + 1]) : "v"(v[(r + 1) & 15]));
                c0 = MFMA(a, x[r], c0);
                v[r & 15] = x[r] + 1.0f;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        else if constexpr (VARIANT == 8) {   // two chains + 4 independent VALU per pair
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c0 = MFMA(a, b, c0); c1 = MFMA(b, a, c1);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[(r + q) & 15] = v[(r + q) & 15] * 1.0001f + 0.5f;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (VARIANT == 9) {   // two chains + 12 independent VALU per pair
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c0 = MFMA(a, b, c0); c1 = MFMA(b, a, c1);
#pragma unroll
                for (int q = 0; q < 12; ++q) v[(r + q) & 15] = v[(r + q) & 15] * 1.0001f + 0.5f;
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (VARIANT == 10) {  // two chains, B from LDS read 3 pairs ahead (as layer 1)
            float bq[19];
#pragma unroll
            for (int r = 0; r < 3; ++r) bq[r] = lds[lane + 64 * r];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                bq[r + 3] = lds[lane + 64 * ((r + 3) & 7)];
                c0 = MFMA(a, bq[r], c0); c1 = MFMA(b, bq[r], c1);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (VARIANT == 11) {  // three chains + 4 VALU per triple
            f32x16 c2 = {0};
#pragma unroll
            for (int r = 0; r < 10; ++r) {
                c0 = MFMA(a, b, c0); c1 = MFMA(b, a, c1); c2 = MFMA(a, a, c2);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[(r + q) & 15] = v[(r + q) & 15] * 1.0001f + 0.5f;
                __builtin_amdgcn_sched_barrier(0);
            }
            c0 = MFMA(a, b, c0); c1 = MFMA(b, a, c1);
            asm volatile("" : "+v"(c2));
            v[0] += c2[0];
        }
        else if constexpr (VARIANT == 12) {  // two chains + 12 SALU per pair
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c0 = MFMA(a, b, c0); c1 = MFMA(b, a, c1);
#pragma unroll
                for (int q = 0; q < 12; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (VARIANT == 13) {  // two chains + 6 LDS reads (+1 wait) per pair
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c0 = MFMA(a, b, c0); c1 = MFMA(b, a, c1);
                float x0 = lds[lane + 64 * (r & 7)], x1 = lds[lane + 1 + 64 * (r & 7)], x2 = lds[lane + 2 + 64 * (r & 7)];
                float x3 = lds[lane + 3 + 64 * (r & 7)], x4 = lds[lane + 4 + 64 * (r & 7)], x5 = lds[lane + 5 + 64 * (r & 7)];
                asm volatile("" :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5));
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (VARIANT == 14) {  // two chains + 6 v_pk_max_f32 per pair (12 registers)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c0 = MFMA(a, b, c0); c1 = MFMA(b, a, c1);
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 t = {v[2 * q], v[2 * q + 1]};
                    asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(t));
                    v[2 * q] = t[0]; v[2 * q + 1] = t[1];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (VARIANT == 15) {  // two chains + 12 v_max_f32 per pair
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c0 = MFMA(a, b, c0); c1 = MFMA(b, a, c1);
#pragma unroll
                for (int q = 0; q < 12; ++q) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[q]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("" : "+v"(c0), "+v"(c1));
    }
    asm volatile("" :: "s"(sacc));
    const unsigned long long t1 = stamp();
    float s = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + v[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V>
void run(const char *name, int blocks, int threads)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 200;
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(probe<V>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[4096]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double m = 0; for (int i = 0; i < blocks; ++i) m += h[i]; m /= blocks;
    printf("%-46s blocks=%4d thr=%3d : %.1f cycles per MFMA per wave\n", name, blocks, threads, m / (iters * 32.0));
    hipFree(out); hipFree(cyc);
}
int main()
{
    // 256 threads = 4 waves = 1 wave per SIMD; 512 blocks on 256 CUs = 2 waves per SIMD
    run<0>("dependent chain", 256, 256);
    run<1>("two interleaved chains", 256, 256);
    run<2>("dependent chain, B from VALU", 256, 256);
    run<3>("chain 2 consumes relu(chain 1 acc)", 256, 256);
    run<4>("B from VALU 2 steps ahead", 256, 256);
    run<5>("dependent chain + 6 independent VALU", 256, 256);
    run<6>("B from 1-instr ReLU 2 steps ahead", 256, 256);
    run<7>("B from 1-instr ReLU 1 step ahead", 256, 256);
    run<8>("two chains + 4 VALU per pair", 256, 256);
    run<9>("two chains + 12 VALU per pair", 256, 256);
    run<10>("two chains, B from LDS 3 pairs ahead", 256, 256);
    run<11>("three chains + 4 VALU per triple", 256, 256);
    run<12>("two chains + 12 SALU per pair", 256, 256);
    run<13>("two chains + 6 LDS reads per pair", 256, 256);
    run<14>("two chains + 6 v_pk_add (12 regs) per pair", 256, 256);
    run<15>("two chains + 12 v_max per pair", 256, 256);
    run<0>("dependent chain, 2 waves/SIMD", 512, 256);
    run<1>("two interleaved chains, 2 waves/SIMD", 512, 256);
    run<2>("dependent chain, B from VALU, 2 waves/SIMD", 512, 256);
    run<4>("B from VALU 2 steps ahead, 2 waves/SIMD", 512, 256);
    run<5>("chain + 6 independent VALU, 2 waves/SIMD", 512, 256);
    run<9>("two chains + 12 VALU per pair, 2 waves/SIMD", 512, 256);
    run<10>("two chains, B from LDS, 2 waves/SIMD", 512, 256);
    run<12>("two chains + 12 SALU per pair, 2 waves/SIMD", 512, 256);
    run<13>("two chains + 6 LDS reads per pair, 2 waves/SIMD", 512, 256);
    run<14>("two chains + 6 v_pk_add per pair, 2 waves/SIMD", 512, 256);
    run<15>("two chains + 12 v_max per pair, 2 waves/SIMD", 512, 256);
    return 0;
}
