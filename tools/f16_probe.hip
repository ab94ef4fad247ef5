// Micro-probe for the split-f16 kernel (srcnn_split16.hip): issue cost of the vector instructions it
// uses, of v_mfma_f32_32x32x16_f16 with vector fillers in its shadow, and whether a second wave on
// the same SIMD overlaps its vector work with the first wave's MFMAs.
// Build: hipcc --offload-arch=gfx950 -O3 -o build/f16_probe tools/f16_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)
__device__ __forceinline__ unsigned long long stamp()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

// role: 0 = MFMA chain, 1..6 = a stream of one vector instruction, 7 = MFMA + NF fillers per gap
template <int ROLE, int NF>
__device__ __forceinline__ void body(int iters, float &sink, u32x4 wa, u32x4 wb)
{
    f32x16 c = {0};
    float x0 = sink, x1 = sink + 1.f, x2 = sink + 2.f, x3 = sink + 3.f;
    unsigned u0 = 0x3c003c00u, u1 = 0x3c003c00u;
    for (int it = 0; it < iters; ++it) {
        if constexpr (ROLE == 0) {
#pragma unroll
            for (int r = 0; r < 64; ++r) c = MFMA16(wa, wb, c);
        } else if constexpr (ROLE == 1) {
            REP64(asm volatile("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1" : "+v"(x0), "+v"(x1));)
        } else if constexpr (ROLE == 2) {
            REP64(asm volatile("v_cvt_pkrtz_f16_f32 %0, %2, %3\n\tv_cvt_pkrtz_f16_f32 %1, %3, %2" : "=v"(u0), "=v"(u1) : "v"(x0), "v"(x1));)
        } else if constexpr (ROLE == 3) {
            REP64(asm volatile("v_fma_mixlo_f16 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %1, %2, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(u0), "+v"(u1) : "v"(u0), "v"(x0));)
        } else if constexpr (ROLE == 4) {
            REP64(asm volatile("v_pk_max_f16 %0, %0, 0\n\tv_pk_max_f16 %1, %1, 0" : "+v"(u0), "+v"(u1));)
        } else if constexpr (ROLE == 5) {
            REP64(asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(x0), "+v"(x1) : "v"(x2), "v"(x3));)
        } else if constexpr (ROLE == 6) {
            REP64(asm volatile("v_add_f32 %0, %0, %2\n\tv_add_f32 %1, %1, %2" : "+v"(x0), "+v"(x1) : "v"(x2));)
        } else if constexpr (ROLE == 7) {
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                c = MFMA16(wa, wb, c);
#pragma unroll
                for (int k = 0; k < NF; ++k) {          // four independent dependency chains
                    if ((k & 3) == 0) asm volatile("v_max_f32 %0, 0, %0" : "+v"(x0));
                    else if ((k & 3) == 1) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(u0) : "v"(u1), "v"(x1));
                    else if ((k & 3) == 2) asm volatile("v_max_f32 %0, 0, %0" : "+v"(x2));
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(x3) : "v"(x1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (ROLE == 9) {      // four independent v_max chains
            REP64(asm volatile("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3));)
        } else if constexpr (ROLE == 10) {     // split_pair as the kernel does it, two independent pairs
            REP64(asm volatile("v_cvt_pkrtz_f16_f32 %0, %2, %3\n\tv_cvt_pkrtz_f16_f32 %1, %3, %2\n\t"
                               "v_fma_mixlo_f16 %4, %0, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixlo_f16 %5, %1, -1.0, %3 op_sel_hi:[1,0,0]\n\t"
                               "v_fma_mixhi_f16 %4, %0, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %5, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                               : "=&v"(u0), "=&v"(u1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));)
        } else if constexpr (ROLE >= 11 && ROLE <= 19) {   // MFMA + the split instructions of srcnn_split16.hip per gap
            unsigned h0 = 0, h1 = 0, l0 = 0, l1 = 0;
#pragma unroll
            for (int r = 0; r < 64; ++r) {
                c = MFMA16(wa, wb, c);
                if constexpr (ROLE == 11) {        // cvt, cvt, pk_max, pk_max
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %2, %3\n\tv_cvt_pkrtz_f16_f32 %1, %4, %5\n\tv_pk_max_f16 %0, %0, 0\n\tv_pk_max_f16 %1, %1, 0"
                                 : "=&v"(h0), "=&v"(h1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
                    u0 ^= h0; u1 ^= h1;
                } else if constexpr (ROLE == 12) { // mixlo, mixlo, mixhi, mixhi
                    asm volatile("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0] clamp\n\tv_fma_mixlo_f16 %1, %3, -1.0, %5 op_sel_hi:[1,0,0] clamp\n\t"
                                 "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n\tv_fma_mixhi_f16 %1, %3, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp"
                                 : "+v"(l0), "+v"(l1) : "v"(u0), "v"(u1), "v"(x0), "v"(x1));
                } else if constexpr (ROLE == 13) { // cvt, cvt only
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %2, %3\n\tv_cvt_pkrtz_f16_f32 %1, %4, %5" : "=&v"(h0), "=&v"(h1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
                    u0 ^= h0; u1 ^= h1;
                } else if constexpr (ROLE == 14) { // four cvt
                    asm volatile("v_cvt_pkrtz_f16_f32 %0, %2, %3\n\tv_cvt_pkrtz_f16_f32 %1, %4, %5\n\tv_cvt_pkrtz_f16_f32 %0, %3, %2\n\tv_cvt_pkrtz_f16_f32 %1, %5, %4" : "=&v"(h0), "=&v"(h1) : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
                } else if constexpr (ROLE == 15) { // four pk_max
                    asm volatile("v_pk_max_f16 %0, %0, 0\n\tv_pk_max_f16 %1, %1, 0\n\tv_pk_max_f16 %2, %2, 0\n\tv_pk_max_f16 %3, %3, 0" : "+v"(u0), "+v"(u1), "+v"(l0), "+v"(l1));
                } else if constexpr (ROLE == 17) { // four independent mixlo
                    asm volatile("v_fma_mixlo_f16 %0, %4, -1.0, %6 op_sel_hi:[1,0,0] clamp\n\tv_fma_mixlo_f16 %1, %5, -1.0, %7 op_sel_hi:[1,0,0] clamp\n\t"
                                 "v_fma_mixlo_f16 %2, %4, -1.0, %7 op_sel_hi:[1,0,0] clamp\n\tv_fma_mixlo_f16 %3, %5, -1.0, %6 op_sel_hi:[1,0,0] clamp"
                                 : "+v"(l0), "+v"(l1), "+v"(h0), "+v"(h1) : "v"(u0), "v"(u1), "v"(x0), "v"(x1));
                } else if constexpr (ROLE == 18) { // four independent mixhi
                    asm volatile("v_fma_mixhi_f16 %0, %4, -1.0, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n\tv_fma_mixhi_f16 %1, %5, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n\t"
                                 "v_fma_mixhi_f16 %2, %4, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp\n\tv_fma_mixhi_f16 %3, %5, -1.0, %6 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp"
                                 : "+v"(l0), "+v"(l1), "+v"(h0), "+v"(h1) : "v"(u0), "v"(u1), "v"(x0), "v"(x1));
                } else if constexpr (ROLE == 19) { // four fma_f32 (the rescale + bias step)
                    asm volatile("v_fma_f32 %0, %0, %4, %5\n\tv_fma_f32 %1, %1, %4, %5\n\tv_fma_f32 %2, %2, %4, %5\n\tv_fma_f32 %3, %3, %4, %5"
                                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(sink), "v"(sink));
                } else {                           // three mix instructions
                    asm volatile("v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel_hi:[1,0,0] clamp\n\tv_fma_mixlo_f16 %1, %3, -1.0, %5 op_sel_hi:[1,0,0] clamp\n\t"
                                 "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0] clamp"
                                 : "+v"(l0), "+v"(l1) : "v"(u0), "v"(u1), "v"(x0), "v"(x1));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            u0 ^= l0; u1 ^= l1;
        } else if constexpr (ROLE == 8) {      // MFMA whose result is read by a vector instruction at once
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                c = MFMA16(wa, wb, c);
                asm volatile("v_max_f32 %0, 0, %1" : "=v"(x0) : "v"(c[0]));
                c[0] = x0;
            }
        }
    }
    sink = x0 + x1 + x2 + x3 + c[0] + c[5] + __builtin_bit_cast(float, u0) + __builtin_bit_cast(float, u1);
}

// waves 0..3 of the block run ROLE_A, waves 4..7 (if present) ROLE_B: wave w and w+4 share a SIMD
template <int ROLE_A, int NF_A, int ROLE_B>
__global__ __launch_bounds__(512, 1) void probe(float *out, unsigned long long *cyc, int iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float sink = 0.001f * lane;
    const u32x4 wa = {0x3c003c00u + lane, 0x38003800u, 0x3c003c00u, 0x34003400u};
    const u32x4 wb = {0x38003800u, 0x3c003c00u + lane, 0x34003400u, 0x3c003c00u};
    __syncthreads();
    const unsigned long long t0 = stamp();
    if (wave < 4) body<ROLE_A, NF_A>(iters, sink, wa, wb);
    else body<ROLE_B, 0>(iters, sink, wa, wb);
    const unsigned long long t1 = stamp();
    out[blockIdx.x * blockDim.x + threadIdx.x] = sink;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int ROLE_A, int NF_A, int ROLE_B>
void run(const char *name, int threads, double per_a, double per_b)
{
    const int blocks = 256, iters = 50;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, blocks * 512 * sizeof(float));
    hipMalloc(&cyc, blocks * 8 * sizeof(unsigned long long));
    hipMemset(cyc, 0, blocks * 8 * sizeof(unsigned long long));
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<ROLE_A, NF_A, ROLE_B>), dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256 * 8];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double a = 0, b = 0;
    for (int i = 0; i < blocks; ++i) {
        a += (h[i * 8] + h[i * 8 + 1] + h[i * 8 + 2] + h[i * 8 + 3]) / 4.0;
        b += (h[i * 8 + 4] + h[i * 8 + 5] + h[i * 8 + 6] + h[i * 8 + 7]) / 4.0;
    }
    a /= blocks;
    b /= blocks;
    printf("%-72s A: %7.2f cyc/op", name, a / (iters * per_a));
    if (threads > 256) printf("   B: %7.2f cyc/op", b / (iters * per_b));
    printf("\n");
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    run<0, 0, 0>("v_mfma_f32_32x32x16_f16 dependent chain, 1 wave/SIMD", 256, 64, 1);
    run<1, 0, 0>("v_max_f32 stream", 256, 128, 1);
    run<2, 0, 0>("v_cvt_pkrtz_f16_f32 stream", 256, 128, 1);
    run<3, 0, 0>("v_fma_mixlo/hi_f16 stream", 256, 128, 1);
    run<4, 0, 0>("v_pk_max_f16 stream", 256, 128, 1);
    run<5, 0, 0>("v_fma_f32 stream", 256, 128, 1);
    run<6, 0, 0>("v_add_f32 stream", 256, 128, 1);
    run<9, 0, 0>("v_max_f32, four independent chains", 256, 256, 1);
    run<10, 0, 0>("split of two pairs (2 cvt_pkrtz + 4 fma_mix), per instruction", 256, 384, 1);
    run<7, 0, 0>("MFMA + 0 fillers per gap (per MFMA)", 256, 64, 1);
    run<7, 2, 0>("MFMA + 2 fillers per gap", 256, 64, 1);
    run<7, 4, 0>("MFMA + 4 fillers per gap", 256, 64, 1);
    run<7, 5, 0>("MFMA + 5 fillers per gap", 256, 64, 1);
    run<7, 6, 0>("MFMA + 6 fillers per gap", 256, 64, 1);
    run<7, 8, 0>("MFMA + 8 fillers per gap", 256, 64, 1);
    run<7, 12, 0>("MFMA + 12 fillers per gap", 256, 64, 1);
    run<11, 0, 0>("MFMA + [cvt_pkrtz x2, pk_max x2] per gap", 256, 64, 1);
    run<12, 0, 0>("MFMA + [fma_mixlo x2, fma_mixhi x2] per gap", 256, 64, 1);
    run<13, 0, 0>("MFMA + [cvt_pkrtz x2] per gap", 256, 64, 1);
    run<14, 0, 0>("MFMA + [cvt_pkrtz x4] per gap", 256, 64, 1);
    run<15, 0, 0>("MFMA + [pk_max x4] per gap", 256, 64, 1);
    run<16, 0, 0>("MFMA + [fma_mix x3] per gap", 256, 64, 1);
    run<17, 0, 0>("MFMA + [fma_mixlo x4, independent] per gap", 256, 64, 1);
    run<18, 0, 0>("MFMA + [fma_mixhi x4, independent] per gap", 256, 64, 1);
    run<19, 0, 0>("MFMA + [v_fma_f32 x4] per gap", 256, 64, 1);
    run<8, 0, 0>("MFMA -> v_max of its result -> next MFMA (per MFMA)", 256, 16, 1);
    run<0, 0, 0>("2 waves/SIMD: A = MFMA chain, B = MFMA chain", 512, 64, 64);
    run<0, 0, 1>("2 waves/SIMD: A = MFMA chain, B = v_max stream", 512, 64, 128);
    run<0, 0, 3>("2 waves/SIMD: A = MFMA chain, B = v_fma_mix stream", 512, 64, 128);
    run<0, 0, 2>("2 waves/SIMD: A = MFMA chain, B = v_cvt_pkrtz stream", 512, 64, 128);
    run<1, 0, 1>("2 waves/SIMD: A = v_max stream, B = v_max stream", 512, 128, 128);
    run<7, 4, 1>("2 waves/SIMD: A = MFMA + 4 fillers, B = v_max stream", 512, 64, 128);
    run<7, 6, 7>("2 waves/SIMD: A = MFMA + 6 fillers, B = MFMA chain (0 fillers)", 512, 64, 64);
    run<9, 0, 9>("2 waves/SIMD: A = B = four v_max chains", 512, 256, 256);
    return 0;
}
