// pk_f32_probe.hip -- issue rate of the packed float32 vector instructions on gfx950, for the exact kernels
// (srcnn_exact.hip: the reference's multiply-THEN-add arithmetic, no FMA): is  v_pk_mul_f32 + v_pk_add_f32  on two channels
// cheaper than two  v_mul_f32 + v_add_f32?  Every variant runs the same number of float32 operations per lane on 16 independent
// accumulators (no dependent-issue stalls), with the multiplier in VGPRs or in an SGPR pair (the exact kernels hold the weights in
// SGPRs), for 1, 2 and 4 waves per SIMD on every CU.  Prints ns per 1,000 lane-operations per SIMD and the ratio to plain code.
// Build: hipcc -O3 --offload-arch=gfx950 -Wno-unused-value tools/pk_f32_probe.hip -o build/pk_f32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum { PLAIN_MULADD = 0, PLAIN_MULADD_S, PK_MULADD, PK_MULADD_S, PLAIN_FMA, PK_FMA, PK_MULADD_BCAST_S, PLAIN_CHAIN_S, N_VARIANTS };
static const char *kNames[N_VARIANTS] = {"v_mul_f32 + v_add_f32 (vgpr weights)", "v_mul_f32 + v_add_f32 (sgpr weights)",
                                         "v_pk_mul_f32 + v_pk_add_f32 (vgpr weights)", "v_pk_mul_f32 + v_pk_add_f32 (sgpr pair)",
                                         "v_fma_f32", "v_pk_fma_f32", "v_pk_mul (op_sel broadcast x, sgpr pair) + v_pk_add",
                                         "v_mul_f32 + v_add_f32, ONE dependent add chain (sgpr weights)"};

template <int V>
__global__ __launch_bounds__(256) void probe(const float *__restrict__ in, float *__restrict__ out, int iters)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = in[(t + q) & 4095];
    float x = in[(t * 7) & 4095];
    f32x2 xx = {x, x + 1.f};
    const float ws0 = in[blockIdx.x & 4095], ws1 = in[(blockIdx.x + 1) & 4095];          // wave-uniform: SGPRs
    const float wv0 = in[(t + 99) & 4095], wv1 = in[(t + 100) & 4095];
    f32x2 wsp = {__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ws0))),
                 __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, ws1)))};
    f32x2 wvp = {wv0, wv1};
    for (int k = 0; k < iters; ++k) {
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            if constexpr (V == PLAIN_MULADD) {
                float p0, p1;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(x), "v"(wv0));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(x), "v"(wv1));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[q]) : "v"(p0));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[q + 1]) : "v"(p1));
            } else if constexpr (V == PLAIN_MULADD_S) {
                float p0, p1;
                asm volatile("v_mul_f32 %0, %2, %1" : "=v"(p0) : "v"(x), "s"(wsp.x));
                asm volatile("v_mul_f32 %0, %2, %1" : "=v"(p1) : "v"(x), "s"(wsp.y));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[q]) : "v"(p0));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc[q + 1]) : "v"(p1));
            } else if constexpr (V == PLAIN_CHAIN_S) {        // what exact_layers12's tap loop is: a = a + w[q] * px[q]
                float p0, p1;
                asm volatile("v_mul_f32 %0, %2, %1" : "=v"(p0) : "v"(acc[q]), "s"(wsp.x));
                asm volatile("v_mul_f32 %0, %2, %1" : "=v"(p1) : "v"(acc[q + 1]), "s"(wsp.y));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(p0));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(p1));
            } else if constexpr (V == PK_MULADD) {
                f32x2 p, a = {acc[q], acc[q + 1]};
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(xx), "v"(wvp));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(p));
                acc[q] = a.x; acc[q + 1] = a.y;
            } else if constexpr (V == PK_MULADD_S) {
                f32x2 p, a = {acc[q], acc[q + 1]};
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(p) : "v"(xx), "s"(wsp));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(p));
                acc[q] = a.x; acc[q + 1] = a.y;
            } else if constexpr (V == PK_MULADD_BCAST_S) {
                f32x2 p, a = {acc[q], acc[q + 1]};
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "v"(xx), "s"(wsp));
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a) : "v"(p));
                acc[q] = a.x; acc[q + 1] = a.y;
            } else if constexpr (V == PLAIN_FMA) {
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[q]) : "v"(x), "v"(wv0));
                asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[q + 1]) : "v"(x), "v"(wv1));
            } else {
                f32x2 a = {acc[q], acc[q + 1]};
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(a) : "v"(xx), "v"(wvp));
                acc[q] = a.x; acc[q + 1] = a.y;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += acc[q];
    out[t] = s + x;
}

template <int V>
static float run(const float *d_in, float *d_out, int n_cu, int waves_per_simd, int iters)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = n_cu * waves_per_simd;                    // 256 threads = 4 waves = one per SIMD
    probe<V><<<blocks, 256>>>(d_in, d_out, iters);               // warm
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        probe<V><<<blocks, 256>>>(d_in, d_out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 0.5f + (float)(i % 17) * 0.03125f;
    float *d_in, *d_out;
    hipMalloc(&d_in, 4096 * 4);
    hipMalloc(&d_out, (size_t)n_cu * 8 * 256 * 4);
    hipMemcpy(d_in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    const int iters = 20000;
    // float32 operations (a multiply or an add; an FMA counts two) per lane per iteration: 32 in every variant
    printf("%d CUs; %d iterations x 32 float32 operations per lane; ms per launch (best of 5), cycles per lane-operation at 2.4 GHz\n", n_cu, iters);
    for (int w : {1, 2, 4, 5, 6, 8}) {
        float ms[N_VARIANTS];
        ms[0] = run<0>(d_in, d_out, n_cu, w, iters); ms[1] = run<1>(d_in, d_out, n_cu, w, iters);
        ms[2] = run<2>(d_in, d_out, n_cu, w, iters); ms[3] = run<3>(d_in, d_out, n_cu, w, iters);
        ms[4] = run<4>(d_in, d_out, n_cu, w, iters); ms[5] = run<5>(d_in, d_out, n_cu, w, iters);
        ms[6] = run<6>(d_in, d_out, n_cu, w, iters); ms[7] = run<7>(d_in, d_out, n_cu, w, iters);
        for (int v = 0; v < N_VARIANTS; ++v)
            printf("  %d wave(s)/SIMD  %-58s %8.3f ms  %6.2f wave-cycles per wave-op (64 lanes)  x%.2f of plain\n", w, kNames[v], ms[v],
                   ms[v] * 1e-3 * 2.4e9 / ((double)iters * 32.0 * w), ms[v] / ms[0]);
    }
    return 0;
}
