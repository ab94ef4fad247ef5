// l3_pattern_probe.hip -- what HBM delivers for the ACCESS PATTERN of Convolution55 alone (MODE_L3 of srcnn_mfma.hip),
// with all arithmetic removed: 32 planar f32 planes of 3840x2160 (the reference's vector<Mat> layout), a workgroup of
// 4 waves walks down a strip and reads, per row and plane, one contiguous run of STRIPW floats; every lane issues 16
// dword loads per row (register r of lane-half h reads plane 2r+h), one row ahead, and writes one byte per pixel.
// STRIPW = 128 is the kernel's strip (512-byte runs); 256 / 512 show what wider strips would buy; "dwordx4" reads the
// same bytes with 16-byte loads (4 pixels per lane) as an upper bound for this layout.
// Build: hipcc -O3 --offload-arch=gfx950 tools/l3_pattern_probe.hip -o build/l3_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int W = 3840, H = 2160, NP = 32;

template <int UNITS, bool NT = false>   // UNITS 32-pixel units per wave and row: strip = 4 waves x UNITS x 32 columns; NT: non-temporal loads
__global__ __launch_bounds__(256) void walk(const float *__restrict__ planes, unsigned char *__restrict__ out, int seg_rows)
{
    constexpr int SW = 128 * UNITS;
    const int n_strips = W / SW;
    const int strip = blockIdx.x % n_strips, seg = blockIdx.x / n_strips;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, half = lane >> 5;
    const int y0 = seg * seg_rows, y1 = min(H, y0 + seg_rows);
    const long pitch = (long)W * H;
    float cur[UNITS][16], nxt[UNITS][16];
    auto load = [&](int y, float (&d)[UNITS][16]) {
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const float *q = planes + (long)min(y, H - 1) * W + strip * SW + (wave * UNITS + u) * 32 + j;
#pragma unroll
            for (int r = 0; r < 16; ++r) d[u][r] = NT ? __builtin_nontemporal_load(q + (long)(2 * r + half) * pitch) : q[(long)(2 * r + half) * pitch];
        }
    };
    load(y0, cur);
    for (int y = y0; y < y1; ++y) {
        load(y + 1, nxt);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            float a = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) a += cur[u][r];
            a += __shfl_xor(a, 32);
            if (half == 0) out[(long)y * W + strip * SW + (wave * UNITS + u) * 32 + j] = (unsigned char)a;
#pragma unroll
            for (int r = 0; r < 16; ++r) cur[u][r] = nxt[u][r];
        }
    }
}

// same bytes, 16-byte loads: a lane reads 4 consecutive pixels of one plane; a wave covers 256 columns of one plane per load
__global__ __launch_bounds__(256) void walk_x4(const float *__restrict__ planes, unsigned char *__restrict__ out, int seg_rows)
{
    constexpr int SW = 256;
    const int n_strips = W / SW;
    const int strip = blockIdx.x % n_strips, seg = blockIdx.x / n_strips;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int y0 = seg * seg_rows, y1 = min(H, y0 + seg_rows);
    const long pitch = (long)W * H;
    for (int y = y0; y < y1; ++y) {
        float4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < 8; ++r) {           // wave w reads planes 8w .. 8w+7
            const float4 v = *reinterpret_cast<const float4 *>(planes + (long)(8 * wave + r) * pitch + (long)y * W + strip * SW + 4 * lane);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        // every wave's loads are live: each writes a quarter of the row's bytes
        if ((lane >> 4) == wave) *reinterpret_cast<unsigned *>(out + (long)y * W + strip * SW + 4 * lane) = (unsigned)(acc.x + acc.y + acc.z + acc.w);
        else if (acc.x == 12345.f) out[0] = 1;
    }
}

// reference: the same bytes as ONE linear stream (every workgroup a contiguous chunk, 16-byte loads)
template <bool NT>
__global__ __launch_bounds__(256) void stream_x4(const float *__restrict__ planes, unsigned char *__restrict__ out, long n4_per_block)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 *p = reinterpret_cast<const f4 *>(planes) + (long)blockIdx.x * n4_per_block;
    f4 acc = {0, 0, 0, 0};
    for (long i = threadIdx.x; i < n4_per_block; i += 256) {
        const f4 v = NT ? __builtin_nontemporal_load(p + i) : p[i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.f) out[0] = 1;
}

template <typename F>
static void run(const char *name, F launch, int blocks)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int k = 0; k < 3; ++k) launch();
    hipEventRecord(a);
    const int reps = 10;
    for (int k = 0; k < reps; ++k) launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    ms /= reps;
    const double bytes = (double)W * H * (NP * 4 + 1);
    std::printf("%-44s %5d workgroups  %.3f ms  %.2f TB/s\n", name, blocks, ms, bytes / ms / 1e9);
}

int main()
{
    float *planes; unsigned char *out;
    // three independent frames so that successive launches do not re-read what the Infinity Cache still holds
    const size_t frame = (size_t)W * H * NP;
    if (hipMalloc(&planes, 3 * frame * sizeof(float)) || hipMalloc(&out, (size_t)W * H)) return 1;
    hipMemset(planes, 0, 3 * frame * sizeof(float));
    hipDeviceSynchronize();
    int k = 0;
    for (int seg_rows : {64, 135}) {
        const int segs = (H + seg_rows - 1) / seg_rows;
        char nm[96];
        std::snprintf(nm, sizeof nm, "512-B runs (128-col strips), %d-row segments", seg_rows);
        run(nm, [&] { hipLaunchKernelGGL(walk<1>, dim3(30 * segs), dim3(256), 0, 0, planes + (k++ % 3) * frame, out, seg_rows); }, 30 * segs);
        std::snprintf(nm, sizeof nm, "1-KB runs (256-col strips), %d-row segments", seg_rows);
        run(nm, [&] { hipLaunchKernelGGL(walk<2>, dim3(15 * segs), dim3(256), 0, 0, planes + (k++ % 3) * frame, out, seg_rows); }, 15 * segs);
        std::snprintf(nm, sizeof nm, "dwordx4, 1-KB runs, %d-row segments", seg_rows);
        run(nm, [&] { hipLaunchKernelGGL(walk_x4, dim3(15 * segs), dim3(256), 0, 0, planes + (k++ % 3) * frame, out, seg_rows); }, 15 * segs);
    }
    for (int seg_rows : {64, 135}) {
        const int segs = (H + seg_rows - 1) / seg_rows;
        char nm[96];
        std::snprintf(nm, sizeof nm, "NON-TEMPORAL 512-B runs, %d-row segments", seg_rows);
        run(nm, [&] { hipLaunchKernelGGL((walk<1, true>), dim3(30 * segs), dim3(256), 0, 0, planes + (k++ % 3) * frame, out, seg_rows); }, 30 * segs);
        std::snprintf(nm, sizeof nm, "NON-TEMPORAL 1-KB runs, %d-row segments", seg_rows);
        run(nm, [&] { hipLaunchKernelGGL((walk<2, true>), dim3(15 * segs), dim3(256), 0, 0, planes + (k++ % 3) * frame, out, seg_rows); }, 15 * segs);
    }
    {
        const int blocks = 4096;
        const long n4 = (long)frame / 4 / blocks;
        run("linear stream, 16-byte loads", [&] { hipLaunchKernelGGL(stream_x4<false>, dim3(blocks), dim3(256), 0, 0, planes + (k++ % 3) * frame, out, n4); }, blocks);
        run("linear stream, non-temporal 16-byte loads", [&] { hipLaunchKernelGGL(stream_x4<true>, dim3(blocks), dim3(256), 0, 0, planes + (k++ % 3) * frame, out, n4); }, blocks);
    }
    for (int seg_rows : {16, 32}) {
        const int segs = (H + seg_rows - 1) / seg_rows;
        char nm[96];
        std::snprintf(nm, sizeof nm, "512-B runs, %d-row segments (more workgroups)", seg_rows);
        run(nm, [&] { hipLaunchKernelGGL(walk<1>, dim3(30 * segs), dim3(256), 0, 0, planes + (k++ % 3) * frame, out, seg_rows); }, 30 * segs);
    }
    return 0;
}
