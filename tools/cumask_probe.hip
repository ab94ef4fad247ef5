// cumask_probe.hip -- where do the blocks of a kernel launched on a CU-masked stream (hipExtStreamCreateWithCUMask) run?
// Prints, for a few masks, how many distinct (XCC, SE, CU) places the blocks landed on and the count per XCC: the order of the
// mask's bits over the 8 XCDs decides how srcnn_forward_y_unfused_dev splits the chip between the matrix-bound layer-1/2
// kernel and the HBM-bound layer-3 kernel.   Build: hipcc -O2 --offload-arch=gfx950 tools/cumask_probe.hip -o build/cumask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <set>
#include <vector>

__global__ void where(unsigned *out, int spin)
{
    if (threadIdx.x == 0) {
        const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_ID, 32 bits
        const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // XCC_ID
        out[2 * blockIdx.x] = hw;
        out[2 * blockIdx.x + 1] = xcc & 0xf;
    }
    // keep the block alive for a while so that every allowed CU gets blocks
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { }
}

int main()
{
    const int nblk = 4096;
    unsigned *d;
    if (hipMalloc(&d, nblk * 8) != hipSuccess) return 1;
    std::vector<unsigned> h(2 * nblk);
    struct M { const char *name; unsigned w[8]; } masks[] = {
        {"all 256", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}},
        {"bits 224..255", {0, 0, 0, 0, 0, 0, 0, ~0u}},
        {"bits 0..223", {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, 0}},
        {"bits 0..31", {~0u, 0, 0, 0, 0, 0, 0, 0}},
        {"every 8th bit (0, 8, 16 ...)", {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}},
    };
    for (const M &m : masks) {
        hipStream_t st;
        hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, m.w);
        if (e != hipSuccess) { std::printf("%s: hipExtStreamCreateWithCUMask failed: %s\n", m.name, hipGetErrorString(e)); continue; }
        hipMemsetAsync(d, 0xff, nblk * 8, st);
        hipLaunchKernelGGL(where, dim3(nblk), dim3(256), 0, st, d, 2000);      // 20 us per block
        hipStreamSynchronize(st);
        hipMemcpy(h.data(), d, nblk * 8, hipMemcpyDeviceToHost);
        std::set<unsigned long long> places;
        int per_xcc[16] = {0};
        std::set<unsigned> cu_ids[16];
        for (int b = 0; b < nblk; ++b) {
            const unsigned hw = h[2 * b], xcc = h[2 * b + 1];
            // HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
            const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
            places.insert(((unsigned long long)xcc << 32) | (se << 8) | (sh << 4) | cu);
            cu_ids[xcc].insert((se << 8) | (sh << 4) | cu);
            ++per_xcc[xcc];
        }
        std::printf("%-30s: %3zu distinct CUs; CUs per XCC:", m.name, places.size());
        for (int x = 0; x < 8; ++x) std::printf(" %zu", cu_ids[x].size());
        std::printf("   blocks per XCC:");
        for (int x = 0; x < 8; ++x) std::printf(" %d", per_xcc[x]);
        std::printf("\n");
        hipStreamDestroy(st);
    }
    return 0;
}
