// Calibration of FETCH_SIZE / WRITE_SIZE for the fused kernel's access widths
// (MI355X_MICROARCH.md "HBM": widths other than 16 B/lane are uncalibrated).
// Kernel A reads N bytes with one global_load_ubyte per lane (as the Y-row staging does),
// kernel B writes N bytes with one global_store_byte per lane (as the output store does).
// Run under rocprofv3 --pmc FETCH_SIZE and, separately, --pmc WRITE_SIZE.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read_bytes(const unsigned char *src, unsigned *sink, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
// kernel C reads N bytes with one global_load_dword per lane (128 B per half-wave, as the layer-3 kernel's plane loads
// and the seam kernels' scratch loads do), kernel D with one global_load_dwordx4 per lane (the guide's calibrated case)
__global__ void read_dwords(const unsigned *src, unsigned *sink, size_t n_words)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n_words; i += (size_t)gridDim.x * blockDim.x) acc += src[i];
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
__global__ void read_dwordx4(const uint4 *src, unsigned *sink, size_t n_vec)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned acc = 0;
    for (; i < n_vec; i += (size_t)gridDim.x * blockDim.x) { const uint4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 0xFFFFFFFFu) sink[0] = acc;
}
__global__ void write_dwords(unsigned *dst, size_t n_words)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n_words; i += (size_t)gridDim.x * blockDim.x) dst[i] = (unsigned)i;
}
__global__ void write_bytes(unsigned char *dst, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (unsigned char)i;
}
int main()
{
    const size_t n = (size_t)1 << 30;       // 1 GiB: well past the 256 MiB Infinity Cache
    unsigned char *a, *b; unsigned *s;
    if (hipMalloc(&a, n) || hipMalloc(&b, n) || hipMalloc(&s, 4)) return 1;
    (void)hipMemset(a, 1, n); (void)hipDeviceSynchronize();
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(read_bytes, dim3(4096), dim3(256), 0, 0, a, s, n);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(write_bytes, dim3(4096), dim3(256), 0, 0, b, n);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(read_dwords, dim3(4096), dim3(256), 0, 0, (const unsigned *)a, s, n / 4);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(read_dwordx4, dim3(4096), dim3(256), 0, 0, (const uint4 *)a, s, n / 16);
    for (int k = 0; k < 3; ++k) hipLaunchKernelGGL(write_dwords, dim3(4096), dim3(256), 0, 0, (unsigned *)b, n / 4);
    (void)hipDeviceSynchronize();
    printf("bytes per launch: %zu\n", n);
    return 0;
}
