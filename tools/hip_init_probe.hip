// hip_init_probe.hip -- what ANY process pays before its first kernel runs on an MI355X: the HIP runtime's start, the first
// allocation, the load of a code object and the first launch.  The baseline tests/checks/time_cli.py holds srcnn_create against.
//   hipcc --offload-arch=gfx950 -O2 tools/hip_init_probe.hip -o build/hip_init_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>

__global__ void touch(int *p) { p[threadIdx.x] = (int)threadIdx.x; }

int main()
{
    using clock = std::chrono::steady_clock;
    auto t0 = clock::now(), last = t0;
    auto mark = [&](const char *what) {
        const auto now = clock::now();
        std::printf("- timing : %-44s %9.3f ms   (at %9.3f)\n", what, std::chrono::duration<double, std::milli>(now - last).count(),
                    std::chrono::duration<double, std::milli>(now - t0).count());
        last = now;
    };
    int n = 0;
    if (hipInit(0) != hipSuccess || hipGetDeviceCount(&n) != hipSuccess || n <= 0) { std::printf("no device\n"); return 1; }
    mark("HIP runtime start (hipInit, device count)");
    hipStream_t st;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return 1;
    mark("first stream");
    int *d = nullptr;
    if (hipMalloc(&d, 4096) != hipSuccess) return 1;
    mark("first hipMalloc");
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, st, d);
    if (hipStreamSynchronize(st) != hipSuccess) return 1;
    mark("first launch (code-object load) + sync");
    hipLaunchKernelGGL(touch, dim3(1), dim3(64), 0, st, d);
    (void)hipStreamSynchronize(st);
    mark("second launch + sync");
    (void)hipFree(d);
    (void)hipStreamDestroy(st);
    return 0;
}
