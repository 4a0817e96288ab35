// mem_power_probe.hip -- package power against HBM traffic: a 16 B/lane streaming copy (optionally throttled by
// dependent VALU work per vector) runs for a few seconds while tools/ablate_run.py samples rocm-smi.
// Gives the energy per byte moved through L2/fabric/HBM that DESIGN.md's energy budget of the walk kernel uses.
// usage: mem_power_probe <mode> <spin> <seconds> [MiB per buffer]: small buffers stay in the 256 MiB Infinity Cache
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mem_power_probe tools/mem_power_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));

template <int MODE> // 0 copy, 1 read only, 2 write only
__global__ void __launch_bounds__(256) stream_kernel(const v2u64 *src, v2u64 *dst, size_t n, int spin) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    v2u64 acc = {0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        v2u64 v = {i, i};
        if (MODE != 2) v = __builtin_nontemporal_load(src + i);
        for (int s = 0; s < spin; s++) v.x = v.x * 0x9E3779B97F4A7C15ULL + v.y; // throttle: dependent 64-bit MADs
        if (MODE != 1) __builtin_nontemporal_store(v, dst + i);
        else acc += v;
    }
    if (MODE == 1 && acc.x == 0x1234567 && acc.y == 1) dst[0] = acc;
}

int main(int argc, char **argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    const int spin = argc > 2 ? atoi(argv[2]) : 0;
    const double seconds = argc > 3 ? atof(argv[3]) : 5.0;
    const size_t n = (argc > 4 ? (size_t)atof(argv[4]) : (size_t)2048) * 65536; // vectors per buffer; argv[4] = MiB per buffer (default 2 GiB)
    v2u64 *a, *b;
    hipMalloc(&a, n * 16);
    hipMalloc(&b, n * 16);
    hipMemset(a, 1, n * 16);
    hipMemset(b, 2, n * 16);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int blocks = 256 * 8;
    auto launch = [&]() {
        if (mode == 0) hipLaunchKernelGGL(stream_kernel<0>, dim3(blocks), dim3(256), 0, 0, a, b, n, spin);
        else if (mode == 1) hipLaunchKernelGGL(stream_kernel<1>, dim3(blocks), dim3(256), 0, 0, a, b, n, spin);
        else hipLaunchKernelGGL(stream_kernel<2>, dim3(blocks), dim3(256), 0, 0, a, b, n, spin);
    };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms1 = 0;
    hipEventElapsedTime(&ms1, e0, e1);
    const int reps = (int)(seconds * 1e3 / ms1) + 1;
    hipEventRecord(e0);
    for (int r = 0; r < reps; r++) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n * 16 * (mode == 0 ? 2 : 1) * reps;
    printf("mode %d (%s) spin %d, %zu MiB/buffer: %.3f TB/s over %.1f s\n", mode, mode == 0 ? "copy" : mode == 1 ? "read" : "write", spin, n / 65536, bytes / (ms * 1e-3) / 1e12, ms * 1e-3);
    return 0;
}
