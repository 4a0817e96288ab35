// clock_probe.hip -- what is the effective shader clock under integer VALU load, and what do
// v_add_u32 / v_mad_u64_u32 cost in SHADER cycles?  (measurement tool)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int OP>
__global__ void k(uint64_t *out, uint64_t *cyc, uint32_t seed, int iters) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 3, a4 = a0 ^ 0x55, a5 = a1 ^ 0x66, a6 = a2 ^ 0x77, a7 = a3 ^ 0x88;
    uint64_t c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a4, c5 = a5, c6 = a6, c7 = a7;
    uint32_t b = seed * 2654435761u + threadIdx.x;
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) {
                asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
                             "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if (OP == 1) {
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n\tv_mad_u64_u32 %1, vcc, %8, %9, %1\n\tv_mad_u64_u32 %2, vcc, %8, %9, %2\n\tv_mad_u64_u32 %3, vcc, %8, %9, %3\n\t"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\n\tv_mad_u64_u32 %5, vcc, %8, %9, %5\n\tv_mad_u64_u32 %6, vcc, %8, %9, %6\n\tv_mad_u64_u32 %7, vcc, %8, %9, %7"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(b), "v"(a0) : "vcc");
            } else if (OP == 2) { // dependent MAD chain (latency)
                asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                             "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0"
                             : "+v"(c0) : "v"(b), "v"(a0) : "vcc");
            } else if (OP == 3) { // dependent add chain
                asm volatile("v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\t"
                             "v_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %1"
                             : "+v"(a0) : "v"(b));
            } else if (OP == 4) { // addc with sgpr carry, independent
                asm volatile("v_addc_co_u32 %0, s[10:11], 0, %0, s[10:11]\n\tv_addc_co_u32 %1, s[12:13], 0, %1, s[12:13]\n\tv_addc_co_u32 %2, s[14:15], 0, %2, s[14:15]\n\tv_addc_co_u32 %3, s[16:17], 0, %3, s[16:17]\n\t"
                             "v_addc_co_u32 %4, s[18:19], 0, %4, s[18:19]\n\tv_addc_co_u32 %5, s[20:21], 0, %5, s[20:21]\n\tv_addc_co_u32 %6, s[22:23], 0, %6, s[22:23]\n\tv_addc_co_u32 %7, s[24:25], 0, %7, s[24:25]"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "s10","s11","s12","s13","s14","s15","s16","s17","s18","s19","s20","s21","s22","s23","s24","s25");
            } else if (OP == 5) { // s_nop 0 x8
                asm volatile("s_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0");
            } else if (OP == 6) { // v_mov
                asm volatile("v_mov_b32 %0, %1\n\tv_mov_b32 %1, %2\n\tv_mov_b32 %2, %3\n\tv_mov_b32 %3, %4\n\tv_mov_b32 %4, %5\n\tv_mov_b32 %5, %6\n\tv_mov_b32 %6, %7\n\tv_mov_b32 %7, %0"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    out[t] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    uint64_t *out, *cyc; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 8)); CK(hipMalloc(&cyc, cus * 8 * 8));
    const char *names[] = {"v_add_u32 x8 indep", "v_mad_u64_u32 x8 indep", "v_mad_u64_u32 dependent", "v_add_u32 dependent", "v_addc_co_u32 sgpr-carry indep", "s_nop 0", "v_mov_b32"};
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int wps = 1; wps <= 4; wps *= 2) {
        for (int op = 0; op < 7; op++) {
            const int blocks = cus * wps, threads = 256, iters = 4000;
            auto launch = [&] {
                switch (op) {
                case 0: hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1u, iters); break;
                case 1: hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1u, iters); break;
                case 2: hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1u, iters); break;
                case 3: hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1u, iters); break;
                case 4: hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1u, iters); break;
                case 5: hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1u, iters); break;
                default: hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1u, iters); break;
                }
            };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0, 0)); launch(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            uint64_t h[8]; CK(hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost));
            const double n_instr = (double)iters * 64; // wave-instructions per wave
            printf("waves/SIMD %d  %-32s: %8.3f ms  s_memtime %9llu ticks -> %6.2f ticks/instr/wave, %6.2f ticks per instr per SIMD; memtime rate %.1f MHz\n",
                   wps, names[op], ms, (unsigned long long)h[0], (double)h[0] / n_instr, (double)h[0] / n_instr / wps, (double)h[0] / (ms * 1e3));
        }
    }
    return 0;
}
