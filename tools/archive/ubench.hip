// ubench.hip -- integer-ALU micro-benchmarks for gfx950 (measurement tool, not part of the product).
// Answers SURVEY.md section 7 "hard parts": what does v_mad_u64_u32 cost on CDNA4, and what is the
// chip-wide ceiling of the 256-bit modular multiplication the walk kernel is made of?
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/ubench tools/ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../kangaroo_amd/csrc/kng_field.h"
#include "../kangaroo_amd/csrc/kng_modinv.h"
#include "fe_extras.h"
using namespace kng;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1);} } while (0)

template <int OP>
__global__ void k_instr(uint64_t *out, uint32_t seed, int iters) {
    uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 7, a3 = a0 * 7 + 3;
    uint64_t c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a0 ^ 0x1111, c5 = a1 ^ 0x2222, c6 = a2 ^ 0x3333, c7 = a3 ^ 0x4444;
    uint32_t b = seed * 2654435761u + 12345;
    double f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a0 + 1.5, f5 = a1 + 2.5, f6 = a2 + 3.5, f7 = a3 + 4.5, fb = 1.000000001;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            if (OP == 0) { // v_mad_u64_u32, 8 independent accumulators
                c0 = (uint64_t)(uint32_t)c0 * b + c0; c1 = (uint64_t)(uint32_t)c1 * b + c1;
                c2 = (uint64_t)(uint32_t)c2 * b + c2; c3 = (uint64_t)(uint32_t)c3 * b + c3;
                c4 = (uint64_t)(uint32_t)c4 * b + c4; c5 = (uint64_t)(uint32_t)c5 * b + c5;
                c6 = (uint64_t)(uint32_t)c6 * b + c6; c7 = (uint64_t)(uint32_t)c7 * b + c7;
            } else if (OP == 1) { // v_mul_lo_u32
                a0 *= b; a1 *= b; a2 *= b; a3 *= b;
                c0 = (uint32_t)c0 * (uint32_t)c4; c1 = (uint32_t)c1 * (uint32_t)c5; c2 = (uint32_t)c2 * (uint32_t)c6; c3 = (uint32_t)c3 * (uint32_t)c7;
            } else if (OP == 2) { // v_mul_hi_u32
                a0 = __umulhi(a0, b); a1 = __umulhi(a1, b); a2 = __umulhi(a2, b); a3 = __umulhi(a3, b);
                c0 = __umulhi((uint32_t)c0, b); c1 = __umulhi((uint32_t)c1, b); c2 = __umulhi((uint32_t)c2, b); c3 = __umulhi((uint32_t)c3, b);
            } else if (OP == 3) { // 64-bit add (v_lshl_add_u64 / add_co+addc)
                c0 += c1; c1 += c2; c2 += c3; c3 += c4; c4 += c5; c5 += c6; c6 += c7; c7 += c0;
            } else if (OP == 4) { // v_fma_f64
                f0 = f0 * fb + f1; f1 = f1 * fb + f2; f2 = f2 * fb + f3; f3 = f3 * fb + f4;
                f4 = f4 * fb + f5; f5 = f5 * fb + f6; f6 = f6 * fb + f7; f7 = f7 * fb + f0;
            } else if (OP == 5) { // 32-bit add (v_add_u32)
                a0 += a1; a1 += a2; a2 += a3; a3 += a0; a0 ^= a2; a1 ^= a3; a2 += b; a3 += b;
            } else if (OP == 6) { // v_mul_u32_u24-ish: 24-bit multiply
                a0 = __umul24(a0, b); a1 = __umul24(a1, b); a2 = __umul24(a2, b); a3 = __umul24(a3, b);
                c0 = __umul24((uint32_t)c0, b); c1 = __umul24((uint32_t)c1, b); c2 = __umul24((uint32_t)c2, b); c3 = __umul24((uint32_t)c3, b);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + a0 + a1 + a2 + a3 +
        (uint64_t)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7);
}

template <int OP>
__global__ void k_field(uint64_t *out, uint64_t seed, int iters) {
    const uint64_t t = blockIdx.x * blockDim.x + threadIdx.x;
    fe a{{seed + t, seed * 3 + t, seed * 5 + 1, seed * 7 + 2}};
    fe b{{seed * 11 + t, seed * 13 + 5, seed * 17 + t, seed * 19 + 3}};
    for (int i = 0; i < iters; i++) {
        if (OP == 0) { a = fe_mul_c32(a, b); b = fe_mul_c32(b, a); }
        else if (OP == 1) { a = fe_mul_c64(a, b); b = fe_mul_c64(b, a); }
        else if (OP == 2) { a = fe_sqr_c64(a); b = fe_sqr_c64(b); }
        else if (OP == 3) { a = fe_sub(a, b); b = fe_sub(b, a); }
        else if (OP == 4) { a = fe_inv(a); b = fe_sub(b, a); }
        else if (OP == 5) { a = fe_inv_fermat(a); b = fe_sub(b, a); }
    }
    out[t] = a.v[0] ^ a.v[1] ^ a.v[2] ^ a.v[3] ^ b.v[0] ^ b.v[3];
}

template <typename F>
static double time_ms(F launch, int reps = 5) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < reps; r++) {
        CK(hipEventRecord(e0, 0));
        launch();
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double ghz = p.clockRate / 1e6;
    printf("device %s arch %s CUs %d clock %.2f GHz\n", p.name, p.gcnArchName, cus, ghz);
    uint64_t *out; CK(hipMalloc(&out, (size_t)cus * 8 * 256 * 8 * 8));
    const char *names[] = {"v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "add_u64", "v_fma_f64", "add/xor_u32", "v_mul_u32_u24"};
    const int per_iter[] = {64, 64, 64, 64, 64, 64, 64};
    printf("\n== instruction throughput (ops counted per lane-instruction; wave64) ==\n");
    for (int wps = 1; wps <= 4; wps *= 2) { // waves per SIMD
        const int blocks = cus * wps, threads = 256, iters = 2000;
        for (int op = 0; op < 7; op++) {
            double ms;
            switch (op) {
            case 0: ms = time_ms([&] { hipLaunchKernelGGL(k_instr<0>, dim3(blocks), dim3(threads), 0, 0, out, 1u, iters); }); break;
            case 1: ms = time_ms([&] { hipLaunchKernelGGL(k_instr<1>, dim3(blocks), dim3(threads), 0, 0, out, 1u, iters); }); break;
            case 2: ms = time_ms([&] { hipLaunchKernelGGL(k_instr<2>, dim3(blocks), dim3(threads), 0, 0, out, 1u, iters); }); break;
            case 3: ms = time_ms([&] { hipLaunchKernelGGL(k_instr<3>, dim3(blocks), dim3(threads), 0, 0, out, 1u, iters); }); break;
            case 4: ms = time_ms([&] { hipLaunchKernelGGL(k_instr<4>, dim3(blocks), dim3(threads), 0, 0, out, 1u, iters); }); break;
            case 5: ms = time_ms([&] { hipLaunchKernelGGL(k_instr<5>, dim3(blocks), dim3(threads), 0, 0, out, 1u, iters); }); break;
            default: ms = time_ms([&] { hipLaunchKernelGGL(k_instr<6>, dim3(blocks), dim3(threads), 0, 0, out, 1u, iters); }); break;
            }
            // wave-instructions per SIMD = wps * iters * per_iter ; cycles per SIMD = ms*1e-3*clk
            const double winstr = (double)wps * iters * per_iter[op];
            const double cyc = ms * 1e-3 * ghz * 1e9;
            printf("  %-14s waves/SIMD %d : %8.3f ms  -> %6.2f cycles per wave-instruction (nominal clock)\n", names[op], wps, ms, cyc / winstr);
        }
    }
    printf("\n== 256-bit field ops, chip-wide (2 ops per iteration per lane) ==\n");
    const char *fnames[] = {"fe_mul_c32(asm comba)", "fe_mul_c64(compiler)", "fe_sqr_c64(compiler)", "fe_sub", "fe_inv(safegcd30)", "fe_inv_fermat"};
    for (int wps = 1; wps <= 4; wps *= 2) {
        const int blocks = cus * wps * 4, threads = 64;
        for (int op = 0; op < 6; op++) {
            const int iters = op >= 4 ? 8 : 2000;
            double ms;
            switch (op) {
            case 0: ms = time_ms([&] { hipLaunchKernelGGL(k_field<0>, dim3(blocks), dim3(threads), 0, 0, out, 12345ull, iters); }); break;
            case 1: ms = time_ms([&] { hipLaunchKernelGGL(k_field<1>, dim3(blocks), dim3(threads), 0, 0, out, 12345ull, iters); }); break;
            case 2: ms = time_ms([&] { hipLaunchKernelGGL(k_field<2>, dim3(blocks), dim3(threads), 0, 0, out, 12345ull, iters); }); break;
            case 3: ms = time_ms([&] { hipLaunchKernelGGL(k_field<3>, dim3(blocks), dim3(threads), 0, 0, out, 12345ull, iters); }); break;
            case 4: ms = time_ms([&] { hipLaunchKernelGGL(k_field<4>, dim3(blocks), dim3(threads), 0, 0, out, 12345ull, iters); }); break;
            default: ms = time_ms([&] { hipLaunchKernelGGL(k_field<5>, dim3(blocks), dim3(threads), 0, 0, out, 12345ull, iters); }); break;
            }
            const double ops = (double)blocks * threads * iters * (op >= 4 ? 1 : 2);
            const double cyc_per_wave_op = ms * 1e-3 * ghz * 1e9 / ((double)wps * iters * (op >= 4 ? 1 : 2));
            printf("  %-22s waves/SIMD %d : %8.3f ms  %9.2f Gop/s chip  %8.1f cycles per wave-op per SIMD\n", fnames[op], wps, ms, ops / ms / 1e6, cyc_per_wave_op);
        }
    }
    return 0;
}
