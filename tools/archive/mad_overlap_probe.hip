// mad_overlap_probe.hip -- does v_mad_u64_u32 on gfx950 tolerate vdst overlapping src0/src1?
// (measurement tool; answer recorded in profiles/ and DESIGN.md)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// variant: which source the low half of the destination pair overlaps
template <int V>
__global__ void k(unsigned *out, const unsigned *in) {
    const int t = threadIdx.x;
    unsigned a = in[4 * t], b = in[4 * t + 1], c0 = in[4 * t + 2], c1 = in[4 * t + 3], lo, hi;
    if (V == 0) // no overlap
        asm volatile("v_mov_b32 v20, %2\n\tv_mov_b32 v21, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\ts_nop 4\n\t"
                     "v_mad_u64_u32 v[8:9], vcc, v20, v21, v[12:13]\n\ts_nop 4\n\tv_mov_b32 %0, v8\n\tv_mov_b32 %1, v9"
                     : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b), "v"(c0), "v"(c1) : "vcc", "v8", "v9", "v12", "v13", "v20", "v21");
    if (V == 1) // dst.lo == src0
        asm volatile("v_mov_b32 v8, %2\n\tv_mov_b32 v21, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\ts_nop 4\n\t"
                     "v_mad_u64_u32 v[8:9], vcc, v8, v21, v[12:13]\n\ts_nop 4\n\tv_mov_b32 %0, v8\n\tv_mov_b32 %1, v9"
                     : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b), "v"(c0), "v"(c1) : "vcc", "v8", "v9", "v12", "v13", "v20", "v21");
    if (V == 2) // dst.lo == src0 == src1 (a squaring term: the case hipcc emitted)
        asm volatile("v_mov_b32 v8, %2\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\ts_nop 4\n\t"
                     "v_mad_u64_u32 v[8:9], vcc, v8, v8, v[12:13]\n\ts_nop 4\n\tv_mov_b32 %0, v8\n\tv_mov_b32 %1, v9"
                     : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b), "v"(c0), "v"(c1) : "vcc", "v8", "v9", "v12", "v13", "v20", "v21");
    if (V == 3) // dst.hi == src1
        asm volatile("v_mov_b32 v20, %2\n\tv_mov_b32 v9, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\ts_nop 4\n\t"
                     "v_mad_u64_u32 v[8:9], vcc, v20, v9, v[12:13]\n\ts_nop 4\n\tv_mov_b32 %0, v8\n\tv_mov_b32 %1, v9"
                     : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b), "v"(c0), "v"(c1) : "vcc", "v8", "v9", "v12", "v13", "v20", "v21");
    if (V == 4) // sdst pair contains the SGPR used as src1
        asm volatile("v_mov_b32 v20, %2\n\tv_readfirstlane_b32 s20, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\ts_nop 4\n\t"
                     "v_mad_u64_u32 v[8:9], s[20:21], v20, s20, v[12:13]\n\ts_nop 4\n\tv_mov_b32 %0, v8\n\tv_mov_b32 %1, v9"
                     : "=&v"(lo), "=&v"(hi) : "v"(a), "v"(b), "v"(c0), "v"(c1) : "vcc", "v8", "v9", "v12", "v13", "v20", "v21", "s20", "s21");
    out[2 * t] = lo;
    out[2 * t + 1] = hi;
}
int main() {
    const int n = 64;
    unsigned h[4 * n], *din, *dout, res[2 * n];
    srand(7);
    for (int i = 0; i < 4 * n; i++) h[i] = ((unsigned)rand() << 16) ^ (unsigned)rand();
    for (int i = 0; i < n; i++) h[4 * i + 1] = h[1]; // uniform b (needed for the SGPR variant)
    CK(hipMalloc(&din, sizeof h)); CK(hipMalloc(&dout, sizeof res));
    CK(hipMemcpy(din, h, sizeof h, hipMemcpyHostToDevice));
    const char *names[] = {"no overlap", "vdst.lo == src0", "vdst.lo == src0 == src1", "vdst.hi == src1", "sdst pair contains SGPR src1"};
    for (int v = 0; v < 5; v++) {
        switch (v) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(n), 0, 0, dout, din); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(n), 0, 0, dout, din); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(n), 0, 0, dout, din); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(n), 0, 0, dout, din); break;
        default: hipLaunchKernelGGL(k<4>, dim3(1), dim3(n), 0, 0, dout, din); break;
        }
        CK(hipMemcpy(res, dout, sizeof res, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int i = 0; i < n; i++) {
            unsigned long long a = h[4 * i], b = (v == 2) ? a : h[4 * i + 1], c = h[4 * i + 2] | ((unsigned long long)h[4 * i + 3] << 32);
            unsigned long long want = a * b + c, got = res[2 * i] | ((unsigned long long)res[2 * i + 1] << 32);
            if (want != got) { if (!bad) printf("   first mismatch: a=%llx b=%llx c=%llx want=%llx got=%llx\n", a, b, c, want, got); bad++; }
        }
        printf("%-32s : %d / %d wrong\n", names[v], bad, n);
    }
    return 0;
}
