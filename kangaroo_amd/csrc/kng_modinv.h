// kng_modinv.h -- modular inverse mod p for gfx950 (device code).
//
// Replaces GPU/GPUMath.h:700-803 (_ModInv, delayed-right-shift-62 binary GCD).  The contract
// is only "canonical inverse in [0,p), inverse of 0 is 0" (GPUMath.h:795-801 ==
// SECPK1/IntMod.cpp:560-565), so any algorithm is bit-identical.
//
// In SIMT one inversion costs a wave the same whether 1 or 64 lanes need it, and a
// data-dependent GCD diverges across the 64 lanes of a wave64.  The fixed-flow Fermat ladder
// a^(p-2) (255 squarings + 15 multiplications, no divergence, no extra registers beyond the
// multiplier's) is used; its cost is amortised over the per-lane Montgomery batch.
#pragma once

#include "kng_field.h"

namespace kng {

__device__ __noinline__ fe fe_sqr_n(fe a, int n) {
#pragma unroll 1
    for (int i = 0; i < n; i++) a = fe_sqr(a);
    return a;
}

__device__ __noinline__ fe fe_mul_noinline(const fe &a, const fe &b) { return fe_mul(a, b); }

// p-2 = 2^256 - 2^32 - 979: 223 ones, 0, 22 ones, 0000, 1, 0, 11, 0, 1
__device__ __noinline__ fe fe_inv(const fe &a_in) {
    const fe a = fe_canon(a_in);
    fe x2 = fe_mul_noinline(fe_sqr_n(a, 1), a);
    fe x3 = fe_mul_noinline(fe_sqr_n(x2, 1), a);
    fe x6 = fe_mul_noinline(fe_sqr_n(x3, 3), x3);
    fe x9 = fe_mul_noinline(fe_sqr_n(x6, 3), x3);
    fe x11 = fe_mul_noinline(fe_sqr_n(x9, 2), x2);
    fe x22 = fe_mul_noinline(fe_sqr_n(x11, 11), x11);
    fe x44 = fe_mul_noinline(fe_sqr_n(x22, 22), x22);
    fe x88 = fe_mul_noinline(fe_sqr_n(x44, 44), x44);
    fe x176 = fe_mul_noinline(fe_sqr_n(x88, 88), x88);
    fe x220 = fe_mul_noinline(fe_sqr_n(x176, 44), x44);
    fe x223 = fe_mul_noinline(fe_sqr_n(x220, 3), x3);
    fe t = fe_mul_noinline(fe_sqr_n(x223, 23), x22);
    t = fe_mul_noinline(fe_sqr_n(t, 5), a);
    t = fe_mul_noinline(fe_sqr_n(t, 3), x2);
    t = fe_mul_noinline(fe_sqr_n(t, 2), a);
    // a == 0 gives 0 (0^(p-2) = 0): same as the reference (inverse of 0 is 0)
    return fe_canon(t);
}

} // namespace kng
