// fe_extras.h -- alternatives to the product arithmetic that measurements ruled out, kept for the tools only
// (tools/ubench.hip compares them, tools/host_field_test.cpp cross-checks the product code against them):
//   fe_mul_c64 / fe_sqr_c64 : 64-bit limbs, everything left to the compiler (250-460 instructions per product)
//   fe_inv_fermat           : a^(p-2) ladder, 255 squarings + 15 multiplications (~3.4x the cost of safegcd)
#pragma once
#include "../kangaroo_amd/csrc/kng_field.h"
#include "../kangaroo_amd/csrc/kng_modinv.h"

namespace kng {

// 512 -> 320 -> 256 fold (GPUMath.h:840-856 / IntMod.cpp:926-942)
KNG_DEV fe fe_fold(const uint64_t w[8]) {
    // t[0..4] = w[4..7] * K1C
    uint64_t t[5];
    u128 c = (u128)w[4] * K1C;
    t[0] = (uint64_t)c;
    c = (c >> 64) + (u128)w[5] * K1C;
    t[1] = (uint64_t)c;
    c = (c >> 64) + (u128)w[6] * K1C;
    t[2] = (uint64_t)c;
    c = (c >> 64) + (u128)w[7] * K1C;
    t[3] = (uint64_t)c;
    t[4] = (uint64_t)(c >> 64);
    fe r;
    unsigned long long cy = 0;
    r.v[0] = __builtin_addcll(w[0], t[0], 0, &cy);
    r.v[1] = __builtin_addcll(w[1], t[1], cy, &cy);
    r.v[2] = __builtin_addcll(w[2], t[2], cy, &cy);
    r.v[3] = __builtin_addcll(w[3], t[3], cy, &cy);
    // second fold: (t[4] + carry) * K1C, t[4]+carry <= K1C so no overflow
    const u128 f = (u128)(t[4] + cy) * K1C;
    r.v[0] = __builtin_addcll(r.v[0], (uint64_t)f, 0, &cy);
    r.v[1] = __builtin_addcll(r.v[1], (uint64_t)(f >> 64), cy, &cy);
    r.v[2] = __builtin_addcll(r.v[2], 0, cy, &cy);
    r.v[3] = __builtin_addcll(r.v[3], 0, cy, &cy);
    // final carry dropped on purpose: identical to the reference (IntMod.cpp:944)
    return r;
}

KNG_DEV fe fe_mul_c64(const fe &a, const fe &b) {
    uint64_t w[8];
    // row 0
    u128 c = (u128)a.v[0] * b.v[0];
    w[0] = (uint64_t)c;
    c = (c >> 64) + (u128)a.v[1] * b.v[0];
    w[1] = (uint64_t)c;
    c = (c >> 64) + (u128)a.v[2] * b.v[0];
    w[2] = (uint64_t)c;
    c = (c >> 64) + (u128)a.v[3] * b.v[0];
    w[3] = (uint64_t)c;
    w[4] = (uint64_t)(c >> 64);
#pragma unroll
    for (int i = 1; i < 4; i++) {
        c = (u128)a.v[0] * b.v[i] + w[i];
        w[i] = (uint64_t)c;
        c = (c >> 64) + (u128)a.v[1] * b.v[i] + w[i + 1];
        w[i + 1] = (uint64_t)c;
        c = (c >> 64) + (u128)a.v[2] * b.v[i] + w[i + 2];
        w[i + 2] = (uint64_t)c;
        c = (c >> 64) + (u128)a.v[3] * b.v[i] + w[i + 3];
        w[i + 3] = (uint64_t)c;
        w[i + 4] = (uint64_t)(c >> 64);
    }
    return fe_fold(w);
}

KNG_DEV fe fe_sqr_c64(const fe &a) {
    // 10 distinct products: 4 squares + 6 cross terms added twice (GPUMath.h:913-1019 idea;
    // the result is the same 512-bit integer as a*a, so the fold is bit-identical)
    uint64_t w[8];
    // cross terms: sum_{i<j} a_i a_j 2^(64(i+j))
    u128 c = (u128)a.v[0] * a.v[1];
    uint64_t x1 = (uint64_t)c;
    c = (c >> 64) + (u128)a.v[0] * a.v[2];
    uint64_t x2 = (uint64_t)c;
    c = (c >> 64) + (u128)a.v[0] * a.v[3];
    uint64_t x3 = (uint64_t)c;
    uint64_t x4 = (uint64_t)(c >> 64);
    c = (u128)a.v[1] * a.v[2] + x3;
    x3 = (uint64_t)c;
    c = (c >> 64) + (u128)a.v[1] * a.v[3] + x4;
    x4 = (uint64_t)c;
    uint64_t x5 = (uint64_t)(c >> 64);
    c = (u128)a.v[2] * a.v[3] + x5;
    x5 = (uint64_t)c;
    uint64_t x6 = (uint64_t)(c >> 64);
    // double
    uint64_t x7 = x6 >> 63;
    x6 = (x6 << 1) | (x5 >> 63);
    x5 = (x5 << 1) | (x4 >> 63);
    x4 = (x4 << 1) | (x3 >> 63);
    x3 = (x3 << 1) | (x2 >> 63);
    x2 = (x2 << 1) | (x1 >> 63);
    x1 = x1 << 1;
    // add squares
    c = (u128)a.v[0] * a.v[0];
    w[0] = (uint64_t)c;
    c = (c >> 64) + x1;
    w[1] = (uint64_t)c;
    c = (c >> 64) + (u128)a.v[1] * a.v[1] + x2;
    w[2] = (uint64_t)c;
    c = (c >> 64) + x3;
    w[3] = (uint64_t)c;
    c = (c >> 64) + (u128)a.v[2] * a.v[2] + x4;
    w[4] = (uint64_t)c;
    c = (c >> 64) + x5;
    w[5] = (uint64_t)c;
    c = (c >> 64) + (u128)a.v[3] * a.v[3] + x6;
    w[6] = (uint64_t)c;
    w[7] = (uint64_t)(c >> 64) + x7;
    return fe_fold(w);
}


// ---- Fermat ladder a^(p-2): 255 squarings + 15 multiplications (cross-check only) ----
KNG_DEV_NOINLINE fe fe_sqr_n(fe a, int n) {
#pragma unroll 1
    for (int i = 0; i < n; i++) a = fe_sqr(a);
    return a;
}
KNG_DEV_NOINLINE fe fe_mul_noinline(const fe &a, const fe &b) { return fe_mul(a, b); }

// p-2 = 2^256 - 2^32 - 979: 223 ones, 0, 22 ones, 0000, 1, 0, 11, 0, 1
KNG_DEV_NOINLINE fe fe_inv_fermat(const fe &a_in) {
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
    return fe_canon(t);
}

} // namespace kng
