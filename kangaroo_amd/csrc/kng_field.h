// kng_field.h -- 256-bit arithmetic mod p = 2^256 - 0x1000003D1 for gfx950 (device code).
//
// Written from scratch for CDNA4: the wave64 VALU has no 64-bit multiplier, so every
// 64x64->128 product below lowers to v_mad_u64_u32 chains (32x32+64->64 MAD, carry-out in an
// SGPR pair); additions lower to v_add_co/v_addc_co chains.  Semantics are those of the
// reference's device math so that results are bit-identical to its CPU path:
//   fe_mul / fe_sqr : GPU/GPUMath.h:810-858, :909-1019 == SECPK1/IntMod.cpp:873-950
//                     (512-bit product, fold hi*0x1000003D1 twice, last carry dropped,
//                      NO comparison with p -> result in [0,2^256))
//   fe_sub          : GPU/GPUMath.h:476-494 == SECPK1/IntMod.cpp:95-99 (a-b, +p on borrow)
//   fe_inv          : GPU/GPUMath.h:700-803 == SECPK1/IntMod.cpp:368-569 (canonical inverse
//                     in [0,p), inverse of 0 is 0).  Any algorithm returning the canonical
//                     inverse is bit-identical; see kng_modinv.h for the one used.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kng {

typedef unsigned __int128 u128;

struct fe {
    uint64_t v[4];
};

#define KNG_DEV __device__ __forceinline__

constexpr uint64_t P0 = 0xFFFFFFFEFFFFFC2FULL; // GPUMath.h:83-88
constexpr uint64_t PX = 0xFFFFFFFFFFFFFFFFULL;
constexpr uint64_t K1C = 0x1000003D1ULL; // 2^256 mod p

KNG_DEV fe fe_zero() { return fe{{0, 0, 0, 0}}; }
KNG_DEV fe fe_one() { return fe{{1, 0, 0, 0}}; }
KNG_DEV bool fe_is_zero(const fe &a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }

// r = a - b ; if borrow r += p        (GPUMath.h:476-494)
KNG_DEV fe fe_sub(const fe &a, const fe &b) {
    fe r;
    unsigned long long br = 0, c = 0;
    r.v[0] = __builtin_subcll(a.v[0], b.v[0], 0, &br);
    r.v[1] = __builtin_subcll(a.v[1], b.v[1], br, &br);
    r.v[2] = __builtin_subcll(a.v[2], b.v[2], br, &br);
    r.v[3] = __builtin_subcll(a.v[3], b.v[3], br, &br);
    const uint64_t m = 0 - (uint64_t)br; // all ones on borrow
    r.v[0] = __builtin_addcll(r.v[0], P0 & m, 0, &c);
    r.v[1] = __builtin_addcll(r.v[1], m, c, &c);
    r.v[2] = __builtin_addcll(r.v[2], m, c, &c);
    r.v[3] = __builtin_addcll(r.v[3], m, c, &c);
    return r;
}

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

KNG_DEV fe fe_mul(const fe &a, const fe &b) {
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

KNG_DEV fe fe_sqr(const fe &a) {
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

// full reduction of a value in [0,2^256) into [0,p)
KNG_DEV fe fe_canon(const fe &a) {
    // a >= p  <=>  a + K1C overflows 2^256
    fe r;
    unsigned long long c = 0;
    r.v[0] = __builtin_addcll(a.v[0], K1C, 0, &c);
    r.v[1] = __builtin_addcll(a.v[1], 0, c, &c);
    r.v[2] = __builtin_addcll(a.v[2], 0, c, &c);
    r.v[3] = __builtin_addcll(a.v[3], 0, c, &c);
    return c ? r : a;
}

} // namespace kng
