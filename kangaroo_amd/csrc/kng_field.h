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

#include <stdint.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KNG_DEV __device__ __forceinline__
#define KNG_DEV_NOINLINE __device__ __noinline__
#else // host build of the same arithmetic (tools/host_field_test.cpp, clang++): unit tests without a GPU
#define KNG_DEV static inline
#define KNG_DEV_NOINLINE static
#endif
// marks a block that is exact but (nearly) never executed: the volatile asm keeps hipcc from
// if-converting the branch back into straight-line selects
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define KNG_RARE_PATH() asm volatile("; rare path")
#else
#define KNG_RARE_PATH() ((void)0)
#endif

namespace kng {

typedef unsigned __int128 u128;

struct fe {
    uint64_t v[4];
};

constexpr uint64_t P0 = 0xFFFFFFFEFFFFFC2FULL; // GPUMath.h:83-88
constexpr uint64_t PX = 0xFFFFFFFFFFFFFFFFULL;
constexpr uint64_t K1C = 0x1000003D1ULL; // 2^256 mod p

KNG_DEV fe fe_zero() { return fe{{0, 0, 0, 0}}; }
KNG_DEV fe fe_one() { return fe{{1, 0, 0, 0}}; }
KNG_DEV bool fe_is_zero(const fe &a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }

// r = a - b ; if borrow r += p        (GPUMath.h:476-494)
// On 32-bit words so that hipcc emits plain v_sub_co/v_subb_co chains (with 64-bit builtins it
// falls back to 64-bit compares and selects: ~50 instructions per subtraction).  Adding p is
// subtracting 2^256 - p = 2^32 + 977 modulo 2^256.
KNG_DEV fe fe_sub(const fe &a, const fe &b) {
    uint32_t r[8];
    unsigned br = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r[2 * i] = __builtin_subc((unsigned)a.v[i], (unsigned)b.v[i], br, &br);
        r[2 * i + 1] = __builtin_subc((unsigned)(a.v[i] >> 32), (unsigned)(b.v[i] >> 32), br, &br);
    }
    const unsigned m = 0u - br; // all ones on borrow
    unsigned c = 0;
    // "x - 0 - borrow" with a literal 0 makes hipcc materialise the borrow as 0/1 (v_cndmask) and
    // then v_sub_co it: two instructions + a hazard nop per limb.  An opaque zero in a VGPR keeps
    // the chain on v_subb_co_u32.
    unsigned zero = 0;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(zero));
#endif
    r[0] = __builtin_subc(r[0], 977u & m, 0u, &c);
    r[1] = __builtin_subc(r[1], 1u & m, c, &c);
    // the borrow out of the low 64 bits needs (a-b) mod 2^64 < 2^32+977: about one subtraction in
    // 2^32.  Rippling it through limbs 2..7 stays exact but sits behind a (practically never taken)
    // branch instead of costing 6 dependent v_subb_co_u32 every time.
    if (__builtin_expect(c != 0, 0)) {
        KNG_RARE_PATH();
#pragma unroll
        for (int i = 2; i < 8; i++) r[i] = __builtin_subc(r[i], zero, c, &c);
    }
    return fe{{(uint64_t)r[0] | ((uint64_t)r[1] << 32), (uint64_t)r[2] | ((uint64_t)r[3] << 32),
               (uint64_t)r[4] | ((uint64_t)r[5] << 32), (uint64_t)r[6] | ((uint64_t)r[7] << 32)}};
}

} // namespace kng

#include "kng_mul32.h"
#ifndef KNG_USE_MULASM
#define KNG_USE_MULASM 1 // product + fold as one scheduled asm statement (kng_mulasm.h); 0 = the per-column form of rounds 1-2
#endif
#include "kng_mulasm.h"

namespace kng {

#define KNG_ADDC32(x, y, ci, co) __builtin_addc((unsigned)(x), (unsigned)(y), (unsigned)(ci), (co))

// 512 -> 256 fold on 32-bit words (GPUMath.h:840-856 / IntMod.cpp:926-942): S = lo + hi*K exactly,
// T = S >> 256 (<= K), r = (S mod 2^256) + T*K mod 2^256 -- the integer the reference computes (last carry
// dropped).  General form, every carry rippled: the rarely-taken twin of fe_fold32 below.
KNG_DEV fe fe_fold32_full(const uint32_t w[16]) {
    unsigned zero = 0; // opaque zero: see fe_sub
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(zero));
#endif
    uint64_t q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = (uint64_t)w[8 + j] * 977u + w[j]; // < 2^43
    uint32_t t[9];
    unsigned c = 0;
    // S = sum q_j 2^(32j) + sum hi_j 2^(32(j+1)) : two carry chains
    t[0] = (uint32_t)q[0];
#pragma unroll
    for (int j = 1; j < 8; j++) t[j] = KNG_ADDC32((uint32_t)q[j], w[8 + j - 1], c, &c);
    t[8] = KNG_ADDC32(zero, w[15], c, &c);
    uint32_t top = c;
    c = 0;
#pragma unroll
    for (int j = 1; j < 9; j++) t[j] = KNG_ADDC32(t[j], (uint32_t)(q[j - 1] >> 32), c, &c);
    top += c; // T = top*2^32 + t[8] <= 2^32 + 977
    // T*K = t8*977 + (t8 << 32) + top*977*2^32 + top*2^64
    const uint64_t f = (uint64_t)t[8] * 977u;
    uint32_t a1 = KNG_ADDC32((uint32_t)(f >> 32), t[8], 0, &c);
    uint32_t a2 = c;
    a1 = KNG_ADDC32(a1, top * 977u, 0, &c);
    a2 += c + top;
    uint32_t r[8];
    c = 0;
    r[0] = KNG_ADDC32(t[0], (uint32_t)f, c, &c);
    r[1] = KNG_ADDC32(t[1], a1, c, &c);
    r[2] = KNG_ADDC32(t[2], a2, c, &c);
#pragma unroll
    for (int j = 3; j < 8; j++) r[j] = t[j];
    // a2 <= 3: the carry out of limb 2 needs t[2] >= 2^32 - 4 (see fe_sub: exact, rarely taken)
    if (__builtin_expect(c != 0, 0)) {
        KNG_RARE_PATH();
#pragma unroll
        for (int j = 3; j < 8; j++) r[j] = KNG_ADDC32(r[j], zero, c, &c);
    }
    // final carry dropped on purpose (IntMod.cpp:944)
    return fe{{(uint64_t)r[0] | ((uint64_t)r[1] << 32), (uint64_t)r[2] | ((uint64_t)r[3] << 32),
               (uint64_t)r[4] | ((uint64_t)r[5] << 32), (uint64_t)r[6] | ((uint64_t)r[7] << 32)}};
}


// d = a*b + c (32x32+64 -> 64), carry-out of the 64-bit sum in `co` (an SGPR lane mask on the device: the
// conditions below are therefore wave-uniform "some lane overflowed" tests that cost no VALU instruction)
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#if !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
#error "KNG_MAD64: the carry-out is a 64-bit SGPR lane mask and vdst carries no early-clobber -- valid for wave64 gfx9 only"
#endif
#define KNG_MAD64(d, co, a, b, c) asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(co) : "v"(a), "s"(b), "v"(c))
#else
#define KNG_MAD64(d, co, a, b, c)                                                                          \
    do {                                                                                                   \
        const uint64_t p_ = (uint64_t)(a) * (b), c_ = (c);                                                 \
        d = p_ + c_;                                                                                       \
        co = (uint64_t)(d < p_);                                                                           \
    } while (0)
#endif

// The same fold, arranged so that the common case needs ONE carry chain instead of three.
// hi*K = hi*977 + (hi << 32).  A v_mad_u64_u32 adds a 64-bit value for free, and the eight products hi_j*977
// (< 2^42) at even j do not overlap each other, nor do those at odd j:
//     E_j = hi_j*977 + (lo_j | lo_j+1 << 32)     j = 0,2,4,6   -> the number  lo + sum_even hi_j*977*2^(32j)
//     O_j = hi_j*977 + (hi_j-1 | hi_j << 32)     j = 1,3,5,7   -> (hi << 32) + sum_odd hi_j*977*2^(32j)
// so S = E + O with E at limbs 0..7 and O at limbs 1..8: one 8-link chain.  A MAD overflows 64 bits only when its
// addend is within 2^42 of 2^64 (2^-22 per MAD); T = S >> 256 exceeds 32 bits only when hi_7 is within 978 of 2^32;
// the second fold's carry leaves limb 2 only when that limb is all ones.  Any of these is reported to the caller
// (`rare`: wave-uniform lane mask of MAD overflows, ORed in; `lane`: this lane's two chain carries), who then folds
// again on the exact path (fe_fold32_checked); tests/test_gpu_parity.py::test_fold_rare_paths and the host-compiled
// header test hit each condition.  14 carry instructions fewer per product; measured +0.4 % on the walk
// (profiles/r02_ab_single_chain_fold.txt) -- the eight lane-mask ORs it adds cost nearly as much as the carries.
KNG_DEV fe fe_fold32(const uint32_t w[16], uint64_t &rare, unsigned &lane) {
    unsigned zero = 0; // opaque zero: see fe_sub
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(zero));
#endif
    const uint32_t k977 = 977u;
    uint64_t e[4], o[4], ce[4], co[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const uint64_t lp = (uint64_t)w[2 * j] | ((uint64_t)w[2 * j + 1] << 32);
        const uint64_t hp = (uint64_t)w[8 + 2 * j] | ((uint64_t)w[9 + 2 * j] << 32);
        KNG_MAD64(e[j], ce[j], w[8 + 2 * j], k977, lp);
        KNG_MAD64(o[j], co[j], w[9 + 2 * j], k977, hp);
    }
    uint32_t s[9];
    unsigned c = 0;
    s[0] = (uint32_t)e[0];
    s[1] = KNG_ADDC32((uint32_t)(e[0] >> 32), (uint32_t)o[0], c, &c);
#pragma unroll
    for (int j = 1; j < 4; j++) {
        s[2 * j] = KNG_ADDC32((uint32_t)e[j], (uint32_t)(o[j - 1] >> 32), c, &c);
        s[2 * j + 1] = KNG_ADDC32((uint32_t)(e[j] >> 32), (uint32_t)o[j], c, &c);
    }
    s[8] = KNG_ADDC32((uint32_t)(o[3] >> 32), zero, c, &c);
    const unsigned top = c; // bit 32 of T
    // second fold, T = s[8] < 2^32:  T*K = s8*977 + (s8 << 32)
    uint64_t r01, c1;
    KNG_MAD64(r01, c1, s[8], k977, (uint64_t)s[0] | ((uint64_t)s[1] << 32));
    const uint32_t r1 = KNG_ADDC32((uint32_t)(r01 >> 32), s[8], 0u, &c);
    const uint32_t r2 = KNG_ADDC32(s[2], zero, c, &c);
    rare |= ce[0] | ce[1] | ce[2] | ce[3] | co[0] | co[1] | co[2] | co[3] | c1;
    lane = top | c;
    return fe{{(uint64_t)(uint32_t)r01 | ((uint64_t)r1 << 32), (uint64_t)r2 | ((uint64_t)s[3] << 32),
               (uint64_t)s[4] | ((uint64_t)s[5] << 32), (uint64_t)s[6] | ((uint64_t)s[7] << 32)}};
}

KNG_DEV void fe_to32(uint32_t r[8], const fe &a) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
        r[2 * i] = (uint32_t)a.v[i];
        r[2 * i + 1] = (uint32_t)(a.v[i] >> 32);
    }
}

KNG_DEV fe fe_from32(const uint32_t r[8]) {
    return fe{{(uint64_t)r[0] | ((uint64_t)r[1] << 32), (uint64_t)r[2] | ((uint64_t)r[3] << 32),
               (uint64_t)r[4] | ((uint64_t)r[5] << 32), (uint64_t)r[6] | ((uint64_t)r[7] << 32)}};
}

// Same integer as GPUMath.h:810-858 on every input: the single-chain fold first; when one of its "this cannot be
// right" conditions holds in some lane (about once in 2^13 wave-products on random operands) the wave folds again
// with every carry rippled.
KNG_DEV fe fe_fold32_checked(const uint32_t w[16]) {
    uint64_t rare = 0;
    unsigned lane;
    fe r = fe_fold32(w, rare, lane);
    if (__builtin_expect((rare != 0) | (lane != 0), 0)) {
        KNG_RARE_PATH();
        r = fe_fold32_full(w);
    }
    return r;
}

KNG_DEV fe fe_mul_c32(const fe &a, const fe &b) {
    uint32_t x[8], y[8], w[16];
    fe_to32(x, a);
    fe_to32(y, b);
#if KNG_MULASM && KNG_USE_MULASM
    // one scheduled statement for product and fold; `rare` collects the lanes for which the short fold is not exact
    // (wave-uniform test: an SGPR lane mask), in which case the wave multiplies again with every carry rippled
    uint32_t r[8];
    uint64_t rare = 0;
    mul_fold_asm(r, x, y, rare);
    if (__builtin_expect(rare != 0, 0)) {
        KNG_RARE_PATH();
        mul_wide32(w, x, y);
        return fe_fold32_full(w);
    }
    return fe{{(uint64_t)r[0] | ((uint64_t)r[1] << 32), (uint64_t)r[2] | ((uint64_t)r[3] << 32),
               (uint64_t)r[4] | ((uint64_t)r[5] << 32), (uint64_t)r[6] | ((uint64_t)r[7] << 32)}};
#else
    mul_wide32(w, x, y);
    return fe_fold32_checked(w);
#endif
}

// a^2 with the 36-MAD squaring schedule (sqr_wide32) -- same 512-bit integer, same fold
KNG_DEV fe fe_sqr_c32(const fe &a) {
    uint32_t x[8], w[16];
    fe_to32(x, a);
    sqr_wide32(w, x);
    return fe_fold32_checked(w);
}

KNG_DEV fe fe_mul(const fe &a, const fe &b) { return fe_mul_c32(a, b); }
KNG_DEV fe fe_sqr(const fe &a) { return fe_sqr_c32(a); }

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
