// kng_field29.h -- carry-free arithmetic mod p = 2^256 - 0x1000003D1 for gfx950 (device code).
//
// WHY: on CDNA4 every carry-propagating instruction (v_add_co/v_addc_co) and every 64-bit or VOP3
// op costs as much as a v_mad_u64_u32 (~1.6x a plain v_add_u32; profiles/r01_instr_throughput_gfx950.txt).
// With saturated 32-bit limbs a modular multiplication is 64 MADs + ~160 carry/move instructions
// and a subtraction is 17 carry instructions.  Here a field element is NINE UNSIGNED 29-BIT LIMBS
//      value = sum l[i] * 2^(29 i),     i = 0..8   (261 bits of room for a 256-bit residue)
// and limbs are allowed to run "lazy" (up to 2^31) between operations, so that
//   * a product column is at most 9 * 2^31 * 2^29.x < 2^64: 81 v_mad_u64_u32 accumulate with NO
//     carry instruction at all, plus 18 MADs that fold the high half (2^261 = 2^37 + 0x7A20 mod p),
//   * a subtraction a - b is nine full-rate v_add_u32 of a with (multiple of p) - b, where the
//     multiple of p is chosen limb-wise large enough that no limb ever goes negative
//     (for the jump table the biased negation 2p - J is precomputed by the host),
//   * only the new x is brought to canonical form each jump (its low 5 bits select the jump, its
//     top bits are the distinguished-point test, GPUCompute.h:69,96); y and the running products
//     stay "almost reduced" (< 2^256 + epsilon, limbs < 2^29) in HBM.
// Values are always exact residues mod p; canonical outputs are therefore identical to the
// reference's SECPK1 results whenever those are canonical (always, up to the reference's own
// 2^-220 "very very unlikely" sliver, IntMod.cpp:944).  The reference-exact lazy-fold arithmetic
// (kng_field.h) stays in the library for the primitive parity tests and as walk policy "32".
#pragma once

#include "kng_field.h"

namespace kng {

constexpr uint32_t M29 = 0x1FFFFFFFu;
constexpr uint32_t R0_29 = 0x7A20u; // 2^261 mod p = 2^37 + 0x7A20  -> h*R0 at limb k, h<<8 at limb k+1

struct fe29 {
    uint32_t l[9];
};

// limbs of p, 2p and 4p (limb-wise multiples: every limb of 2p/4p dominates a normalised limb)
KNG_DEV uint32_t p29(int i) { return i == 0 ? 0x1FFFFC2Fu : i == 1 ? 0x1FFFFFF7u : i == 8 ? 0x00FFFFFFu : 0x1FFFFFFFu; }
KNG_DEV uint32_t p29x2(int i) { return 2u * p29(i); }
KNG_DEV uint32_t p29x4(int i) { return 4u * p29(i); }

// 256-bit little-endian words <-> nine 29-bit limbs (value < 2^256 in, normalised limbs out)
KNG_DEV fe29 fe29_unpack(const fe &a) {
    uint32_t w[8];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        w[2 * i] = (uint32_t)a.v[i];
        w[2 * i + 1] = (uint32_t)(a.v[i] >> 32);
    }
    fe29 r;
    r.l[0] = w[0] & M29;
    r.l[1] = ((w[0] >> 29) | (w[1] << 3)) & M29;
    r.l[2] = ((w[1] >> 26) | (w[2] << 6)) & M29;
    r.l[3] = ((w[2] >> 23) | (w[3] << 9)) & M29;
    r.l[4] = ((w[3] >> 20) | (w[4] << 12)) & M29;
    r.l[5] = ((w[4] >> 17) | (w[5] << 15)) & M29;
    r.l[6] = ((w[5] >> 14) | (w[6] << 18)) & M29;
    r.l[7] = ((w[6] >> 11) | (w[7] << 21)) & M29;
    r.l[8] = w[7] >> 8;
    return r;
}
// canonical (normalised, < 2^256) limbs -> 256-bit words
KNG_DEV fe fe29_pack(const fe29 &a) {
    uint32_t w[8];
    w[0] = a.l[0] | (a.l[1] << 29);
    w[1] = (a.l[1] >> 3) | (a.l[2] << 26);
    w[2] = (a.l[2] >> 6) | (a.l[3] << 23);
    w[3] = (a.l[3] >> 9) | (a.l[4] << 20);
    w[4] = (a.l[4] >> 12) | (a.l[5] << 17);
    w[5] = (a.l[5] >> 15) | (a.l[6] << 14);
    w[6] = (a.l[6] >> 18) | (a.l[7] << 11);
    w[7] = (a.l[7] >> 21) | (a.l[8] << 8);
    return fe{{(uint64_t)w[0] | ((uint64_t)w[1] << 32), (uint64_t)w[2] | ((uint64_t)w[3] << 32),
               (uint64_t)w[4] | ((uint64_t)w[5] << 32), (uint64_t)w[6] | ((uint64_t)w[7] << 32)}};
}

// fold everything at or above 2^256 back in: limbs normalised on entry except that `over` (units of
// 2^261) and the top 5 bits of l[8] carry the excess.  2^256 = 2^32 + 977 (mod p): 977 at limb 0,
// 2^32 = 2^3 * 2^29 at limb 1.  Leaves l[0..2] < 2^29, l[3] <= 2^29, l[8] < 2^24.
KNG_DEV void fe29_fold_top(fe29 &r, uint64_t over) {
    const uint64_t top = (over << 5) | (r.l[8] >> 24);
    r.l[8] &= 0x00FFFFFFu;
    uint64_t c = (uint64_t)r.l[0] + top * 977u;
    r.l[0] = (uint32_t)c & M29;
    c >>= 29;
    c += (uint64_t)r.l[1] + (top << 3);
    r.l[1] = (uint32_t)c & M29;
    c >>= 29;
    c += r.l[2];
    r.l[2] = (uint32_t)c & M29;
    c >>= 29;
    r.l[3] += (uint32_t)c;
}

} // namespace kng
#include "kng_mul29.h"
namespace kng {

// r = a*b mod p.  a: limbs < 2^31 (lazy), b: limbs < 2^29 + 2^21 (a product or a loaded element).
// Result: "almost reduced" -- limbs < 2^29 (l[3] <= 2^29), l[8] < 2^24, value < 2^256 + 2^117.
// Columns k+9 (high) and k (low) are interleaved so that each high limb is folded as soon as
// it is known: low_k += h_k * 0x7A20 + (h_{k-1} << 8).  99 MADs, no carry instruction
// The MAD chains are generated asm (kng_mul29.h), the masks/shifts between them plain C++.  Besides
// saving the 64-bit adds hipcc would insert, the asm is needed for correctness: the straight C++
// accumulation `cl += (uint64_t)a*b` of this function was MISCOMPILED by hipcc (ROCm 7.2) when
// inlined behind another product (limbs 0-2 and 8 of s*s wrong on the GPU, same source correct on
// the host); tests/test_gpu_parity.py::test_radix29_full_jump_sequence pins the composition.
KNG_DEV fe29 fe29_mul(const fe29 &a, const fe29 &b) {
    fe29 r;
    // multiplier constants live in VGPRs: an inline-asm "v" operand cannot be a literal
    uint32_t c256 = 256u, r0 = R0_29;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(c256));
    asm("" : "+v"(r0));
#endif
    uint64_t over;
    mul29_columns(r.l, &over, a.l, b.l, r0, c256);
    fe29_fold_top(r, over);
    return r;
}

KNG_DEV fe29 fe29_sqr(const fe29 &a) { return fe29_mul(a, a); }

// carry pass over lazy limbs (each < 2^32 - 16) + top fold -> almost reduced (as fe29_mul's result)
KNG_DEV fe29 fe29_norm(const fe29 &a) {
    fe29 r;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t t = a.l[i] + c;
        r.l[i] = t & M29;
        c = t >> 29;
    }
    fe29_fold_top(r, c);
    return r;
}

// Slow half of fe29_canon: exact for ANY input with limbs < 2^30 and value < 2^261.  Kept out of
// line: it runs for a whole wave about once per 2^18 calls.
KNG_DEV_NOINLINE fe29 fe29_canon_slow(const fe29 &in) {
    fe29 r = in;
    // carry pass + fold of everything at or above 2^256; three rounds reach a fixed point
    for (int rep = 0; rep < 3; rep++) {
        uint32_t c = 0;
        fe29 n;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const uint32_t t = r.l[i] + c;
            n.l[i] = t & M29;
            c = t >> 29;
        }
        const uint32_t q = (c << 5) | (n.l[8] >> 24);
        n.l[8] &= 0x00FFFFFFu;
        n.l[0] += q * 977u;
        n.l[1] += q << 3;
        r = n;
    }
    // normalised and < 2^256 now.  value >= p  <=>  value + (2^32 + 977) carries out of 2^256
    fe29 t;
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        const uint32_t add = i == 0 ? 977u : i == 1 ? 8u : 0u;
        const uint32_t s = r.l[i] + add + c;
        t.l[i] = s & M29;
        c = s >> 29;
    }
    const bool ge = (t.l[8] >> 24) != 0; // value >= p: take value - p = value + K - 2^256
    t.l[8] &= 0x00FFFFFFu;
    fe29 o;
#pragma unroll
    for (int i = 0; i < 9; i++) o.l[i] = ge ? t.l[i] : r.l[i];
    return o;
}

// exact canonical form in [0,p): normalised limbs, l[8] < 2^24.  Input: lazy limbs (< 2^32 - 16).
// The fast path is straight-line; the two events that need more work (a carry rippling past
// limb 3, or a value in [p, 2^256)) have probability ~2^-29 / ~2^-24 per call and are handled by
// the out-of-line slow half behind a wave-uniform branch, so the result is always exact.
KNG_DEV fe29 fe29_canon(const fe29 &a) {
    fe29 r = fe29_norm(a); // l[0..2] < 2^29, l[3] <= 2^29, l[4..7] < 2^29, l[8] < 2^24
    const bool again = (r.l[3] >> 29) != 0 || r.l[8] == 0x00FFFFFFu;
    if (again) r = fe29_canon_slow(r); // ordinary divergent branch: skipped (execz) by practically every wave
    return r;
}

// lazy subtraction helpers (all full-rate limb-wise adds)
// a + (2p - b), with nb = 2p - b precomputed limb-wise: limbs < 2^29.x + 2^30
KNG_DEV fe29 fe29_add(const fe29 &a, const fe29 &nb) {
    fe29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + nb.l[i];
    return r;
}
// 2p - b, b normalised (limbs < 2^29 + 2^21 is fine: 2p limbs are >= 2^30 - 2^12 except the top one,
// and the top limb of b is < 2^24 + 1 <= 2^25 - 2)
KNG_DEV fe29 fe29_neg2p(const fe29 &b) {
    fe29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = p29x2(i) - b.l[i];
    return r;
}
// a - b + 2p
KNG_DEV fe29 fe29_sub2p(const fe29 &a, const fe29 &b) {
    fe29 r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.l[i] = a.l[i] + (p29x2(i) - b.l[i]);
    return r;
}

} // namespace kng
