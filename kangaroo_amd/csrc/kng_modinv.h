// kng_modinv.h -- modular inverse mod p for gfx950 (device code).
//
// Replaces GPU/GPUMath.h:700-803 (_ModInv: delayed-right-shift-62 binary GCD with Pornin-style
// divstep, variable time).  The contract is only "canonical inverse in [0,p), inverse of 0 is 0"
// (GPUMath.h:795-801 == SECPK1/IntMod.cpp:560-565), so any algorithm is bit-identical.
//
// In SIMT one inversion costs a wave the same whether 1 or 64 lanes need it, and a
// data-dependent GCD makes the 64 lanes of a wave64 diverge (the wave runs the worst lane).
// We therefore use a FIXED-FLOW algorithm: Bernstein-Yang "safegcd" division steps in the
// half-delta form, 600 = 20 x 30 steps on signed 30-bit limbs.  Everything is 32-bit VALU work
// (the inner step is ~17 and/xor/add/shift instructions with the matrix rows packed two per register, no
// branches); the
// 2x2 transition matrix of each 30-step block is applied to the 9-limb f,g / d,e with
// v_mad_i64_i32.  Measured ~7x cheaper than the Fermat ladder a^(p-2)
// (tools/fe_extras.h keeps one for cross-checking).
#pragma once

#include "kng_field.h"

namespace kng {

constexpr int32_t M30 = 0x3FFFFFFF;
// p in 30-bit limbs, and p^-1 mod 2^30
constexpr int32_t P30_0 = 0x3FFFFC2F, P30_1 = 0x3FFFFFFB, P30_MID = 0x3FFFFFFF, P30_8 = 0xFFFF;
constexpr uint32_t PINV30 = 0x2DDACACF;

KNG_DEV int32_t p30(int i) { return i == 0 ? P30_0 : i == 1 ? P30_1 : i == 8 ? P30_8 : P30_MID; }

// 15 division steps on the low bits of (f, g) with the matrix rows PACKED two entries per register:
//   P = u + v*2^16,  Q = q + r*2^16   (plain 32-bit two's-complement words, identities mod 2^32)
// Every update of a row is linear (conditional negation, masked addition, doubling), so it acts on the packed
// word exactly as on the two halves: 5 instructions per step for the four entries instead of 10.
// Bounds after 15 steps (all 3^15 event sequences enumerated): q, r in [-32767, 32767]; u, v even, in
// [-32766, 32768] -- the decode below maps the half-word 0x8000 to +32768 accordingly.
KNG_DEV int32_t divsteps15_packed(int32_t zeta, uint32_t &f, uint32_t &g, int32_t &tu, int32_t &tv, int32_t &tq, int32_t &tr) {
    uint32_t P = 1u, Q = 1u << 16;
#pragma unroll
    for (int i = 0; i < 15; i++) {
        uint32_t c1 = (uint32_t)(zeta >> 31); // all ones when zeta < 0
        const uint32_t c2 = 0u - (g & 1u);    // all ones when g is odd
        const uint32_t x = (f ^ c1) - c1;     // +-f
        const uint32_t Y = (P ^ c1) - c1;     // +-(u, v)
        g += x & c2;
        Q += Y & c2;
        c1 &= c2;                             // swap: zeta < 0 and g odd
        zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
        f += g & c1;
        P += Q & c1;
        g >>= 1;
        P <<= 1;
    }
    // q = sext16(Q), r = (Q - q) >> 16 ; u = sext16(P - 1) + 1, v = ((P - u - 2^16) >> 16) + 1
    tq = (int32_t)(int16_t)(uint16_t)Q;
    tr = (int32_t)(Q - (uint32_t)tq) >> 16;
    tu = (int32_t)(int16_t)(uint16_t)(P - 1u) + 1;
    tv = ((int32_t)(P - (uint32_t)tu - 0x10000u) >> 16) + 1;
    return zeta;
}

// 30 division steps on the low bits of (f, g); returns the new zeta and the transition
// matrix t = [[u,v],[q,r]] scaled by 2^30:  2^30 * (f', g') = t * (f, g).  Two packed 15-step halves,
// t = t2 * t1 (entries < 2^15 in magnitude: 24-bit multiplies, sums < 2^31).
KNG_DEV int32_t divsteps30(int32_t zeta, uint32_t f, uint32_t g, int32_t &tu, int32_t &tv, int32_t &tq, int32_t &tr) {
    int32_t u1, v1, q1, r1, u2, v2, q2, r2;
    zeta = divsteps15_packed(zeta, f, g, u1, v1, q1, r1);
    zeta = divsteps15_packed(zeta, f, g, u2, v2, q2, r2);
    tu = u2 * u1 + v2 * q1;
    tv = u2 * v1 + v2 * r1;
    tq = q2 * u1 + r2 * q1;
    tr = q2 * v1 + r2 * r1;
    return zeta;
}

// Signed 32x32+64 multiply-accumulate chains of the matrix application.  From C++ hipcc builds each
// int64 product out of v_mad_u64_u32 + v_mul_lo_u32 sign corrections (186 multiply-class
// instructions per round for 90 products); v_mad_i64_i32 does it in one.  Two accumulators are
// interleaved per statement; vcc only receives the (unused) carry-out.
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define KNG_FG_STEP(cf, cg, u, v, q, r, fi, gi)                                                            \
    asm("v_mad_i64_i32 %0, vcc, %2, %6, %0\n\tv_mad_i64_i32 %1, vcc, %4, %6, %1\n\t"                         \
        "v_mad_i64_i32 %0, vcc, %3, %7, %0\n\tv_mad_i64_i32 %1, vcc, %5, %7, %1"                             \
        : "+v"(cf), "+v"(cg)                                                                               \
        : "v"(u), "v"(v), "v"(q), "v"(r), "v"(fi), "v"(gi)                                                 \
        : "vcc")
#define KNG_DE_STEP(cd, ce, u, v, q, r, di, ei, md, me, pi)                                                \
    asm("v_mad_i64_i32 %0, vcc, %2, %6, %0\n\tv_mad_i64_i32 %1, vcc, %4, %6, %1\n\t"                         \
        "v_mad_i64_i32 %0, vcc, %3, %7, %0\n\tv_mad_i64_i32 %1, vcc, %5, %7, %1\n\t"                         \
        "v_mad_i64_i32 %0, vcc, %8, %10, %0\n\tv_mad_i64_i32 %1, vcc, %9, %10, %1"                            \
        : "+v"(cd), "+v"(ce)                                                                               \
        : "v"(u), "v"(v), "v"(q), "v"(r), "v"(di), "v"(ei), "v"(md), "v"(me), "v"(pi)                      \
        : "vcc")
#else
#define KNG_FG_STEP(cf, cg, u, v, q, r, fi, gi)                                                            \
    do {                                                                                                   \
        cf += (int64_t)(u) * (fi) + (int64_t)(v) * (gi);                                                   \
        cg += (int64_t)(q) * (fi) + (int64_t)(r) * (gi);                                                   \
    } while (0)
#define KNG_DE_STEP(cd, ce, u, v, q, r, di, ei, md, me, pi)                                                \
    do {                                                                                                   \
        cd += (int64_t)(u) * (di) + (int64_t)(v) * (ei) + (int64_t)(pi) * (md);                            \
        ce += (int64_t)(q) * (di) + (int64_t)(r) * (ei) + (int64_t)(pi) * (me);                            \
    } while (0)
#endif

// (f, g) <- t * (f, g) / 2^30   (exact)
KNG_DEV void update_fg30(int32_t f[9], int32_t g[9], int32_t u, int32_t v, int32_t q, int32_t r) {
    int64_t cf = 0, cg = 0;
    KNG_FG_STEP(cf, cg, u, v, q, r, f[0], g[0]);
    cf >>= 30;
    cg >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        KNG_FG_STEP(cf, cg, u, v, q, r, f[i], g[i]);
        f[i - 1] = (int32_t)cf & M30;
        g[i - 1] = (int32_t)cg & M30;
        cf >>= 30;
        cg >>= 30;
    }
    f[8] = (int32_t)cf;
    g[8] = (int32_t)cg;
}

// (d, e) <- t * (d, e) / 2^30 mod p, keeping both in (-2p, p)
KNG_DEV void update_de30(int32_t d[9], int32_t e[9], int32_t u, int32_t v, int32_t q, int32_t r, int32_t pc0,
                         int32_t pc1, int32_t pcm, int32_t pc8) {
    const int32_t sd = d[8] >> 31, se = e[8] >> 31; // sign masks
    // multiples of p that bring negative inputs back up ...
    int32_t md = (u & sd) + (v & se);
    int32_t me = (q & sd) + (r & se);
    int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0];
    int64_t ce = (int64_t)q * d[0] + (int64_t)r * e[0];
    // ... minus the multiple that clears the low 30 bits
    md -= (int32_t)((PINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
    me -= (int32_t)((PINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
    cd += (int64_t)P30_0 * md;
    ce += (int64_t)P30_0 * me;
    cd >>= 30;
    ce >>= 30;
#pragma unroll
    for (int i = 1; i < 9; i++) {
        const int32_t pi = i == 1 ? pc1 : i == 8 ? pc8 : pcm;
        KNG_DE_STEP(cd, ce, u, v, q, r, d[i], e[i], md, me, pi);
        d[i - 1] = (int32_t)cd & M30;
        e[i - 1] = (int32_t)ce & M30;
        cd >>= 30;
        ce >>= 30;
    }
    (void)pc0;
    d[8] = (int32_t)cd;
    e[8] = (int32_t)ce;
}

// canonical inverse in [0,p); 0 -> 0
KNG_DEV_NOINLINE fe fe_inv(const fe &a_in) {
    const fe a = fe_canon(a_in);
    int32_t f[9], g[9], d[9], e[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        f[i] = p30(i);
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    g[0] = (int32_t)(a.v[0] & M30);
    g[1] = (int32_t)((a.v[0] >> 30) & M30);
    g[2] = (int32_t)(((a.v[0] >> 60) | (a.v[1] << 4)) & M30);
    g[3] = (int32_t)((a.v[1] >> 26) & M30);
    g[4] = (int32_t)(((a.v[1] >> 56) | (a.v[2] << 8)) & M30);
    g[5] = (int32_t)((a.v[2] >> 22) & M30);
    g[6] = (int32_t)(((a.v[2] >> 52) | (a.v[3] << 12)) & M30);
    g[7] = (int32_t)((a.v[3] >> 18) & M30);
    g[8] = (int32_t)(a.v[3] >> 48);

    // the limbs of p as register operands of the asm MAD chains (an inline-asm "v" operand cannot be a literal)
    int32_t pc0 = P30_0, pc1 = P30_1, pcm = P30_MID, pc8 = P30_8;
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(pc1));
    asm("" : "+v"(pcm));
    asm("" : "+v"(pc8));
#endif
    int32_t zeta = -1;
#pragma unroll 1
    for (int it = 0; it < 20; it++) { // 600 >= 590 division steps suffice for 256-bit inputs
        int32_t u, v, q, r;
        const uint32_t f0 = (uint32_t)f[0] | ((uint32_t)f[1] << 30);
        const uint32_t g0 = (uint32_t)g[0] | ((uint32_t)g[1] << 30);
        zeta = divsteps30(zeta, f0, g0, u, v, q, r);
        update_de30(d, e, u, v, q, r, pc0, pc1, pcm, pc8);
        update_fg30(f, g, u, v, q, r);
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
        // g == 0 is a fixed point (f stays +-1, d stays the answer): leave as soon as EVERY lane of
        // the wave got there -- wave-uniform, so no divergence; typically saves 1-3 of the 20 rounds
        {
            uint32_t nz = 0;
#pragma unroll
            for (int i = 0; i < 9; i++) nz |= (uint32_t)g[i];
            if (__ballot(nz != 0) == 0) break;
        }
#endif
    }
    // g == 0 now and f == +-1 (or f == p when a == 0, then d == 0).  result = sign(f) * d mod p
    const int32_t sf = f[8] >> 31;
    // d in (-2p, p): add p if negative, conditionally negate, add p if negative again
    int32_t cond = d[8] >> 31;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int32_t t = d[i] + (p30(i) & cond);
        t = (t ^ sf) - sf;
        t += c;
        c = t >> 30;
        d[i] = t & M30;
    }
    d[8] += c << 30; // keep the sign in the top limb
    cond = d[8] >> 31;
    c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int32_t t = d[i] + (p30(i) & cond) + c;
        c = t >> 30;
        d[i] = t & M30;
    }
    d[8] += c << 30;
    fe r;
    r.v[0] = (uint64_t)(uint32_t)d[0] | ((uint64_t)(uint32_t)d[1] << 30) | ((uint64_t)(uint32_t)d[2] << 60);
    r.v[1] = ((uint64_t)(uint32_t)d[2] >> 4) | ((uint64_t)(uint32_t)d[3] << 26) | ((uint64_t)(uint32_t)d[4] << 56);
    r.v[2] = ((uint64_t)(uint32_t)d[4] >> 8) | ((uint64_t)(uint32_t)d[5] << 22) | ((uint64_t)(uint32_t)d[6] << 52);
    r.v[3] = ((uint64_t)(uint32_t)d[6] >> 12) | ((uint64_t)(uint32_t)d[7] << 18) | ((uint64_t)(uint32_t)d[8] << 48);
    return r;
}

#if defined(__HIPCC__)
// ---- one inversion, two waves (round 6; small herds: a launch of 65 536 kangaroos is 64 SERIAL inversions).
// A lone wave issues one dependent VALU instruction every ~5 cycles, so the latency of fe_inv is its instruction count: per
// round 360 for the 30 division steps, ~80 for (f, g) <- t (f, g) / 2^30, ~115 for (d, e) <- t (d, e) / 2^30 mod p.  Only the
// first two are on the critical path: the division steps of round r + 1 need f, g of round r, nobody needs d, e before the
// end.  The LEAD wave therefore runs division steps + update_fg30 and publishes each round's matrix in LDS; the FOLLOW wave --
// another wave of the block, on another SIMD, idle at the barrier otherwise -- applies the matrices to d, e one round behind
// and finishes with the sign of f.  Same arithmetic, same result as fe_inv; 1.917 -> 1.72 ms per launch at 65 536 kangaroos.
// (Measured and NOT adopted: a third wave applying the matrices to f, g as well, the lead keeping only the three low limbs
// and borrowing limb 2 back every round -- the LDS round trip on the lead's critical path costs more than the 50 instructions
// it saves: 1.81 ms, and 1.90 ms with the reads issued between the two halves of the division steps;
// profiles/r06_small_herd_waves_ab.txt.)
struct InvRing {
    int32_t m[20][4][64]; // u, v, q, r of every round, per lane of the lead wave
    int32_t sf[64];       // sign of the final f
    uint32_t progress;    // rounds published so far; | 0x100 once the lead has left its loop.  The follow wave resets it.
};
#define KNG_RING_LOAD(p) __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
#define KNG_RING_STORE(p, v) __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP)

KNG_DEV void fe_to30(const fe &a, int32_t g[9]) {
    g[0] = (int32_t)(a.v[0] & M30);
    g[1] = (int32_t)((a.v[0] >> 30) & M30);
    g[2] = (int32_t)(((a.v[0] >> 60) | (a.v[1] << 4)) & M30);
    g[3] = (int32_t)((a.v[1] >> 26) & M30);
    g[4] = (int32_t)(((a.v[1] >> 56) | (a.v[2] << 8)) & M30);
    g[5] = (int32_t)((a.v[2] >> 22) & M30);
    g[6] = (int32_t)(((a.v[2] >> 52) | (a.v[3] << 12)) & M30);
    g[7] = (int32_t)((a.v[3] >> 18) & M30);
    g[8] = (int32_t)(a.v[3] >> 48);
}

KNG_DEV void fe_inv_lead(const fe &a_in, InvRing *ring) {
    const fe a = fe_canon(a_in);
    const uint32_t lane = threadIdx.x & 63;
    int32_t f[9], g[9];
#pragma unroll
    for (int i = 0; i < 9; i++) f[i] = p30(i);
    fe_to30(a, g);
    int32_t zeta = -1;
    uint32_t rounds = 0;
#pragma unroll 1
    for (int it = 0; it < 20; it++) {
        int32_t u, v, q, r;
        zeta = divsteps30(zeta, (uint32_t)f[0] | ((uint32_t)f[1] << 30), (uint32_t)g[0] | ((uint32_t)g[1] << 30), u, v, q, r);
        ring->m[it][0][lane] = u;
        ring->m[it][1][lane] = v;
        ring->m[it][2][lane] = q;
        ring->m[it][3][lane] = r;
        rounds = (uint32_t)it + 1;
        if (lane == 0) KNG_RING_STORE(&ring->progress, rounds);
        update_fg30(f, g, u, v, q, r);
        // (measured and left: testing for the exit only from round 15 on, and publishing behind the update so that the release store
        // finds the matrix stores landed -- 1.64-1.66 ms against 1.63 at 65 536 kangaroos, profiles/r06_small_herd_flat*_ab.txt)
        uint32_t nz = 0;
#pragma unroll
        for (int i = 0; i < 9; i++) nz |= (uint32_t)g[i];
        if (__ballot(nz != 0) == 0) break; // (wave-uniform, as in fe_inv)
    }
    ring->sf[lane] = f[8] >> 31;
    if (lane == 0) KNG_RING_STORE(&ring->progress, rounds | 0x100u);
}

KNG_DEV fe fe_inv_follow(InvRing *ring) {
    const uint32_t lane = threadIdx.x & 63;
    int32_t d[9], e[9];
#pragma unroll
    for (int i = 0; i < 9; i++) {
        d[i] = 0;
        e[i] = 0;
    }
    e[0] = 1;
    int32_t pc0 = P30_0, pc1 = P30_1, pcm = P30_MID, pc8 = P30_8;
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(pc1));
    asm("" : "+v"(pcm));
    asm("" : "+v"(pc8));
#endif
    uint32_t done = 0;
#pragma unroll 1
    for (;;) {
        const uint32_t pr = KNG_RING_LOAD(&ring->progress);
        if ((pr & 0xFFu) > done) {
            const int32_t u = ring->m[done][0][lane], v = ring->m[done][1][lane], q = ring->m[done][2][lane], r = ring->m[done][3][lane];
            update_de30(d, e, u, v, q, r, pc0, pc1, pcm, pc8);
            done++;
        } else if (pr & 0x100u) {
            break;
        } else {
            __builtin_amdgcn_s_sleep(1);
        }
    }
    const int32_t sf = ring->sf[lane];
    // the lead has left its loop and this wave has read everything: the ring is free for the next inversion (two barriers away)
    if (lane == 0) KNG_RING_STORE(&ring->progress, 0u);
    // result = sign(f) * d mod p, d in (-2p, p): the tail of fe_inv
    int32_t cond = d[8] >> 31;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int32_t t = d[i] + (p30(i) & cond);
        t = (t ^ sf) - sf;
        t += c;
        c = t >> 30;
        d[i] = t & M30;
    }
    d[8] += c << 30;
    cond = d[8] >> 31;
    c = 0;
#pragma unroll
    for (int i = 0; i < 9; i++) {
        int32_t t = d[i] + (p30(i) & cond) + c;
        c = t >> 30;
        d[i] = t & M30;
    }
    d[8] += c << 30;
    fe r;
    r.v[0] = (uint64_t)(uint32_t)d[0] | ((uint64_t)(uint32_t)d[1] << 30) | ((uint64_t)(uint32_t)d[2] << 60);
    r.v[1] = ((uint64_t)(uint32_t)d[2] >> 4) | ((uint64_t)(uint32_t)d[3] << 26) | ((uint64_t)(uint32_t)d[4] << 56);
    r.v[2] = ((uint64_t)(uint32_t)d[4] >> 8) | ((uint64_t)(uint32_t)d[5] << 22) | ((uint64_t)(uint32_t)d[6] << 52);
    r.v[3] = ((uint64_t)(uint32_t)d[6] >> 12) | ((uint64_t)(uint32_t)d[7] << 18) | ((uint64_t)(uint32_t)d[8] << 48);
    return r;
}
#endif

} // namespace kng
