// kng_engine.hip -- MI355X (gfx950) kangaroo jump engine: walk kernel + C ABI (include/kangaroo_hip.h).
//
// Replaces, from scratch, the reference's GPU/GPUEngine.cu + GPU/GPUCompute.h + GPU/GPUMath.h.
//
// Design:
//  * herd state lives in HBM as plane-major SoA of 16-byte vectors, indexed by kIdx:
//        X01[N] X23[N] Y01[N] Y23[N] D[N]   (+ scratch product planes S01[N] S23[N])
//    so that the 64 lanes of a wave read/write 1 KiB contiguous per instruction.
//  * lane t of L lanes owns kangaroos {t, t+L, t+2L, ...} (G = N/L of them) and amortises ONE
//    modular inversion over those G with Montgomery's trick, like the reference's 128 per
//    thread (GPUMath.h:1166-1190) -- but the G states are streamed through HBM instead of
//    living in 18 KB/lane of scratch ("local") memory.
//  * one jump of the whole group is ONE pass: the pass that consumes the running inverse
//    (backward over the previous pass's prefix products) also emits the prefix products of the
//    next jump's dx in its own order, so passes alternate direction and each kangaroo's state is
//    read once and written once per jump: 80 B in + 80 B out + 32 B product in + 32 B out.
//  * the 32-entry jump table (GPUMath.h:51-54 __constant__) is staged in LDS, limb-major, so
//    that the per-lane index x&31 hits 32 distinct bank pairs (conflict-free ds_read_b64).
//  * distinguished points are compacted per wave (ballot + popcount prefix) with one atomic
//    per wave and DP-bearing step instead of one per DP (GPUCompute.h:96-105).
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <thread>
#include <vector>

#include "../../include/kangaroo_hip.h"
#include "kng_field.h"
#include "kng_modinv.h"
#include "kng_walk_asm.h"

using namespace kng;

// --------------------------------------------------------------------------------------------
// device side
// --------------------------------------------------------------------------------------------

typedef ulonglong2 v16; // one 16-byte vector = two limbs

struct DpRecord { // 64-byte device record: four aligned 16-byte stores
    uint64_t x[4];
    uint64_t d[2];
    uint64_t kidx;
    uint64_t pad;
};

struct WalkArgs {
    v16 *x01, *x23, *y01, *y23, *d, *s01, *s23;
    const uint64_t *jtab; // limb-major: jx[4][32] jy[4][32] jd[2][32]
    uint64_t dp_mask;
    uint32_t *dp_count;
    DpRecord *dp_items;
    uint32_t max_found;
    uint32_t lanes; // L
    uint32_t group; // nominal G (exact when lanes divides the herd)
    uint64_t n_kang; // N: lane t walks kangaroos t, t+L, ... < N, i.e. ceil((N-t)/L) of them (uniform per wave: L is a multiple of 64)
    uint32_t nsteps;
    uint32_t resume; // 1: the previous launch left the prefix products of the next jump's dx in the S planes (ascending order): no pass 0
    uint64_t asm_args; // device address of this launch's WalkAsmArgs (the scheduled loop reads its constants with s_load)
};

// what the generated loop (kng_walk_asm.h, tools/gen_walk_asm.py: OFF_PLANES / OFF_DP) loads: plane bases, then the DP block
struct WalkAsmArgs {
    uint64_t x01, x23, y01, y23, dlo, dhi, s01, s23;
    uint64_t dp_mask, dp_count, dp_items;
    uint32_t max_found, pad;
};
static_assert(sizeof(WalkAsmArgs) == 96 && offsetof(WalkAsmArgs, dp_mask) == 0x40, "layout the generated loop expects");

#define JT_JX 0
#define JT_JY (4 * 32)
#define JT_JD (8 * 32)
#define JT_WORDS (10 * 32)

// The herd state (x, y, d) streams: every vector is read once and written once per jump and not touched again for a
// whole pass (~0.4 ms, ~1 GB of traffic later).  Non-temporal accesses ("nt": stream through L2 without
// displacing anything worth keeping) are worth +1.9 % on the walk (profiles/r01_ab_nontemporal.txt).
#ifndef KNG_NT_LOAD
#define KNG_NT_LOAD 1
#endif
#ifndef KNG_NT_STORE
#define KNG_NT_STORE 1
#endif
#ifndef KNG_NT_D
#define KNG_NT_D 0 // the 8-byte distance words: the hint does not pay there
#endif
typedef unsigned long long v2u64 __attribute__((ext_vector_type(2)));
KNG_DEV v16 ld_stream(const v16 *p) {
#if KNG_NT_LOAD
    const v2u64 v = __builtin_nontemporal_load(reinterpret_cast<const v2u64 *>(p));
    return make_ulonglong2(v.x, v.y);
#else
    return *p;
#endif
}
KNG_DEV void st_stream(v16 *p, unsigned long long a, unsigned long long b) {
#if KNG_NT_STORE
    v2u64 v;
    v.x = a;
    v.y = b;
    __builtin_nontemporal_store(v, reinterpret_cast<v2u64 *>(p));
#else
    *p = make_ulonglong2(a, b);
#endif
}
KNG_DEV uint64_t ld_stream64(const uint64_t *p) {
#if KNG_NT_D
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
KNG_DEV void st_stream64(uint64_t *p, uint64_t v) {
#if KNG_NT_D
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
// The running products are the exception to the streaming rule: a pass overwrites product slot k one iteration after it
// read that very slot (as the neighbour product of the kangaroo before); with the streaming hint the read does not keep
// the line and the write misses L2.  Plain accesses for the S planes: +17 % L2 hits, +1.2 % walk rate
// (profiles/r02_pmc_tcc_product_planes.txt, profiles/r02_ab_micro_variants.txt).
KNG_DEV fe ld_prod(const v16 *p01, const v16 *p23, size_t i) {
    const v16 a = p01[i], b = p23[i];
    return fe{{a.x, a.y, b.x, b.y}};
}
KNG_DEV void st_prod(v16 *p01, v16 *p23, size_t i, const fe &v) {
    p01[i] = make_ulonglong2(v.v[0], v.v[1]);
    p23[i] = make_ulonglong2(v.v[2], v.v[3]);
}
KNG_DEV fe ld_fe(const v16 *p01, const v16 *p23, size_t i) {
    const v16 a = ld_stream(p01 + i), b = ld_stream(p23 + i);
    return fe{{a.x, a.y, b.x, b.y}};
}
KNG_DEV void st_fe(v16 *p01, v16 *p23, size_t i, const fe &v) {
    st_stream(p01 + i, v.v[0], v.v[1]);
    st_stream(p23 + i, v.v[2], v.v[3]);
}
// The distance plane holds the N low words followed by the N high words (two 8-byte planes): a walk whose
// jump distances are far below 2^64 then only streams the low words (DSPLIT, see walk_body).
KNG_DEV v16 ld_d(const v16 *d, size_t n, size_t i) {
    const uint64_t *p = reinterpret_cast<const uint64_t *>(d);
    return make_ulonglong2(ld_stream64(p + i), ld_stream64(p + n + i));
}
KNG_DEV void st_d(v16 *d, size_t n, size_t i, const v16 &v) {
    uint64_t *p = reinterpret_cast<uint64_t *>(d);
    st_stream64(p + i, v.x);
    st_stream64(p + n + i, v.y);
}
KNG_DEV fe lds_fe(const uint64_t *tab, int base, uint32_t j) {
    return fe{{tab[base + j], tab[base + 32 + j], tab[base + 64 + j], tab[base + 96 + j]}};
}

// wave64 compaction of distinguished points: one atomic per wave per DP-bearing step
KNG_DEV void emit_dp(bool is_dp, const fe &x, const v16 &d, uint64_t kidx, const WalkArgs &a) {
    const uint64_t m = __ballot(is_dp);
    if (m == 0) return;
    const uint32_t lane = __lane_id();
    uint32_t base = 0;
    const int leader = __ffsll((unsigned long long)m) - 1;
    if ((int)lane == leader) base = atomicAdd(a.dp_count, (uint32_t)__popcll(m));
    base = __shfl(base, leader);
    if (is_dp) {
        const uint32_t pos = base + (uint32_t)__popcll(m & ((1ULL << lane) - 1ULL));
        if (pos < a.max_found) {
            v16 *rec = reinterpret_cast<v16 *>(&a.dp_items[pos]);
            rec[0] = make_ulonglong2(x.v[0], x.v[1]);
            rec[1] = make_ulonglong2(x.v[2], x.v[3]);
            rec[2] = d;
            rec[3] = make_ulonglong2(kidx, 0);
        }
    }
}

// One kangaroo's jump given its loaded state: P += J[x & 31] (GPUCompute.h:67-94), d += jD, store, DP test and record
// (GPUCompute.h:96-105), prefix product of the next jump's dx.  Shared by the compiler-scheduled loop (which prefetches
// cx/cy/cd/nb one kangaroo ahead) and by the exact path behind the scheduled asm loop (walk_one).
template <bool DSPLIT>
KNG_DEV void walk_core(const WalkArgs &a, const uint64_t *tab, uint64_t *dlo, uint64_t *dhi, size_t idx, const fe &cx, const fe &cy,
                       v16 cd, const fe &nb, bool have_nb, bool first, bool last, fe &inv, fe &acc) {
    const uint32_t j = (uint32_t)cx.v[0] & (KNG_NB_JUMP - 1);
    const fe jx = lds_fe(tab, JT_JX, j);
    const fe jy = lds_fe(tab, JT_JY, j);
    const fe dx = fe_sub(cx, jx);
    fe invk;
    if (have_nb) {
        invk = fe_mul(inv, nb); // 1/dx      (GPUMath.h:1182-1186)
        inv = fe_mul(inv, dx);  // 1/(product of the remaining dx)
    } else {
        invk = inv;
    }
    const fe dy = fe_sub(cy, jy);
    const fe s = fe_mul(dy, invk);
    const fe p2 = fe_sqr(s);
    const fe rx = fe_sub(fe_sub(p2, jx), cx);
    const fe ry = fe_sub(fe_mul(fe_sub(cx, rx), s), cy);
    // d += jD[j]: raw 128-bit add (GPUMath.h:119-121)
    bool hi_known = !DSPLIT;
    {
        const uint64_t jd0 = tab[JT_JD + j];
        unsigned long long c = 0;
        cd.x = __builtin_addcll(cd.x, jd0, 0, &c);
        if (DSPLIT) {
            if (__builtin_expect(c != 0, 0)) {
                KNG_RARE_PATH();
                // at L2, like the scheduled loop's carries (global_atomic_add_x2): the two must not meet through a stale L1 line
                cd.y = atomicAdd(reinterpret_cast<unsigned long long *>(dhi + idx), 1ULL) + 1;
                hi_known = true;
            }
        } else {
            cd.y = cd.y + tab[JT_JD + 32 + j] + c;
        }
    }
    st_fe(a.x01, a.x23, idx, rx);
    st_fe(a.y01, a.y23, idx, ry);
    st_stream64(dlo + idx, cd.x);
    if (!DSPLIT) st_stream64(dhi + idx, cd.y);

    // ---- distinguished point? (GPUCompute.h:96-105) ----
    {
        const bool is_dp = (rx.v[3] & a.dp_mask) == 0;
        if (DSPLIT && is_dp && !hi_known) {
            cd.y = __hip_atomic_load(dhi + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (sees this kernel's L2 atomics)
            // consume the value inside the branch: a load left pending at the join would make hipcc
            // drain every outstanding memory operation (vmcnt(0)) at the top of the next iteration
            asm volatile("" ::"v"(cd.y));
        }
        emit_dp(is_dp, rx, cd, (uint64_t)idx, a);
    }

    // ---- prefix product of the NEXT jump's dx, in this pass's order ----
    if (!last) {
        const uint32_t j2 = (uint32_t)rx.v[0] & (KNG_NB_JUMP - 1);
        const fe dx2 = fe_sub(rx, lds_fe(tab, JT_JX, j2));
        acc = first ? dx2 : fe_mul(acc, dx2);
        st_prod(a.s01, a.s23, idx, acc);
    }
}

// The hot kernel.  Replaces comp_kangaroos/ComputeKangaroos (GPUEngine.cu:35-40, GPUCompute.h:22-117).
//
// SHARE = 8 (512-thread blocks, option "share", the default): the eight waves of a CU share ONE inversion per jump
// through a two-level product tree (see the step loop).  Round 1-2's SHARE = 2 (one inversion per SIMD: waves w and
// w+4) is its first level; sharing the whole CU was +2.5 % on top and replaced it (every wave for itself against one
// inversion per CU, both at two waves per SIMD: +5.3 %, profiles/r03_ab_share.txt).
//     i = 1/(a*b) ;  1/a = i*b ;  1/b = i*a          (3 multiplications per pair and level)
// Results are unchanged (the canonical residue is the same).
//
// DSPLIT = true: the 128-bit distance only streams its LOW word through HBM.  d += jD[j] carries out of bit 64
// with probability jD/2^64 (2^-23 per jump at an 80-bit range, 2^-9.5 at 109 bits); the lanes that carry add 1 to their
// high word with an L2 atomic -- a cold block inside the scheduled loop, nothing waits for it -- and the high word is
// fetched (coherently) when a distinguished point is emitted.  Saves 16 of 224 B/jump.  The host enables it when every
// jump distance is below 2^58 (kng_set_params; 2^50 for the compiler-scheduled loop, whose carry path is a divergent
// read-modify-write): ranges up to 115 bits, BASELINE configs[3] included.  Results are identical.
//
// ASM = true (option "asm", the default): the per-kangaroo loop of every step runs as ONE scheduled asm
// statement (kng_walk_asm.h, generated by tools/gen_walk_asm.py) instead of the compiler-scheduled loop below -- same
// data flow, same results, no hazard nops and a third of the register moves.  Its short arithmetic forms flag the lanes
// for which they are not exact; the statement then returns BEFORE storing anything of that iteration and the wave runs
// it through walk_core (exact on every input), then re-enters the statement behind it.
template <int SHARE, bool DSPLIT, bool ASM, bool NOMEM = false>
KNG_DEV void walk_body(const WalkArgs &a, const uint64_t *tab, v16 *xch, InvRing *ring = nullptr) {
    uint64_t *const dlo = reinterpret_cast<uint64_t *>(a.d), *const dhi = dlo + a.n_kang;
    const size_t L = a.lanes;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (SHARE == 1 && t >= L) return;
    // SHARE: every thread reaches the barriers; lanes beyond L walk nothing (wave-uniform: L % 64 == 0)
    const uint32_t G = t < L ? (uint32_t)((a.n_kang - t + L - 1) / L) : 0;

    // pass 0: prefix products of dx in ascending order (GPUCompute.h:52-61 + GPUMath.h:1173-1177).
    // Round 4: EVERY step of a launch, the last one included, leaves the products of the next jump's dx behind, and an even
    // number of steps leaves them in ascending order -- exactly what pass 0 computes.  The host then starts the next launch
    // with `resume` (nothing touched the herd, the table or the S planes in between: kng_launch) and the pass -- 1/64 of a
    // launch's x reads and product writes, one multiplication per kangaroo -- is skipped; the lane only re-reads its
    // whole-batch product, the last slot of its chain.
    fe acc = fe_one();
    if (a.resume) {
        if (G) acc = ld_prod(a.s01, a.s23, (size_t)(G - 1) * L + t);
    } else {
        for (uint32_t g = 0; g < G; g++) {
            const size_t idx = (size_t)g * L + t;
            const fe x = ld_fe(a.x01, a.x23, idx);
            const uint32_t j = (uint32_t)x.v[0] & (KNG_NB_JUMP - 1);
            const fe dx = fe_sub(x, lds_fe(tab, JT_JX, j));
            acc = g ? fe_mul(acc, dx) : dx;
            st_prod(a.s01, a.s23, idx, acc);
        }
    }

    for (uint32_t step = 0; step < a.nsteps; step++) {
        // one inversion per lane per jump of the whole group (GPUMath.h:1179-1180)
        fe inv;
        if (SHARE == 8) {
            // ONE inversion per CU and jump: the eight waves of the 512-thread block (two per SIMD) form a two-level
            // product tree.  Level 1 as for SHARE = 2 (waves w and w+4, four SIMDs side by side: 1 multiplication up,
            // 2 down); level 2: wave 0 multiplies the four pair products (3), inverts ONCE, and walks back (6).
            // 3 of the 4 inversions of a CU-step (17 K instructions each) are traded for 9 serial multiplications
            // and two more barriers.  xch: slot s = two 64-lane rows of 16-byte halves.
            const uint32_t col = threadIdx.x & 63, w = threadIdx.x >> 6;
            auto put = [&](uint32_t slot, const fe &v) {
                xch[(2 * slot) * 64 + col] = make_ulonglong2(v.v[0], v.v[1]);
                xch[(2 * slot + 1) * 64 + col] = make_ulonglong2(v.v[2], v.v[3]);
            };
            auto get = [&](uint32_t slot) -> fe {
                const v16 b0 = xch[(2 * slot) * 64 + col], b1 = xch[(2 * slot + 1) * 64 + col];
                return fe{{b0.x, b0.y, b1.x, b1.y}};
            };
            // (Round 4 measured two variations here and kept neither: loading the coming pass's first kangaroo before these
            // barriers so that its lines wait in L2 -- -0.6 % when consumed at once, +-0 when consumed behind the inversion --
            // and two 256-thread blocks per CU with one tree level each, so that one block's inversion overlaps the other's
            // walking -- -3.4 %, -1.5 % with a rotating root wave.  profiles/r04_ab_warm_l2.txt, profiles/r04_ab_kernel.txt.)
            if (w >= 4) put(w, acc);
            __syncthreads();
            fe pb = fe_one(), pre = fe_one(), i = fe_one();
            if (w < 4) {
                pb = get(w + 4);
                pre = fe_mul(acc, pb);
                if (w) put(w, pre);
            }
            __syncthreads();
            if (w == 0) {
                const fe q1 = get(1), q2 = get(2), q3 = get(3);
                const fe m1 = fe_mul(pre, q1), m2 = fe_mul(m1, q2), m3 = fe_mul(m2, q3);
#ifdef KNG_ABL_NOINV // measurement builds only (tools/archive/r3_sensitivity.sh): what the one inversion per CU-step costs
                fe t = m3;
#else
                fe t = fe_inv(m3);
#endif
                const fe i3 = fe_mul(t, m2); // 1/q3
                t = fe_mul(t, q3);           // 1/m2
                const fe i2 = fe_mul(t, m1); // 1/q2
                t = fe_mul(t, q2);           // 1/m1
                const fe i1 = fe_mul(t, pre); // 1/q1
                i = fe_mul(t, q1);            // 1/pre of wave 0
                put(1, i1);
                put(2, i2);
                put(3, i3);
            }
            __syncthreads();
            if (w > 0 && w < 4) i = get(w);
            if (w < 4) {
                put(w + 4, fe_mul(i, acc)); // 1/pb
                inv = fe_mul(i, pb);        // 1/acc
            }
            __syncthreads();
            if (w >= 4) inv = get(w);
            if (G == 0) continue;
        } else if (SHARE == 4) {
            // Small herds (fewer lanes than 512 x CUs): 256-thread blocks, one wave per SIMD, so that EVERY CU gets a block
            // -- a launch of such a herd is 64 serial inversions plus 64 walks of one kangaroo per lane, nothing overlaps, and
            // 512-thread blocks would leave half the CUs idle while two waves share each SIMD of the others.  One tree level:
            // waves 2, 3 hand their products to waves 0, 1; wave 0 multiplies the two pair products, inverts ONCE, walks back.
            const uint32_t col = threadIdx.x & 63, w = threadIdx.x >> 6;
            auto put = [&](uint32_t slot, const fe &v) {
                xch[(2 * slot) * 64 + col] = make_ulonglong2(v.v[0], v.v[1]);
                xch[(2 * slot + 1) * 64 + col] = make_ulonglong2(v.v[2], v.v[3]);
            };
            auto get = [&](uint32_t slot) -> fe {
                const v16 b0 = xch[(2 * slot) * 64 + col], b1 = xch[(2 * slot + 1) * 64 + col];
                return fe{{b0.x, b0.y, b1.x, b1.y}};
            };
#ifdef KNG_INV_ONE_WAVE // measurement builds only (tools/build_variant.sh): round 5's form -- a pair tree, wave 0 inverts alone
            if (w >= 2) put(w, acc);
            __syncthreads();
            fe pb = fe_one(), pre = fe_one(), i = fe_one();
            if (w < 2) {
                pb = get(w + 2);
                pre = fe_mul(acc, pb);
                put(w, pre);
            }
            __syncthreads();
            if (w == 0) {
                const fe q1 = get(1);
                const fe t = fe_inv(fe_mul(pre, q1));
                put(1, fe_mul(t, pre)); // 1/q1
                i = fe_mul(t, q1);      // 1/pre of wave 0
            }
            __syncthreads();
            if (w == 1) i = get(1);
            if (w < 2) {
                put(w + 2, fe_mul(i, acc)); // 1/pb
                inv = fe_mul(i, pb);        // 1/acc
            }
            __syncthreads();
            if (w >= 2) inv = get(w);
#else
            // Round 6.  (i) ONE inversion on two waves (kng_modinv.h): wave 0 leads -- division steps and f, g --, wave 1 follows
            // one round behind with d, e and ends up holding t = 1 / (a0 a1 a2 a3).  Round 5: wave 0 alone, waves 1-3 at the
            // barrier: the 64 inversions of a launch of such a herd ARE the launch (1.91 ms of 1.91 ms at 65 536 kangaroos).
            // (ii) What follows the inversion is ONE multiplication per wave: every wave has multiplied the products of the
            // OTHER three together while it had nothing to do (round 5 and the first form of this round walked the pair tree
            // back: two multiplications on the follower, a barrier, two on waves 0 and 1, a barrier -- all of it serial latency).
            put(w, acc);
            __syncthreads();
            const fe a0 = get(0), a1 = get(1), a2 = get(2), a3 = get(3);
            fe p01 = fe_one();
            if (w == 0) put(4, p01 = fe_mul(a0, a1));
            if (w == 2) put(5, fe_mul(a2, a3));
            __syncthreads();
            fe others; // the product of the other three waves' products
            if (w == 0) {
                const fe p23 = get(5);
                fe_inv_lead(fe_mul(p01, p23), ring);
                others = fe_mul(a1, p23); // (while the follower finishes)
            } else if (w == 1) {
                others = fe_mul(a0, get(5)); // (while the lead runs its first division steps)
                put(6, fe_inv_follow(ring));
            } else {
                others = fe_mul(get(4), w == 2 ? a3 : a2);
            }
            __syncthreads();
            inv = fe_mul(get(6), others); // 1 / acc
#endif
            if (G == 0) continue;
        } else {
            inv = fe_inv(acc);
        }
        const bool backward = !(step & 1); // reverse of the pass that produced the products
        const bool last = false; // (rounds 1-3: the last step of a launch skipped the products of the next jump; see pass 0)
        // slot(k): kangaroo processed k-th in this pass
        auto slot = [&](uint32_t k) -> size_t { return (size_t)(backward ? (G - 1 - k) : k) * L + t; };

#if defined(__HIP_DEVICE_COMPILE__)
        if (ASM) {
            // G is wave-uniform (L is a multiple of 64); the statement keeps its loop counter in an SGPR
            const uint32_t Gs = __builtin_amdgcn_readfirstlane(G);
            // (readfirstlane: the "s" operands must be provably wave-uniform for the compiler)
            const uint32_t ldstab = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)tab); // LDS byte address = low half of the flat address
            const int32_t stride = __builtin_amdgcn_readfirstlane(backward ? -(int32_t)(L * 16) : (int32_t)(L * 16));
            // (the builtin returns int: widen through uint32_t, or a low word with bit 31 set sign-extends over the high word)
            const uint64_t aargs = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(a.asm_args >> 32)) << 32) |
                                   (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)a.asm_args);
            uint32_t iv[8], ac[8] = {1, 0, 0, 0, 0, 0, 0, 0}; // acc' = acc * dx2 with acc = 1 for the first kangaroo (exact)
            fe_to32(iv, inv);
            uint32_t k = 0;
            while (true) {
                uint32_t voff = (uint32_t)slot(k) * 16u;
                uint32_t ks = __builtin_amdgcn_readfirstlane(k);
                if (NOMEM) KNG_WALK_ASM_LOOP(KNG_WALK_ASM_TEXT_VALU, KNG_WALK_ASM_CLOBBERS_VALU, iv, ac, ks, voff, aargs, stride, Gs, ldstab); // measurement, option "asm" 2
                else if (DSPLIT) KNG_WALK_ASM_LOOP(KNG_WALK_ASM_TEXT_DSPLIT, KNG_WALK_ASM_CLOBBERS_DSPLIT, iv, ac, ks, voff, aargs, stride, Gs, ldstab);
                else KNG_WALK_ASM_LOOP(KNG_WALK_ASM_TEXT_FULL, KNG_WALK_ASM_CLOBBERS_FULL, iv, ac, ks, voff, aargs, stride, Gs, ldstab);
                k = ks;
                if (k >= Gs) break;
                // exact path: some lane of this wave needs the general arithmetic for kangaroo k; nothing of it was stored
                KNG_RARE_PATH();
                if (__lane_id() == 0) atomicAdd(a.dp_count + 1, 1u); // how often: kng_get_option("exact_exits")
                // the statement updates the running product in place: its value before kangaroo k is what kangaroo k - 1 stored
                fe xi = fe_from32(iv), xa = k ? ld_prod(a.s01, a.s23, slot(k - 1)) : fe_one();
                const size_t idx = slot(k);
                const bool have_nb = k + 1 < G;
                const fe nb = have_nb ? ld_prod(a.s01, a.s23, slot(k + 1)) : fe_one();
                const v16 cd = DSPLIT ? make_ulonglong2(ld_stream64(dlo + idx), 0) : ld_d(a.d, a.n_kang, idx);
                walk_core<DSPLIT>(a, tab, dlo, dhi, idx, ld_fe(a.x01, a.x23, idx), ld_fe(a.y01, a.y23, idx), cd, nb, have_nb, k == 0, false, xi, xa);
                fe_to32(iv, xi);
                fe_to32(ac, xa);
                if (++k >= Gs) break;
            }
            acc = fe_from32(ac);
            continue;
        }
#endif
        size_t idx = slot(0);
        fe cx = ld_fe(a.x01, a.x23, idx);
        fe cy = ld_fe(a.y01, a.y23, idx);
        v16 cd = DSPLIT ? make_ulonglong2(ld_stream64(dlo + idx), 0) : ld_d(a.d, a.n_kang, idx);
        fe nb = (G > 1) ? ld_prod(a.s01, a.s23, slot(1)) : fe_one();

        for (uint32_t k = 0; k < G; k++) {
            // ---- prefetch the next kangaroo and the product after it (before any store) ----
            fe nx = cx, ny = cy, nnb = nb;
            v16 nd = cd;
            size_t nidx = idx;
            if (k + 1 < G) {
                nidx = slot(k + 1);
                nx = ld_fe(a.x01, a.x23, nidx);
                ny = ld_fe(a.y01, a.y23, nidx);
                nd = DSPLIT ? make_ulonglong2(ld_stream64(dlo + nidx), 0) : ld_d(a.d, a.n_kang, nidx);
            }
            if (k + 2 < G) nnb = ld_prod(a.s01, a.s23, slot(k + 2));
            walk_core<DSPLIT>(a, tab, dlo, dhi, idx, cx, cy, cd, nb, k + 1 < G, k == 0, last, inv, acc);
            cx = nx;
            cy = ny;
            cd = nd;
            nb = nnb;
            idx = nidx;
        }
    }
}

// SHARE = 1: 256-thread blocks, every wave inverts for itself (option "share": a reference point and the form that needs
// no barrier).  SHARE = 8: 512-thread blocks, one inversion per CU.  (Round 2 also carried SHARE = 2, SHARE = 3 and two
// non-template twins of <1,.>: share 3 lost at every herd size, profiles/r02_group_share_sweep.txt; share 2 loses to 8.)
template <int SHARE, bool DSPLIT, bool ASM>
__global__ void __launch_bounds__(SHARE == 8 ? 512 : 256) __attribute__((amdgpu_waves_per_eu(2, 2))) kng_walk_share_kernel(const WalkArgs a) {
    __shared__ uint64_t tab[JT_WORDS];
    __shared__ v16 xch[SHARE == 8 ? 1024 : SHARE == 4 ? 1024 : 1]; // share 4: seven 64-lane slots of one field element (4 products, p01, p23, t)
    // share 4: the matrices the two waves of an inversion hand over (20 KB); the other forms get a word
    __shared__ typename std::conditional<SHARE == 4, InvRing, uint32_t>::type ring_mem;
    InvRing *const ring = reinterpret_cast<InvRing *>(&ring_mem);
    for (uint32_t i = threadIdx.x; i < JT_WORDS; i += blockDim.x) tab[i] = a.jtab[i];
    if (SHARE == 4 && threadIdx.x == 0) ring->progress = 0;
    __syncthreads();
    walk_body<SHARE, DSPLIT, ASM>(a, tab, xch, ring);
}

// MEASUREMENT ONLY (option "asm" 2, bench.py's roofline.alu_ceiling): kng_walk_share_kernel<8, true, true> with the global and
// LDS accesses of the per-kangaroo loop and the collection of its exactness flags left out of the scheduled statement
// (tools/gen_walk_asm.py VALU_ONLY) -- the same 1025 VALU instructions per kangaroo-jump on whatever the registers hold, the
// same inversion tree, the same grid: the rate the integer ALUs allow when memory costs neither cycles nor power (SURVEY 8d
// (ii)).  Results are WRONG on purpose and the herd is not usable afterwards.
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) kng_walk_valu_only_kernel(const WalkArgs a) {
    __shared__ uint64_t tab[JT_WORDS];
    __shared__ v16 xch[1024];
    for (uint32_t i = threadIdx.x; i < JT_WORDS; i += blockDim.x) tab[i] = a.jtab[i];
    __syncthreads();
    walk_body<8, true, true, true>(a, tab, xch);
}

// --------------------------------------------------------------------------------------------
// herd creation on the device (replaces Kangaroo::CreateHerd, Kangaroo.cpp:670-738,
// followed by SetKangaroos).  Kangaroo i gets a device distance dd uniform in [1, 2^range_power) from a
// counter-based generator and the point  B_type + dd*G - b*G  built with the SAME batched-inverse pass
// structure as the walk: one pass per 8-bit window of dd adds table[w][byte] = byte*256^w*G (a zero byte
// skips: dx := 1 keeps the lane's product chain intact), a last pass adds the constant -b*G.  Starting
// from the random offset points B_tame = b*G, B_wild = K - (N/2)*G + b*G (computed by the host library)
// keeps every addition generic (no doubling, no point at infinity).
struct HerdArgs {
    v16 *x01, *x23, *y01, *y23, *d, *s01, *s23;
    const uint64_t *table; // [windows][256][8]: x limbs 0..3, y limbs 0..3 ; entry 0 of each window unused
    uint64_t base[2][8];   // B_tame, B_wild
    uint64_t fin[8];       // -b*G
    uint64_t seed, n_kang;
    uint32_t windows, range_power, lanes;
};

KNG_DEV uint64_t herd_mix(uint64_t z) { // splitmix64 finaliser
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
KNG_DEV v16 herd_distance(uint64_t seed, uint64_t idx, uint32_t rp) {
    uint64_t lo = herd_mix(seed + 0x9E3779B97F4A7C15ULL * (2 * idx + 1));
    uint64_t hi = herd_mix(seed + 0x9E3779B97F4A7C15ULL * (2 * idx + 2));
    if (rp < 64) {
        lo &= (1ULL << rp) - 1;
        hi = 0;
    } else if (rp < 128) {
        hi &= (rp == 64) ? 0 : ((1ULL << (rp - 64)) - 1);
    }
    if ((lo | hi) == 0) lo = 1;
    return make_ulonglong2(lo, hi);
}
// addend of pass `step` for a kangaroo with distance d: table point of byte `step`, or the final constant
KNG_DEV bool herd_addend(const HerdArgs &a, uint32_t step, const v16 &d, fe &qx, fe &qy) {
    if (step >= a.windows) {
        qx = fe{{a.fin[0], a.fin[1], a.fin[2], a.fin[3]}};
        qy = fe{{a.fin[4], a.fin[5], a.fin[6], a.fin[7]}};
        return true;
    }
    const uint64_t word = step < 8 ? d.x : d.y;
    const uint32_t byte = (uint32_t)(word >> (8 * (step & 7))) & 0xFF;
    const v16 *e = reinterpret_cast<const v16 *>(a.table + ((size_t)step * 256 + byte) * 8);
    const v16 e0 = e[0], e1 = e[1], e2 = e[2], e3 = e[3];
    qx = fe{{e0.x, e0.y, e1.x, e1.y}};
    qy = fe{{e2.x, e2.y, e3.x, e3.y}};
    return byte != 0;
}

__global__ void __launch_bounds__(256) kng_herd_kernel(const HerdArgs a) {
    const size_t L = a.lanes;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L) return;
    const uint32_t G = (uint32_t)((a.n_kang - t + L - 1) / L);
    const uint32_t nsteps = a.windows + 1;

    // pass 0: distances, start points, products of the first window's dx
    fe acc;
    for (uint32_t g = 0; g < G; g++) {
        const size_t idx = (size_t)g * L + t;
        const v16 d = herd_distance(a.seed, idx, a.range_power);
        const uint64_t *b = a.base[idx & 1];
        const fe x{{b[0], b[1], b[2], b[3]}}, y{{b[4], b[5], b[6], b[7]}};
        st_d(a.d, a.n_kang, idx, d);
        st_fe(a.x01, a.x23, idx, x);
        st_fe(a.y01, a.y23, idx, y);
        fe qx, qy;
        const bool on = herd_addend(a, 0, d, qx, qy);
        const fe dx = on ? fe_sub(x, qx) : fe_one();
        acc = g ? fe_mul(acc, dx) : dx;
        st_fe(a.s01, a.s23, idx, acc);
    }
    for (uint32_t step = 0; step < nsteps; step++) {
        fe inv = fe_inv(acc);
        const bool backward = !(step & 1);
        const bool last = (step + 1 == nsteps);
        auto slot = [&](uint32_t k) -> size_t { return (size_t)(backward ? (G - 1 - k) : k) * L + t; };
        for (uint32_t k = 0; k < G; k++) {
            const size_t idx = slot(k);
            const fe cx = ld_fe(a.x01, a.x23, idx), cy = ld_fe(a.y01, a.y23, idx);
            const v16 d = ld_d(a.d, a.n_kang, idx);
            fe qx, qy;
            const bool on = herd_addend(a, step, d, qx, qy);
            const fe dx = on ? fe_sub(cx, qx) : fe_one();
            fe invk;
            if (k + 1 < G) {
                invk = fe_mul(inv, ld_fe(a.s01, a.s23, slot(k + 1)));
                inv = fe_mul(inv, dx);
            } else {
                invk = inv;
            }
            fe rx = cx, ry = cy;
            if (on) { // P + Q, affine (SECP256K1.cpp:238-263 formulas, operand order of GPUCompute.h:75-88)
                const fe s = fe_mul(fe_sub(cy, qy), invk);
                rx = fe_sub(fe_sub(fe_sqr(s), qx), cx);
                ry = fe_sub(fe_mul(fe_sub(cx, rx), s), cy);
                if (last) { // hand the walk canonical coordinates
                    rx = fe_canon(rx);
                    ry = fe_canon(ry);
                }
                st_fe(a.x01, a.x23, idx, rx);
                st_fe(a.y01, a.y23, idx, ry);
            }
            if (!last) {
                fe nqx, nqy;
                const bool non = herd_addend(a, step + 1, d, nqx, nqy);
                const fe dx2 = non ? fe_sub(rx, nqx) : fe_one();
                acc = k ? fe_mul(acc, dx2) : dx2;
                st_fe(a.s01, a.s23, idx, acc);
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// work-file snapshot.  The reference saves a herd by parking the GPU thread, converting the whole device state
// into 3 x N host `Int` (GPUEngine::GetKangaroos, GPUEngine.cu:443-500: a mod-n subtraction per wild kangaroo, serial) and
// writing it with three 32-byte fwrite calls per kangaroo (Backup.cpp:525-546) while every GPU idles.  Here one kernel packs
// the SoA planes into the work file's own byte layout -- 96-byte records {x[4], y[4], d[4]}, d = TRUE distance mod n,
// i.e. the wild offset already removed -- in a second device buffer (96 B x herd: 805 MB of 288 GB at the default herd), stream-
// ordered between two launches: the walk goes on at once, and a saver thread copies the frozen records out in large
// pieces on its own stream.  The inverse kernel turns records uploaded from a work file back into planes (-i,
// Backup.cpp:211-231 + GPUEngine::SetKangaroos, GPUEngine.cu:381-441).
struct SnapArgs {
    v16 *x01, *x23, *y01, *y23;
    uint64_t *dlo, *dhi;
    uint64_t *rec; // 12 words per kangaroo
    uint64_t first, count;
    uint64_t woff[4];           // wild offset, reduced mod n; all zero = records carry device distances
    unsigned long long *status; // unpack: [0] = records whose device distance does not fit 128 bits, [1] = first such index + 1
};
__device__ static const uint64_t KNG_ORDER_N[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL};

__global__ void __launch_bounds__(256) kng_snapshot_pack_kernel(const SnapArgs a) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.count) return;
    const uint64_t i = a.first + r;
    const v16 x0 = a.x01[i], x1 = a.x23[i], y0 = a.y01[i], y1 = a.y23[i];
    uint64_t d[4] = {a.dlo[i], a.dhi[i], 0, 0};
    if (i & 1) { // wild: (dd - offset) mod n, what ModSubK1order leaves (GPUEngine.cu:477)
        uint64_t t[4], borrow = 0;
        for (int k = 0; k < 4; k++) {
            const unsigned __int128 v = (unsigned __int128)d[k] - a.woff[k] - borrow;
            t[k] = (uint64_t)v;
            borrow = (uint64_t)(v >> 64) & 1;
        }
        if (borrow) {
            uint64_t carry = 0;
            for (int k = 0; k < 4; k++) {
                const unsigned __int128 v = (unsigned __int128)t[k] + KNG_ORDER_N[k] + carry;
                t[k] = (uint64_t)v;
                carry = (uint64_t)(v >> 64);
            }
        }
        for (int k = 0; k < 4; k++) d[k] = t[k];
    }
    v16 *o = reinterpret_cast<v16 *>(a.rec + 12 * i);
    o[0] = x0; o[1] = x1; o[2] = y0; o[3] = y1;
    o[4] = make_ulonglong2(d[0], d[1]);
    o[5] = make_ulonglong2(d[2], d[3]);
}

__global__ void __launch_bounds__(256) kng_snapshot_unpack_kernel(const SnapArgs a) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= a.count) return;
    const uint64_t i = a.first + r;
    const v16 *o = reinterpret_cast<const v16 *>(a.rec + 12 * i);
    const v16 d01 = o[4], d23 = o[5];
    uint64_t d[4] = {d01.x, d01.y, d23.x, d23.y};
    if (i & 1) { // wild: (d + offset) mod n (ModAddK1order, GPUEngine.cu:406-409)
        uint64_t t[4], carry = 0;
        for (int k = 0; k < 4; k++) {
            const unsigned __int128 v = (unsigned __int128)d[k] + a.woff[k] + carry;
            t[k] = (uint64_t)v;
            carry = (uint64_t)(v >> 64);
        }
        uint64_t u[4], borrow = 0;
        for (int k = 0; k < 4; k++) {
            const unsigned __int128 v = (unsigned __int128)t[k] - KNG_ORDER_N[k] - borrow;
            u[k] = (uint64_t)v;
            borrow = (uint64_t)(v >> 64) & 1;
        }
        const bool ge = carry || !borrow; // t >= n
        for (int k = 0; k < 4; k++) d[k] = ge ? u[k] : t[k];
    }
    if (d[2] | d[3]) { // the reference truncates silently (GPUEngine.cu:410-411); such a kangaroo could never be right again
        atomicAdd(&a.status[0], 1ULL);
        atomicMin(&a.status[1], (unsigned long long)i);
    }
    a.x01[i] = o[0]; a.x23[i] = o[1]; a.y01[i] = o[2]; a.y23[i] = o[3];
    a.dlo[i] = d[0];
    a.dhi[i] = d[1];
}

// --------------------------------------------------------------------------------------------
// whole-run audit on the device (new; the device-side counterpart of the reference's -wcheck, Check.cpp:141-411, which
// re-derives every stored distinguished point from its distance, and of Kangaroo::Output's final check, Kangaroo.cpp:196-206).
// A walk error is permanent for its kangaroo: (x, y) = d*G (tame) / K + d*G (wild) holds after every exact jump and never
// again after an inexact one.  The audit recomputes that point from the 128-bit device distance alone -- with the herd
// builder's machinery (16 windows of 8 bits + the closing constant, one batched inversion per lane and window; general
// arithmetic only, nothing of the scheduled loop's short forms) -- and compares it with what the walk left:
//   HERD mode  every kangaroo of the engine's herd, x AND y (256 bits each, canonical residues compared);
//   RECS mode  an array of 64-byte DP records {x, d, kidx, mode}: mode 0 compares all of x, mode 1 only what a hash-table
//              entry keeps of it (limbs 0-1 and the 18 bucket bits of limb 2, HashTable.h:27-56).
// A distance of zero has no affine point (the offset form would add b*G and -b*G): it is reported as a mismatch without
// entering the lane's product chain.
struct AuditArgs {
    HerdArgs h;                            // x01..y23, s01, s23 = SCRATCH planes of h.n_kang vectors; table / base / fin / windows / lanes as for the builder
    const v16 *ex01, *ex23, *ey01, *ey23;  // HERD: the herd's own planes
    const uint64_t *dlo, *dhi;             // HERD: its distance words
    const DpRecord *recs;                  // RECS
    unsigned long long *result;            // [0] mismatches, [1] indices recorded, [2 .. 2+cap) the first mismatching indices
    uint32_t cap;
};

template <bool RECS>
KNG_DEV v16 audit_distance(const AuditArgs &a, size_t idx) {
    if (RECS) return *reinterpret_cast<const v16 *>(a.recs[idx].d);
    return make_ulonglong2(a.dlo[idx], a.dhi[idx]);
}
KNG_DEV void audit_report(const AuditArgs &a, size_t idx) {
    atomicAdd(a.result, 1ULL);
    const unsigned long long slot = atomicAdd(a.result + 1, 1ULL);
    if (slot < a.cap) a.result[2 + slot] = (unsigned long long)idx;
}

template <bool RECS>
__global__ void __launch_bounds__(256) kng_audit_kernel(const AuditArgs a) {
    const HerdArgs &h = a.h;
    const size_t L = h.lanes;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= L || t >= h.n_kang) return;
    const uint32_t G = (uint32_t)((h.n_kang - t + L - 1) / L);
    const uint32_t nsteps = h.windows + 1;

    // pass 0: start points, products of the first window's dx
    fe acc = fe_one();
    for (uint32_t g = 0; g < G; g++) {
        const size_t idx = (size_t)g * L + t;
        const v16 d = audit_distance<RECS>(a, idx);
        const uint64_t *b = h.base[(RECS ? a.recs[idx].kidx : (uint64_t)idx) & 1];
        const fe x{{b[0], b[1], b[2], b[3]}}, y{{b[4], b[5], b[6], b[7]}};
        st_fe(h.x01, h.x23, idx, x);
        st_fe(h.y01, h.y23, idx, y);
        fe qx, qy;
        const bool on = herd_addend(h, 0, d, qx, qy) && (d.x | d.y) != 0;
        const fe dx = on ? fe_sub(x, qx) : fe_one();
        acc = g ? fe_mul(acc, dx) : dx;
        st_fe(h.s01, h.s23, idx, acc);
    }
    for (uint32_t step = 0; step < nsteps; step++) {
        fe inv = fe_inv(acc);
        const bool backward = !(step & 1);
        const bool last = (step + 1 == nsteps);
        auto slot = [&](uint32_t k) -> size_t { return (size_t)(backward ? (G - 1 - k) : k) * L + t; };
        for (uint32_t k = 0; k < G; k++) {
            const size_t idx = slot(k);
            const fe cx = ld_fe(h.x01, h.x23, idx), cy = ld_fe(h.y01, h.y23, idx);
            const v16 d = audit_distance<RECS>(a, idx);
            const bool live = (d.x | d.y) != 0;
            fe qx, qy;
            const bool on = herd_addend(h, step, d, qx, qy) && live;
            const fe dx = on ? fe_sub(cx, qx) : fe_one();
            fe invk;
            if (k + 1 < G) {
                invk = fe_mul(inv, ld_fe(h.s01, h.s23, slot(k + 1)));
                inv = fe_mul(inv, dx);
            } else {
                invk = inv;
            }
            fe rx = cx, ry = cy;
            if (on) { // P + Q, affine: the builder's formulas
                const fe s = fe_mul(fe_sub(cy, qy), invk);
                rx = fe_sub(fe_sub(fe_sqr(s), qx), cx);
                ry = fe_sub(fe_mul(fe_sub(cx, rx), s), cy);
                if (!last) {
                    st_fe(h.x01, h.x23, idx, rx);
                    st_fe(h.y01, h.y23, idx, ry);
                }
            }
            if (last) {
                rx = fe_canon(rx);
                ry = fe_canon(ry);
                bool ok = live;
                if (RECS) {
                    const DpRecord &r = a.recs[idx];
                    if (r.pad == 0) {
                        const fe ex = fe_canon(fe{{r.x[0], r.x[1], r.x[2], r.x[3]}});
                        ok = ok && ex.v[0] == rx.v[0] && ex.v[1] == rx.v[1] && ex.v[2] == rx.v[2] && ex.v[3] == rx.v[3];
                    } else { // what a hash-table entry keeps: x limbs 0-1 and the bucket bits of limb 2
                        ok = ok && r.x[0] == rx.v[0] && r.x[1] == rx.v[1] && ((r.x[2] ^ rx.v[2]) & 0x3FFFFULL) == 0;
                    }
                } else {
                    const fe ex = fe_canon(ld_fe(a.ex01, a.ex23, idx)), ey = fe_canon(ld_fe(a.ey01, a.ey23, idx));
                    for (int i = 0; i < 4; i++) ok = ok && ex.v[i] == rx.v[i] && ey.v[i] == ry.v[i];
                }
                if (!ok) audit_report(a, idx);
            } else {
                fe nqx, nqy;
                const bool non = herd_addend(h, step + 1, d, nqx, nqy) && live;
                const fe dx2 = non ? fe_sub(rx, nqx) : fe_one();
                acc = k ? fe_mul(acc, dx2) : dx2;
                st_fe(h.s01, h.s23, idx, acc);
            }
        }
    }
}

// overwrite one kangaroo, stream-ordered (replaces the ten 8-byte copies of GPUEngine.cu:504-530)
__global__ void kng_patch_kernel(v16 *x01, v16 *x23, v16 *y01, v16 *y23, v16 *d, uint64_t n, uint64_t idx, fe x,
                                 fe y, v16 dd) {
    st_fe(x01, x23, idx, x);
    st_fe(y01, y23, idx, y);
    st_d(d, n, idx, dd);
}

// device self-test of the primitives (replaces the compiled-out check_gpu, GPUEngine.cu:43-92)
__global__ void __launch_bounds__(64) kng_fieldop_kernel(int op, const uint64_t *a, const uint64_t *b, uint64_t *r, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const fe x{{a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]}};
    const fe y{{b[4 * i], b[4 * i + 1], b[4 * i + 2], b[4 * i + 3]}};
    fe z;
    switch (op) {
    case KNG_OP_MODMUL: z = fe_mul(x, y); break;
    case KNG_OP_MODSQR: z = fe_sqr(x); break;
    case KNG_OP_MODSUB: z = fe_sub(x, y); break;
    case KNG_OP_MODINV: z = fe_inv(x); break;
    default: z = fe_zero(); break;
    }
    for (int k = 0; k < 4; k++) r[4 * i + k] = z.v[k];
}

// --------------------------------------------------------------------------------------------
// host side: the C ABI
// --------------------------------------------------------------------------------------------

static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(KNG_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));         \
    } while (0)

// temporary device allocation released on every exit path of the function that owns it
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes); }
    template <typename T>
    T *as() const { return static_cast<T *>(p); }
};

struct kng_engine {
    int dev = 0;
    int grid_x = 0, grid_y = 0;
    uint64_t n = 0; // kangaroos
    uint32_t max_found = 0;
    // tuning
    uint32_t group = 0; // kangaroos per lane
    uint32_t block = 256;
    uint32_t nsteps = KNG_NB_RUN;
    uint32_t lanes = 0;
    int cu_count = 0;
    // device memory
    int dsplit = -1;       // distance plane: -1 = low-word streaming when every jump distance < 2^50, 0 = never, 1 = whenever the table allows (high words all zero)
    bool dsplit_on = false; // decided by kng_set_params / the option
    uint64_t jd_max = 0;    // largest low word of the jump distances, UINT64_MAX when a high word is set
    int share = -1;        // waves that share one inversion per jump: 8 (512-thread blocks, one inversion per CU), 4 (256-thread blocks), -1 = by herd size (kng_launch)
    int share_used = 8;    // what the last launch ran with
    int use_asm = 1;       // the scheduled asm loop (kng_walk_asm.h) instead of the compiler-scheduled one; herds beyond 2^28 cannot
    WalkAsmArgs *asm_args = nullptr; // device: one block per DP buffer
    v16 *planes = nullptr; // 7 planes of n v16
    uint64_t *jtab = nullptr;
    uint32_t *dp_count[2] = {nullptr, nullptr};
    DpRecord *dp_items[2] = {nullptr, nullptr};
    // pinned host memory
    uint32_t *h_count[2] = {nullptr, nullptr};
    DpRecord *h_items = nullptr;
    // option "dp_ring": the kernel writes its DP records straight into pinned, device-mapped host memory (one buffer per
    // launch slot; the counter stays in device memory -- an atomic per DP-bearing wave-step across PCIe would stall the
    // walk -- and lands last, stream-ordered behind the kernel).  No second hop, no host-synchronous copy in land_points.
    int dp_ring = 1; // default since round 3: kernel time -0.8 % at DP 14, -0.4 % at DP 11, no host-synchronous copy (profiles/r03_ab_dp_ring.txt)
    DpRecord *ring[2] = {nullptr, nullptr};     // host view
    DpRecord *ring_dev[2] = {nullptr, nullptr}; // device view of the same memory
    const DpRecord *view = nullptr;             // where the records of the last drained launch are
    v16 *h_stage = nullptr;
    size_t stage_kang = 0;
    // streams / events
    hipStream_t walk = nullptr, copy = nullptr;
    hipEvent_t ev_start[2] = {nullptr, nullptr}, ev_stop[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    // state
    uint64_t dp_mask = 0;
    bool have_params = false, have_herd = false;
    bool products_valid = false; // the S planes hold the prefix products the next launch needs (left by the previous launch, nothing touched since)
    bool outstanding = false; // launched, not yet waited
    int slot_next = 0;        // DP buffer the next launch writes
    int slot_ready = -1;      // DP buffer of the most recently waited launch
    float last_ms = 0.f;
    uint32_t last_exact_exits = 0; // wave-iterations the scheduled loop handed to the general arithmetic in the last waited launch
    bool lost_warned = false;
    uint64_t bytes = 0;
    // whole-run audit (kng_audit_*): window table + offset points on the device, result block (device + pinned)
    uint64_t *audit_tab = nullptr;
    uint64_t audit_base[2][8] = {{0}}, audit_fin[8] = {0};
    unsigned long long *audit_res = nullptr, *audit_res_host = nullptr;
    bool audit_ready = false;
    float last_audit_ms = 0.f;
    WalkAsmArgs *asm_args_host = nullptr; // pinned staging of the loop constants: uploaded stream-ordered (kng_set_params)
    // work-file snapshot (kng_snapshot*): 96-byte records of the whole herd, its own stream (a saver thread reads while the walk runs)
    uint64_t *snap = nullptr;
    hipStream_t snap_stream = nullptr;
    hipEvent_t snap_ev = nullptr;
    unsigned long long *snap_status = nullptr, *snap_status_dev = nullptr; // pinned / device: {count, first index} of distances that do not fit
    bool snap_taken = false;
};

static uint64_t device_bytes(const kng_engine *h); // GetMemory()
static inline v16 *plane(const kng_engine *h, int k) { return h->planes + (size_t)k * h->n; }
// plane 4 = distances, stored as the N low words followed by the N high words
static inline uint64_t *dplane(const kng_engine *h, int hi) { return reinterpret_cast<uint64_t *>(plane(h, 4)) + (hi ? h->n : 0); }
// low-word streaming of the distances (walk_body DSPLIT): needs every high word of the table zero; automatic
// only when a carry out of the low word is rare enough for its read-modify-write not to matter
static void decide_dsplit(kng_engine *h) {
    const bool possible = h->jd_max != UINT64_MAX;
    h->dsplit_on = possible && (h->dsplit == 1 || (h->dsplit == -1 && h->jd_max < (1ULL << (h->use_asm ? 58 : 50))));
}
static DpRecord *dp_buffer(const kng_engine *h, int s) { return h->dp_ring ? h->ring_dev[s] : h->dp_items[s]; }
// constants of the scheduled loop (WalkAsmArgs), one block per DP buffer.  Uploaded from a pinned block with a copy on the
// WALK stream: ordered behind a launch in flight (whose loop re-reads them at every entry) and ahead of the next one --
// the reference's cudaMemcpyToSymbol blocks behind its kernel the same way (GPUEngine.cu:565-583).  The caller
// synchronises the stream before the pinned block is reused.
static int upload_loop_args(kng_engine *h) {
    WalkAsmArgs *aa = h->asm_args_host;
    for (int s = 0; s < 2; s++) {
        aa[s].x01 = (uint64_t)plane(h, 0); aa[s].x23 = (uint64_t)plane(h, 1); aa[s].y01 = (uint64_t)plane(h, 2); aa[s].y23 = (uint64_t)plane(h, 3);
        aa[s].dlo = (uint64_t)dplane(h, 0); aa[s].dhi = (uint64_t)dplane(h, 1);
        aa[s].s01 = (uint64_t)plane(h, 5); aa[s].s23 = (uint64_t)plane(h, 6);
        aa[s].dp_mask = h->dp_mask; aa[s].dp_count = (uint64_t)h->dp_count[s]; aa[s].dp_items = (uint64_t)dp_buffer(h, s);
        aa[s].max_found = h->max_found; aa[s].pad = 0;
    }
    HIP_TRY(hipMemcpyAsync(h->asm_args, aa, 2 * sizeof(WalkAsmArgs), hipMemcpyHostToDevice, h->walk));
    HIP_TRY(hipStreamSynchronize(h->walk));
    return KNG_OK;
}

// DP landing buffers by mode (option "dp_ring"): the pinned rings the kernel writes directly, or the device buffers +
// the pinned landing buffer of the copy path -- never both (a solver may ask for 4e8 slots = 25.6 GB per buffer)
static int alloc_dp_buffers(kng_engine *h) {
    const size_t bytes = (size_t)h->max_found * sizeof(DpRecord);
    hipError_t e;
    if (h->dp_ring) {
        for (int s = 0; s < 2; s++) {
            if (h->ring[s]) continue;
            if ((e = hipHostMalloc((void **)&h->ring[s], bytes, hipHostMallocMapped | hipHostMallocPortable)) != hipSuccess)
                return fail(KNG_E_ALLOC, "pinned DP ring (%zu bytes): %s", bytes, hipGetErrorString(e));
            if ((e = hipHostGetDevicePointer((void **)&h->ring_dev[s], h->ring[s], 0)) != hipSuccess)
                return fail(KNG_E_HIP, "device view of the DP ring: %s", hipGetErrorString(e));
        }
    } else {
        for (int s = 0; s < 2; s++) {
            if (h->dp_items[s]) continue;
            if ((e = hipMalloc((void **)&h->dp_items[s], bytes)) != hipSuccess) return fail(KNG_E_ALLOC, "dp items (%zu bytes): %s", bytes, hipGetErrorString(e));
        }
        if (!h->h_items && (e = hipHostMalloc((void **)&h->h_items, bytes, hipHostMallocDefault)) != hipSuccess)
            return fail(KNG_E_ALLOC, "pinned dp items (%zu bytes): %s", bytes, hipGetErrorString(e));
    }
    return KNG_OK;
}
static void free_dp_buffers(kng_engine *h, bool rings) {
    if (rings) {
        for (int s = 0; s < 2; s++) {
            if (h->ring[s]) (void)hipHostFree(h->ring[s]);
            h->ring[s] = h->ring_dev[s] = nullptr;
        }
    } else {
        for (int s = 0; s < 2; s++) {
            if (h->dp_items[s]) (void)hipFree(h->dp_items[s]);
            h->dp_items[s] = nullptr;
        }
        if (h->h_items) (void)hipHostFree(h->h_items);
        h->h_items = nullptr;
    }
}

#define KNG_AUDIT_CAP 1024 // mismatching indices kept per audit launch
// one audit launch over `n` items on `stream`; scratch = 6 planes of n vectors (x', y', products)
template <bool RECS>
static int run_audit(kng_engine *h, hipStream_t stream, AuditArgs &a, v16 *scratch, v16 *s01, v16 *s23, uint64_t n, uint32_t lanes, uint64_t *n_bad,
                     uint64_t *bad_idx, uint32_t bad_cap, uint64_t idx_offset, uint32_t *recorded) {
    a.h.x01 = scratch; a.h.x23 = scratch + n; a.h.y01 = scratch + 2 * n; a.h.y23 = scratch + 3 * n;
    a.h.d = nullptr; a.h.s01 = s01; a.h.s23 = s23;
    a.h.table = h->audit_tab;
    memcpy(a.h.base, h->audit_base, sizeof a.h.base);
    memcpy(a.h.fin, h->audit_fin, sizeof a.h.fin);
    a.h.seed = 0; a.h.n_kang = n; a.h.windows = KNG_AUDIT_WINDOWS; a.h.range_power = 128; a.h.lanes = lanes;
    a.result = h->audit_res; a.cap = KNG_AUDIT_CAP;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return fail(KNG_E_HIP, "event creation failed"); }
    int rc = KNG_OK;
    do {
        hipError_t e;
        if ((e = hipMemsetAsync(h->audit_res, 0, (2 + KNG_AUDIT_CAP) * 8, stream)) != hipSuccess || (e = hipEventRecord(e0, stream)) != hipSuccess) { rc = fail(KNG_E_HIP, "audit: %s", hipGetErrorString(e)); break; }
        hipLaunchKernelGGL((kng_audit_kernel<RECS>), dim3((lanes + 255) / 256), dim3(256), 0, stream, a);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipEventRecord(e1, stream)) != hipSuccess ||
            (e = hipMemcpyAsync(h->audit_res_host, h->audit_res, (2 + KNG_AUDIT_CAP) * 8, hipMemcpyDeviceToHost, stream)) != hipSuccess ||
            (e = hipStreamSynchronize(stream)) != hipSuccess) { rc = fail(KNG_E_HIP, "audit kernel: %s", hipGetErrorString(e)); break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        h->last_audit_ms += ms;
        *n_bad += h->audit_res_host[0];
        const uint64_t got = h->audit_res_host[1] < KNG_AUDIT_CAP ? h->audit_res_host[1] : KNG_AUDIT_CAP;
        for (uint64_t i = 0; i < got && bad_idx && *recorded < bad_cap; i++) bad_idx[(*recorded)++] = idx_offset + h->audit_res_host[2 + i];
    } while (0);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
}

extern "C" {

const char *kng_last_error(void) { return g_err.c_str(); }
const char *kng_version(void) { return "kangaroo_hip 0.2 (gfx950)"; }

int kng_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int kng_device_info(int dev, char *name, size_t name_cap, int *cu_count, uint64_t *mem_bytes, char *arch,
                    size_t arch_cap) {
    if (dev < 0 || dev >= kng_device_count()) return fail(KNG_E_NODEVICE, "invalid device %d", dev);
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));
    // the marketing name comes from libdrm's amdgpu.ids and is empty on boxes that lack the file
    if (name && name_cap) snprintf(name, name_cap, "%s", p.name[0] ? p.name : p.gcnArchName);
    if (arch && arch_cap) snprintf(arch, arch_cap, "%s", p.gcnArchName);
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (mem_bytes) *mem_bytes = (uint64_t)p.totalGlobalMem;
    return KNG_OK;
}

// sysfs numa_node of a PCI address ("0000:C3:00.0", any case), -1 when unknown.  No device needed: the parsing is tested
// against a made-up tree (KNG_SYSFS_ROOT replaces "/sys").
int kng_numa_node_of_bdf(const char *bdf_in) {
    if (!bdf_in || !*bdf_in || strlen(bdf_in) >= 64) return -1;
    char bdf[64];
    snprintf(bdf, sizeof bdf, "%s", bdf_in);
    for (char *c = bdf; *c; c++)
        if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a'); // sysfs spells the address in lower case
    const char *root = getenv("KNG_SYSFS_ROOT");
    char path[4200];
    snprintf(path, sizeof path, "%s/bus/pci/devices/%s/numa_node", root && *root ? root : "/sys", bdf);
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1 || node < 0) node = -1; // "-1" = the platform does not say
    fclose(f);
    return node;
}

int kng_device_pci_bdf(int dev, char *bdf, size_t cap) {
    if (!bdf || cap < 13) return fail(KNG_E_ARG, "bdf buffer too small");
    if (dev < 0 || dev >= kng_device_count()) return fail(KNG_E_NODEVICE, "invalid device %d", dev);
    HIP_TRY(hipDeviceGetPCIBusId(bdf, (int)cap, dev));
    for (char *c = bdf; *c; c++)
        if (*c >= 'A' && *c <= 'F') *c = (char)(*c - 'A' + 'a');
    return KNG_OK;
}

int kng_device_numa_node(int dev) {
    char bdf[64] = {0};
    if (kng_device_pci_bdf(dev, bdf, sizeof bdf) != KNG_OK) return -1;
    return kng_numa_node_of_bdf(bdf);
}

int kng_device_free_bytes(int dev, uint64_t *free_bytes, uint64_t *total_bytes) {
    if (dev < 0 || dev >= kng_device_count()) return fail(KNG_E_NODEVICE, "invalid device %d", dev);
    HIP_TRY(hipSetDevice(dev));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = (uint64_t)f;
    if (total_bytes) *total_bytes = (uint64_t)t;
    return KNG_OK;
}

int kng_default_grid(int dev, int *x, int *y) {
    if (!x || !y) return fail(KNG_E_ARG, "null grid pointer");
    if (*x <= 0 || *y <= 0) {
        int cu = 0;
        int rc = kng_device_info(dev, nullptr, 0, &cu, nullptr, nullptr, 0);
        if (rc) return rc;
        if (*x <= 0) *x = 2 * cu; // GPUEngine.cu:301
        if (*y <= 0) *y = 128;    // GPUEngine.cu:302-303 (no per-SM core count for this arch)
    }
    return KNG_OK;
}

static void choose_geometry(kng_engine *h) {
    // lanes L = n / group.  The per-lane batch amortises one modular inversion, so bigger is
    // cheaper -- but one wave can only issue a VALU instruction every ~5 cycles on gfx950
    // (profiles/r01_clock_probe.txt), so every SIMD needs at least two resident waves to be
    // kept busy.  Default: the largest group that still leaves >= 2 waves per SIMD; herds too small to
    // fill the chip shrink the batch down to 1 -- spreading over idle SIMDs beats amortising the inversion
    // (2^16 kangaroos: 0.67 -> 2.2 GK/s, 2^20: 10.5 -> 15.0 GK/s, profiles/r01_herd_size_sweep.txt).
    if (h->group == 0) {
        const uint64_t want_lanes = (uint64_t)h->cu_count * 4 * 64 * 2;
        uint32_t g = 4 * KNG_GRP_SIZE; // herds beyond 2^24 keep 2 waves/SIMD with even longer batches
        while (g > 1 && h->n / g < want_lanes) g >>= 1;
        h->group = g;
    }
    while (h->group > 1 && (h->n % h->group)) h->group >>= 1;
    h->lanes = (uint32_t)(h->n / h->group);
}

int kng_create(int dev, int grid_x, int grid_y, uint32_t max_found, kng_engine **out) {
    if (!out) return fail(KNG_E_ARG, "null out");
    *out = nullptr;
    if (grid_x <= 0 || grid_y <= 0 || max_found == 0) return fail(KNG_E_ARG, "bad grid %dx%d / max_found %u", grid_x, grid_y, max_found);
    const int ndev = kng_device_count();
    if (ndev == 0) return fail(KNG_E_NODEVICE, "no HIP device available (this engine has no CPU fallback)");
    if (dev < 0 || dev >= ndev) return fail(KNG_E_NODEVICE, "invalid device %d (have %d)", dev, ndev);
    HIP_TRY(hipSetDevice(dev));
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, dev));

    kng_engine *h = new kng_engine();
    h->dev = dev;
    h->grid_x = grid_x;
    h->grid_y = grid_y;
    h->n = (uint64_t)grid_x * (uint64_t)grid_y * KNG_GRP_SIZE;
    h->max_found = max_found;
    h->cu_count = p.multiProcessorCount;
    choose_geometry(h);

    auto bail = [&](int code) {
        kng_destroy(h);
        (void)hipGetLastError(); // a failed hipMalloc leaves a sticky error that the next kernel launch would report as its own
        return code;
    };
    hipError_t e;
    const size_t plane_bytes = (size_t)h->n * sizeof(v16);
    const size_t state_bytes = 7 * plane_bytes;
    if ((e = hipMalloc((void **)&h->planes, state_bytes)) != hipSuccess)
        return bail(fail(KNG_E_ALLOC, "herd state (%zu bytes): %s", state_bytes, hipGetErrorString(e)));
    if ((e = hipMalloc((void **)&h->jtab, JT_WORDS * 8)) != hipSuccess) return bail(fail(KNG_E_ALLOC, "jump table: %s", hipGetErrorString(e)));
    if ((e = hipMalloc((void **)&h->asm_args, 2 * sizeof(WalkAsmArgs))) != hipSuccess) return bail(fail(KNG_E_ALLOC, "loop arguments: %s", hipGetErrorString(e)));
    // the scheduled loop addresses a plane as base + 32-bit byte offset and a DP record as base + 32-bit offset
    if (h->n > (1ull << 28) || max_found > (1u << 26)) h->use_asm = 0;
    if ((e = hipHostMalloc((void **)&h->asm_args_host, 2 * sizeof(WalkAsmArgs), hipHostMallocDefault)) != hipSuccess)
        return bail(fail(KNG_E_ALLOC, "pinned loop arguments: %s", hipGetErrorString(e)));
    for (int s = 0; s < 2; s++) {
        if ((e = hipMalloc((void **)&h->dp_count[s], 64)) != hipSuccess) return bail(fail(KNG_E_ALLOC, "dp counter: %s", hipGetErrorString(e)));
        // zero at allocation: the reference reads an uninitialised counter on its first Launch
        if ((e = hipMemset(h->dp_count[s], 0, 64)) != hipSuccess) return bail(fail(KNG_E_HIP, "memset: %s", hipGetErrorString(e)));
        if ((e = hipHostMalloc((void **)&h->h_count[s], 64, hipHostMallocDefault)) != hipSuccess)
            return bail(fail(KNG_E_ALLOC, "pinned counter: %s", hipGetErrorString(e)));
        *h->h_count[s] = 0;
        if (hipEventCreate(&h->ev_start[s]) != hipSuccess || hipEventCreate(&h->ev_stop[s]) != hipSuccess ||
            hipEventCreateWithFlags(&h->ev_done[s], hipEventBlockingSync) != hipSuccess)
            return bail(fail(KNG_E_HIP, "event creation failed"));
    }
    // DP landing buffers of the mode in use only (default: the pinned rings); the other kind when "dp_ring" is switched
    if (int rc = alloc_dp_buffers(h)) return bail(rc);
    h->stage_kang = h->n < (1u << 16) ? (size_t)h->n : (size_t)(1u << 16);
    if ((e = hipHostMalloc((void **)&h->h_stage, 6 * h->stage_kang * sizeof(v16), hipHostMallocDefault)) != hipSuccess)
        return bail(fail(KNG_E_ALLOC, "pinned staging: %s", hipGetErrorString(e)));
    if (hipStreamCreateWithFlags(&h->walk, hipStreamNonBlocking) != hipSuccess ||
        hipStreamCreateWithFlags(&h->copy, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(KNG_E_HIP, "stream creation failed"));
    h->bytes = state_bytes + JT_WORDS * 8 + 2 * 64 + (h->dp_ring ? 0 : 2 * (uint64_t)max_found * sizeof(DpRecord)); // device memory (GetMemory())
    *out = h;
    return KNG_OK;
}

void kng_destroy(kng_engine *h) {
    if (!h) return;
    (void)hipSetDevice(h->dev);
    if (h->walk) (void)hipStreamSynchronize(h->walk);
    if (h->copy) (void)hipStreamSynchronize(h->copy);
    if (h->planes) (void)hipFree(h->planes);
    if (h->jtab) (void)hipFree(h->jtab);
    if (h->asm_args) (void)hipFree(h->asm_args);
    for (int s = 0; s < 2; s++) {
        if (h->dp_count[s]) (void)hipFree(h->dp_count[s]);
        if (h->h_count[s]) (void)hipHostFree(h->h_count[s]);
        if (h->ev_start[s]) (void)hipEventDestroy(h->ev_start[s]);
        if (h->ev_stop[s]) (void)hipEventDestroy(h->ev_stop[s]);
        if (h->ev_done[s]) (void)hipEventDestroy(h->ev_done[s]);
    }
    free_dp_buffers(h, true);
    free_dp_buffers(h, false);
    if (h->asm_args_host) (void)hipHostFree(h->asm_args_host);
    if (h->audit_tab) (void)hipFree(h->audit_tab);
    if (h->audit_res) (void)hipFree(h->audit_res);
    if (h->audit_res_host) (void)hipHostFree(h->audit_res_host);
    if (h->h_stage) (void)hipHostFree(h->h_stage);
    if (h->snap_stream) { (void)hipStreamSynchronize(h->snap_stream); (void)hipStreamDestroy(h->snap_stream); }
    if (h->snap_ev) (void)hipEventDestroy(h->snap_ev);
    if (h->snap) (void)hipFree(h->snap);
    if (h->snap_status) (void)hipHostFree(h->snap_status);
    if (h->snap_status_dev) (void)hipFree(h->snap_status_dev);
    if (h->walk) (void)hipStreamDestroy(h->walk);
    if (h->copy) (void)hipStreamDestroy(h->copy);
    delete h;
}

uint64_t kng_nb_kangaroos(const kng_engine *h) { return h ? h->n : 0; }
uint64_t kng_memory_bytes(const kng_engine *h) { return h ? h->bytes : 0; }

int kng_set_option(kng_engine *h, const char *key, int64_t value) {
    if (!h || !key) return fail(KNG_E_ARG, "null argument");
    if (h->outstanding) return fail(KNG_E_STATE, "cannot change options while a launch is outstanding");
    h->products_valid = false; // geometry, layout or loop may change: the next launch recomputes its products
    std::string k(key);
    if (k == "group") {
        if (value < 1 || (value & (value - 1)) || (h->n % (uint64_t)value)) return fail(KNG_E_ARG, "group must be a power of two dividing the herd");
        h->group = (uint32_t)value;
        h->lanes = (uint32_t)(h->n / h->group);
    } else if (k == "lanes") {
        // free choice of the lane count (multiple of 64): groups become ragged, ceil/floor(N/lanes) per wave
        if (value < 64 || (value % 64) || (uint64_t)value > h->n) return fail(KNG_E_ARG, "lanes must be a multiple of 64 and <= herd size");
        h->lanes = (uint32_t)value;
        h->group = (uint32_t)((h->n + (uint64_t)value - 1) / (uint64_t)value);
    } else if (k == "block") {
        if (value < 64 || value > 256 || (value % 64)) return fail(KNG_E_ARG, "block must be 64,128,192 or 256");
        h->block = (uint32_t)value;
    } else if (k == "steps") {
        if (value < 1 || value > 1 << 20) return fail(KNG_E_ARG, "steps out of range");
        h->nsteps = (uint32_t)value;
    } else if (k == "dsplit") {
        if (value < -1 || value > 1) return fail(KNG_E_ARG, "dsplit must be -1 (auto), 0 or 1");
        h->dsplit = (int)value;
        decide_dsplit(h);
    } else if (k == "share") {
        if (value != 8 && value != 4 && value != -1) return fail(KNG_E_ARG, "share must be 8 (one inversion per CU), 4 (one per 256-thread block) or -1 (by herd size); the every-wave-inverts form of rounds 1-3 was removed");
        h->share = (int)value;
    } else if (k == "dp_ring") {
        if (value < 0 || value > 1) return fail(KNG_E_ARG, "dp_ring must be 0 or 1");
        if ((int)value == h->dp_ring) return KNG_OK;
        // points of a launch that was waited for but not drained live in the buffers about to be released
        if (h->slot_ready >= 0) return fail(KNG_E_STATE, "drain the points of the last waited launch before switching dp_ring");
        HIP_TRY(hipSetDevice(h->dev));
        const int before = h->dp_ring;
        h->dp_ring = (int)value;
        if (int rc = alloc_dp_buffers(h)) {
            free_dp_buffers(h, h->dp_ring != 0); // whatever part of the new kind was obtained
            h->dp_ring = before;
            return rc;
        }
        free_dp_buffers(h, before != 0);
        h->view = nullptr;
        h->bytes = device_bytes(h);
        if (h->have_params) {
            int rc = upload_loop_args(h);
            if (rc != KNG_OK) return rc;
        }
    } else if (k == "asm") {
        if (value < 0 || value > 2) return fail(KNG_E_ARG, "asm must be 0, 1 or 2 (2 = measurement only: the scheduled loop without its memory accesses)");
        if (value && (h->n > (1ull << 28) || h->max_found > (1u << 26))) return fail(KNG_E_ARG, "the scheduled loop addresses at most 2^28 kangaroos and 2^26 DP records");
        h->use_asm = (int)value;
        decide_dsplit(h); // (the automatic choice depends on which loop runs)
    } else {
        return fail(KNG_E_ARG, "unknown option '%s'", key);
    }
    return KNG_OK;
}

int kng_reserve_points(kng_engine *h, uint32_t points) {
    if (!h) return fail(KNG_E_ARG, "null engine");
    if (points <= h->max_found) return KNG_OK;
    if (h->outstanding || h->slot_ready >= 0) return fail(KNG_E_STATE, "kng_reserve_points: a launch is outstanding or its points have not been drained");
    if (h->use_asm && points > (1u << 26)) return fail(KNG_E_ARG, "the scheduled loop addresses at most 2^26 DP records");
    HIP_TRY(hipSetDevice(h->dev));
    // The new buffers are obtained BEFORE the old ones are given up: when the bigger allocation is refused -- a
    // pinned ring of 2 x 268 MB against a memlock limit -- the engine keeps the buffers the device-side loop arguments point
    // at, at their old size, and stays launchable.  (Test hook: KNG_TEST_FAIL_RESERVE=1 refuses every growth.)
    const uint32_t before = h->max_found;
    DpRecord *old_ring[2] = {h->ring[0], h->ring[1]}, *old_ring_dev[2] = {h->ring_dev[0], h->ring_dev[1]};
    DpRecord *old_items[2] = {h->dp_items[0], h->dp_items[1]}, *old_h_items = h->h_items;
    for (int s = 0; s < 2; s++) h->ring[s] = h->ring_dev[s] = h->dp_items[s] = nullptr;
    h->h_items = nullptr;
    h->max_found = points;
    int rc = getenv("KNG_TEST_FAIL_RESERVE") ? fail(KNG_E_ALLOC, "pinned DP ring (%zu bytes): refused (KNG_TEST_FAIL_RESERVE)", (size_t)points * sizeof(DpRecord))
                                             : alloc_dp_buffers(h);
    if (rc != KNG_OK) {
        const std::string why = kng_last_error(); // (freeing does not fail() -- but keep the text of the allocation anyway)
        free_dp_buffers(h, h->dp_ring != 0); // whatever part of the new set was obtained
        (void)hipGetLastError();
        for (int s = 0; s < 2; s++) { h->ring[s] = old_ring[s]; h->ring_dev[s] = old_ring_dev[s]; h->dp_items[s] = old_items[s]; }
        h->h_items = old_h_items;
        h->max_found = before;
        return fail(rc, "%s", why.c_str());
    }
    for (int s = 0; s < 2; s++) {
        if (old_ring[s]) (void)hipHostFree(old_ring[s]);
        if (old_items[s]) (void)hipFree(old_items[s]);
    }
    if (old_h_items) (void)hipHostFree(old_h_items);
    h->view = nullptr;
    h->bytes = device_bytes(h);
    if (h->have_params) return upload_loop_args(h);
    return KNG_OK;
}

int kng_get_option(const kng_engine *h, const char *key, int64_t *value) {
    if (!h || !key || !value) return fail(KNG_E_ARG, "null argument");
    std::string k(key);
    if (k == "group") *value = h->group;
    else if (k == "block") *value = h->block;
    else if (k == "steps") *value = h->nsteps;
    else if (k == "lanes") *value = h->lanes;
    else if (k == "share") *value = h->share == -1 ? ((uint64_t)h->lanes < (uint64_t)h->cu_count * 512 ? 4 : 8) : h->share;
    else if (k == "asm") *value = h->use_asm;
    else if (k == "dp_ring") *value = h->dp_ring;
    else if (k == "dsplit") *value = h->dsplit_on ? 1 : 0;
    else if (k == "exact_exits") *value = h->last_exact_exits;
    else if (k == "cu_count") *value = h->cu_count;
    else if (k == "max_found") *value = h->max_found;
    else if (k == "audit_us") *value = (int64_t)(h->last_audit_ms * 1000.0f + 0.5f);
    else if (k == "waves_per_cu") *value = h->cu_count ? (int64_t)((h->lanes / 64 + h->cu_count - 1) / h->cu_count) : 0;
    else return fail(KNG_E_ARG, "unknown option '%s'", key);
    return KNG_OK;
}

int kng_set_params(kng_engine *h, uint64_t dp_mask, const uint64_t *jd, const uint64_t *jx, const uint64_t *jy) {
    if (!h || !jd || !jx || !jy) return fail(KNG_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->dev));
    // limb-major table: word (plane p, entry j) at p*32 + j
    uint64_t tab[JT_WORDS];
    for (int j = 0; j < KNG_NB_JUMP; j++) {
        for (int k = 0; k < 4; k++) {
            tab[JT_JX + k * 32 + j] = jx[4 * j + k];
            tab[JT_JY + k * 32 + j] = jy[4 * j + k];
        }
        tab[JT_JD + j] = jd[2 * j];
        tab[JT_JD + 32 + j] = jd[2 * j + 1];
    }
    h->jd_max = 0;
    for (int j = 0; j < KNG_NB_JUMP; j++) {
        if (jd[2 * j + 1]) h->jd_max = UINT64_MAX;
        else if (h->jd_max != UINT64_MAX && jd[2 * j] > h->jd_max) h->jd_max = jd[2 * j];
    }
    decide_dsplit(h);
    h->dp_mask = dp_mask;
    if (getenv("KNG_TRACE"))
        fprintf(stderr, "kng: planes %p..%p (n=%llu) jtab %p asm_args %p dp_count %p %p dp_items %p %p max_found %u\n", (void *)h->planes,
                (void *)(h->planes + 7 * h->n), (unsigned long long)h->n, (void *)h->jtab, (void *)h->asm_args, (void *)h->dp_count[0],
                (void *)h->dp_count[1], (void *)h->dp_items[0], (void *)h->dp_items[1], h->max_found);
    // Both uploads are copies on the walk stream: a launch in flight finishes with the table and mask it started with
    // (its loop re-reads WalkAsmArgs at every entry), the next launch sees the new ones; the call returns when they have
    // landed, i.e. it blocks behind a running kernel like the reference's cudaMemcpyToSymbol (GPUEngine.cu:565-583).
    HIP_TRY(hipMemcpyAsync(h->jtab, tab, sizeof tab, hipMemcpyHostToDevice, h->walk));
    int rc = upload_loop_args(h); // (synchronises the stream: `tab` lives on this stack)
    if (rc != KNG_OK) return rc;
    h->have_params = true;
    h->products_valid = false; // another jump table: the stored products are those of the old one's dx
    return KNG_OK;
}

int kng_set_kangaroos(kng_engine *h, const uint64_t *x, size_t xs, const uint64_t *y, size_t ys, const uint64_t *d,
                      size_t ds, uint64_t n) {
    if (!h) return fail(KNG_E_ARG, "null argument");
    if (n != h->n) return fail(KNG_E_ARG, "expected %llu kangaroos, got %llu", (unsigned long long)h->n, (unsigned long long)n);
    return kng_set_kangaroos_range(h, 0, n, x, xs, y, ys, d, ds);
}

int kng_set_kangaroos_range(kng_engine *h, uint64_t first, uint64_t count, const uint64_t *x, size_t xs, const uint64_t *y,
                            size_t ys, const uint64_t *d, size_t ds) {
    if (!h || !x || !y || !d) return fail(KNG_E_ARG, "null argument");
    if (first > h->n || count > h->n - first) return fail(KNG_E_ARG, "range %llu+%llu outside the herd of %llu", (unsigned long long)first, (unsigned long long)count, (unsigned long long)h->n);
    if (xs < 4 || ys < 4 || ds < 2) return fail(KNG_E_ARG, "bad stride");
    HIP_TRY(hipSetDevice(h->dev));
    const size_t C = h->stage_kang;
    // the caller's arrays hold kangaroo first + r at index r
    for (uint64_t r0 = 0; r0 < count; r0 += C) {
        const size_t m = (size_t)((count - r0 < C) ? (count - r0) : C);
        const uint64_t c0 = first + r0;
        v16 *st = h->h_stage;
        uint64_t *sd = reinterpret_cast<uint64_t *>(st + 4 * C); // d low words [C], high words [C]
        for (size_t i = 0; i < m; i++) {
            const uint64_t *px = x + (r0 + i) * xs, *py = y + (r0 + i) * ys, *pd = d + (r0 + i) * ds;
            st[0 * C + i] = make_ulonglong2(px[0], px[1]);
            st[1 * C + i] = make_ulonglong2(px[2], px[3]);
            st[2 * C + i] = make_ulonglong2(py[0], py[1]);
            st[3 * C + i] = make_ulonglong2(py[2], py[3]);
            sd[i] = pd[0];
            sd[C + i] = pd[1];
        }
        for (int k = 0; k < 4; k++)
            HIP_TRY(hipMemcpyAsync(plane(h, k) + c0, st + (size_t)k * C, m * sizeof(v16), hipMemcpyHostToDevice, h->walk));
        for (int k = 0; k < 2; k++) // distance plane: N low words, then N high words
            HIP_TRY(hipMemcpyAsync(dplane(h, k) + c0, sd + (size_t)k * C, m * sizeof(uint64_t), hipMemcpyHostToDevice, h->walk));
        HIP_TRY(hipStreamSynchronize(h->walk)); // staging buffer is reused
    }
    // the herd counts as loaded once its last kangaroo has been written (ranges are normally uploaded in order)
    h->products_valid = false;
    if (first + count == h->n) h->have_herd = true;
    return KNG_OK;
}

int kng_get_kangaroos(kng_engine *h, uint64_t *x, size_t xs, uint64_t *y, size_t ys, uint64_t *d, size_t ds, uint64_t n) {
    if (!h) return fail(KNG_E_ARG, "null argument");
    if (n != h->n) return fail(KNG_E_ARG, "expected %llu kangaroos, got %llu", (unsigned long long)h->n, (unsigned long long)n);
    return kng_get_kangaroos_range(h, 0, n, x, xs, y, ys, d, ds);
}

int kng_get_kangaroos_range(kng_engine *h, uint64_t first, uint64_t count, uint64_t *x, size_t xs, uint64_t *y, size_t ys,
                            uint64_t *d, size_t ds) {
    if (!h || !x || !y || !d) return fail(KNG_E_ARG, "null argument");
    if (first > h->n || count > h->n - first) return fail(KNG_E_ARG, "range %llu+%llu outside the herd of %llu", (unsigned long long)first, (unsigned long long)count, (unsigned long long)h->n);
    if (xs < 4 || ys < 4 || ds < 2) return fail(KNG_E_ARG, "bad stride");
    if (!h->have_herd) return fail(KNG_E_STATE, "no herd loaded");
    HIP_TRY(hipSetDevice(h->dev));
    const size_t C = h->stage_kang;
    for (uint64_t r0 = 0; r0 < count; r0 += C) {
        const size_t m = (size_t)((count - r0 < C) ? (count - r0) : C);
        const uint64_t c0 = first + r0;
        v16 *st = h->h_stage;
        uint64_t *sd = reinterpret_cast<uint64_t *>(st + 4 * C); // d low words [C], high words [C]
        // stream-ordered behind an in-flight launch: returns the state that launch leaves
        for (int k = 0; k < 4; k++)
            HIP_TRY(hipMemcpyAsync(st + (size_t)k * C, plane(h, k) + c0, m * sizeof(v16), hipMemcpyDeviceToHost, h->walk));
        for (int k = 0; k < 2; k++)
            HIP_TRY(hipMemcpyAsync(sd + (size_t)k * C, dplane(h, k) + c0, m * sizeof(uint64_t), hipMemcpyDeviceToHost, h->walk));
        HIP_TRY(hipStreamSynchronize(h->walk));
        for (size_t i = 0; i < m; i++) {
            uint64_t *px = x + (r0 + i) * xs, *py = y + (r0 + i) * ys, *pd = d + (r0 + i) * ds;
            px[0] = st[0 * C + i].x; px[1] = st[0 * C + i].y; px[2] = st[1 * C + i].x; px[3] = st[1 * C + i].y;
            py[0] = st[2 * C + i].x; py[1] = st[2 * C + i].y; py[2] = st[3 * C + i].x; py[3] = st[3 * C + i].y;
            pd[0] = sd[i]; pd[1] = sd[C + i];
        }
    }
    return KNG_OK;
}

// ---- work-file snapshot: Backup.cpp:525-546 / :211-231 through GPUEngine::GetKangaroos / SetKangaroos ---------------------
static uint64_t device_bytes(const kng_engine *h) {
    return 7 * (uint64_t)h->n * sizeof(v16) + JT_WORDS * 8 + 2 * 64 + (h->dp_ring ? 0 : 2 * (uint64_t)h->max_found * sizeof(DpRecord)) +
           (h->snap ? 96 * (uint64_t)h->n : 0);
}
static int snapshot_buffers(kng_engine *h) {
    hipError_t e;
    if (!h->snap_stream && hipStreamCreateWithFlags(&h->snap_stream, hipStreamNonBlocking) != hipSuccess) return fail(KNG_E_HIP, "stream creation failed");
    if (!h->snap_ev && hipEventCreateWithFlags(&h->snap_ev, hipEventDisableTiming) != hipSuccess) return fail(KNG_E_HIP, "event creation failed");
    if (!h->snap_status && (e = hipHostMalloc((void **)&h->snap_status, 64, hipHostMallocDefault)) != hipSuccess)
        return fail(KNG_E_ALLOC, "pinned status block: %s", hipGetErrorString(e));
    if (!h->snap_status_dev && (e = hipMalloc((void **)&h->snap_status_dev, 64)) != hipSuccess) return fail(KNG_E_ALLOC, "status block: %s", hipGetErrorString(e));
    if (!h->snap) {
        if ((e = hipMalloc((void **)&h->snap, 96 * (size_t)h->n)) != hipSuccess) {
            (void)hipGetLastError();
            h->snap = nullptr;
            return fail(KNG_E_ALLOC, "snapshot records (%zu bytes): %s", 96 * (size_t)h->n, hipGetErrorString(e));
        }
        h->bytes = device_bytes(h);
    }
    return KNG_OK;
}
static void snap_args(const kng_engine *h, SnapArgs &a, uint64_t first, uint64_t count, const uint64_t wild_offset[4]) {
    a.x01 = plane(h, 0); a.x23 = plane(h, 1); a.y01 = plane(h, 2); a.y23 = plane(h, 3);
    a.dlo = dplane(h, 0); a.dhi = dplane(h, 1);
    a.rec = h->snap; a.first = first; a.count = count;
    for (int k = 0; k < 4; k++) a.woff[k] = wild_offset ? wild_offset[k] : 0;
    a.status = nullptr;
}

int kng_snapshot(kng_engine *h, const uint64_t wild_offset[4]) {
    if (!h) return fail(KNG_E_ARG, "null engine");
    if (!h->have_herd) return fail(KNG_E_STATE, "no herd loaded");
    if (getenv("KNG_TEST_FAIL_SNAPSHOT")) return fail(KNG_E_ALLOC, "snapshot records (%zu bytes): refused (KNG_TEST_FAIL_SNAPSHOT)", 96 * (size_t)h->n); // test hook
    HIP_TRY(hipSetDevice(h->dev));
    if (int rc = snapshot_buffers(h)) return rc;
    // a reader of the previous snapshot must be done before its records are overwritten (the caller's protocol says so;
    // this makes a violation a delay, not a torn file)
    HIP_TRY(hipStreamSynchronize(h->snap_stream));
    SnapArgs a;
    snap_args(h, a, 0, h->n, wild_offset);
    hipLaunchKernelGGL(kng_snapshot_pack_kernel, dim3((unsigned)((h->n + 255) / 256)), dim3(256), 0, h->walk, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->snap_ev, h->walk));
    h->snap_taken = true;
    return KNG_OK;
}

int kng_snapshot_read(kng_engine *h, uint64_t first, uint64_t count, void *dst) {
    if (!h || (!dst && count)) return fail(KNG_E_ARG, "null argument");
    if (!h->snap_taken) return fail(KNG_E_STATE, "no snapshot taken");
    if (first > h->n || count > h->n - first) return fail(KNG_E_ARG, "range %llu+%llu outside the herd of %llu", (unsigned long long)first, (unsigned long long)count, (unsigned long long)h->n);
    if (!count) return KNG_OK;
    HIP_TRY(hipSetDevice(h->dev));
    HIP_TRY(hipStreamWaitEvent(h->snap_stream, h->snap_ev, 0));
    HIP_TRY(hipMemcpyAsync(dst, h->snap + 12 * first, 96 * (size_t)count, hipMemcpyDeviceToHost, h->snap_stream));
    HIP_TRY(hipStreamSynchronize(h->snap_stream));
    return KNG_OK;
}

int kng_snapshot_write(kng_engine *h, uint64_t first, uint64_t count, const void *src) {
    if (!h || (!src && count)) return fail(KNG_E_ARG, "null argument");
    if (first > h->n || count > h->n - first) return fail(KNG_E_ARG, "range %llu+%llu outside the herd of %llu", (unsigned long long)first, (unsigned long long)count, (unsigned long long)h->n);
    HIP_TRY(hipSetDevice(h->dev));
    if (int rc = snapshot_buffers(h)) return rc;
    h->snap_taken = false; // the records are no longer those of the herd
    if (!count) return KNG_OK;
    HIP_TRY(hipMemcpyAsync(h->snap + 12 * first, src, 96 * (size_t)count, hipMemcpyHostToDevice, h->snap_stream));
    HIP_TRY(hipStreamSynchronize(h->snap_stream));
    return KNG_OK;
}

int kng_snapshot_restore(kng_engine *h, uint64_t first, uint64_t count, const uint64_t wild_offset[4], uint64_t *bad_index) {
    if (!h) return fail(KNG_E_ARG, "null engine");
    if (!h->snap) return fail(KNG_E_STATE, "nothing uploaded (kng_snapshot_write)");
    if (first > h->n || count > h->n - first) return fail(KNG_E_ARG, "range %llu+%llu outside the herd of %llu", (unsigned long long)first, (unsigned long long)count, (unsigned long long)h->n);
    HIP_TRY(hipSetDevice(h->dev));
    if (count) {
        SnapArgs a;
        snap_args(h, a, first, count, wild_offset);
        h->snap_status[0] = 0;
        h->snap_status[1] = ~0ULL;
        a.status = h->snap_status_dev;
        // on the walk stream: ordered behind a launch in flight, like kng_set_kangaroos_range
        HIP_TRY(hipMemcpyAsync(h->snap_status_dev, h->snap_status, 16, hipMemcpyHostToDevice, h->walk));
        hipLaunchKernelGGL(kng_snapshot_unpack_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, h->walk, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(h->snap_status, h->snap_status_dev, 16, hipMemcpyDeviceToHost, h->walk));
        HIP_TRY(hipStreamSynchronize(h->walk));
        h->products_valid = false;
        if (h->snap_status[0]) {
            if (bad_index) *bad_index = h->snap_status[1];
            return fail(KNG_E_ARG, "%llu restored distances do not fit the 128-bit device distance (first: kangaroo %llu)",
                        (unsigned long long)h->snap_status[0], (unsigned long long)h->snap_status[1]);
        }
    }
    if (first + count == h->n) h->have_herd = true;
    return KNG_OK;
}

int kng_snapshot_release(kng_engine *h) {
    if (!h) return fail(KNG_E_ARG, "null engine");
    HIP_TRY(hipSetDevice(h->dev));
    if (h->snap_stream) HIP_TRY(hipStreamSynchronize(h->snap_stream));
    HIP_TRY(hipStreamSynchronize(h->walk));
    if (h->snap) (void)hipFree(h->snap);
    h->snap = nullptr;
    h->snap_taken = false;
    h->bytes = device_bytes(h);
    return KNG_OK;
}

int kng_build_herd(kng_engine *h, int range_power, uint64_t seed, const uint64_t *table, uint32_t windows,
                   const uint64_t base_tame[8], const uint64_t base_wild[8], const uint64_t final_add[8]) {
    if (!h || !table || !base_tame || !base_wild || !final_add) return fail(KNG_E_ARG, "null argument");
    if (range_power < 1 || range_power > 128) return fail(KNG_E_ARG, "range_power must be 1..128");
    if (windows != (uint32_t)(range_power + 7) / 8) return fail(KNG_E_ARG, "windows must be ceil(range_power/8)");
    if (h->outstanding) return fail(KNG_E_STATE, "a launch is outstanding");
    HIP_TRY(hipSetDevice(h->dev));
    DevBuf dtab;
    const size_t tbytes = (size_t)windows * 256 * 8 * sizeof(uint64_t);
    HIP_TRY(dtab.alloc(tbytes));
    HIP_TRY(hipMemcpyAsync(dtab.p, table, tbytes, hipMemcpyHostToDevice, h->walk));
    HerdArgs a;
    a.x01 = plane(h, 0); a.x23 = plane(h, 1); a.y01 = plane(h, 2); a.y23 = plane(h, 3);
    a.d = plane(h, 4); a.s01 = plane(h, 5); a.s23 = plane(h, 6);
    a.table = dtab.as<uint64_t>();
    memcpy(a.base[0], base_tame, 64);
    memcpy(a.base[1], base_wild, 64);
    memcpy(a.fin, final_add, 64);
    a.seed = seed;
    a.n_kang = h->n;
    a.windows = windows;
    a.range_power = (uint32_t)range_power;
    a.lanes = h->lanes;
    const uint32_t blocks = (h->lanes + h->block - 1) / h->block;
    hipLaunchKernelGGL(kng_herd_kernel, dim3(blocks), dim3(h->block), 0, h->walk, a);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(h->walk));
    h->have_herd = true;
    h->products_valid = false;
    return KNG_OK;
}

// ---- whole-run audit: every kangaroo / every DP record re-derived from its distance (kng_audit_kernel) ----

int kng_audit_setup(kng_engine *h, const uint64_t *table, const uint64_t base_tame[8], const uint64_t base_wild[8], const uint64_t final_add[8]) {
    if (!h || !table || !base_tame || !base_wild || !final_add) return fail(KNG_E_ARG, "null argument");
    HIP_TRY(hipSetDevice(h->dev));
    const size_t tbytes = (size_t)KNG_AUDIT_WINDOWS * 256 * 8 * sizeof(uint64_t);
    hipError_t e;
    if (!h->audit_tab && (e = hipMalloc((void **)&h->audit_tab, tbytes)) != hipSuccess) return fail(KNG_E_ALLOC, "audit table: %s", hipGetErrorString(e));
    if (!h->audit_res && (e = hipMalloc((void **)&h->audit_res, (2 + KNG_AUDIT_CAP) * 8)) != hipSuccess) return fail(KNG_E_ALLOC, "audit result: %s", hipGetErrorString(e));
    if (!h->audit_res_host && (e = hipHostMalloc((void **)&h->audit_res_host, (2 + KNG_AUDIT_CAP) * 8, hipHostMallocDefault)) != hipSuccess)
        return fail(KNG_E_ALLOC, "pinned audit result: %s", hipGetErrorString(e));
    HIP_TRY(hipMemcpy(h->audit_tab, table, tbytes, hipMemcpyHostToDevice));
    memcpy(h->audit_base[0], base_tame, 64);
    memcpy(h->audit_base[1], base_wild, 64);
    memcpy(h->audit_fin, final_add, 64);
    h->audit_ready = true;
    return KNG_OK;
}

int kng_audit_herd(kng_engine *h, uint64_t *n_bad, uint64_t *bad_idx, uint32_t bad_cap) {
    if (!h || !n_bad) return fail(KNG_E_ARG, "null argument");
    *n_bad = 0;
    if (!h->audit_ready) return fail(KNG_E_STATE, "kng_audit_setup has not been called");
    if (!h->have_herd) return fail(KNG_E_STATE, "no herd loaded");
    if (h->outstanding) return fail(KNG_E_STATE, "a launch is outstanding (the audit borrows the walk's product planes)");
    HIP_TRY(hipSetDevice(h->dev));
    DevBuf scratch;
    hipError_t e = scratch.alloc(4 * (size_t)h->n * sizeof(v16));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(KNG_E_ALLOC, "audit scratch (%zu bytes): %s", 4 * (size_t)h->n * sizeof(v16), hipGetErrorString(e));
    }
    AuditArgs a{};
    a.ex01 = plane(h, 0); a.ex23 = plane(h, 1); a.ey01 = plane(h, 2); a.ey23 = plane(h, 3);
    a.dlo = dplane(h, 0); a.dhi = dplane(h, 1);
    a.recs = nullptr;
    h->last_audit_ms = 0.f;
    h->products_valid = false;
    uint32_t recorded = 0;
    // the S planes are borrowed: the next launch starts with its own product pass
    return run_audit<false>(h, h->walk, a, scratch.as<v16>(), plane(h, 5), plane(h, 6), h->n, h->lanes, n_bad, bad_idx, bad_cap, 0, &recorded);
}

int kng_audit_points(kng_engine *h, const kng_dp_record *recs, uint64_t n, uint64_t *n_bad, uint64_t *bad_idx, uint32_t bad_cap) {
    if (!h || !n_bad || (!recs && n)) return fail(KNG_E_ARG, "null argument");
    *n_bad = 0;
    if (!h->audit_ready) return fail(KNG_E_STATE, "kng_audit_setup has not been called");
    if (n == 0) return KNG_OK;
    HIP_TRY(hipSetDevice(h->dev));
    const uint64_t C = n < (1ull << 21) ? n : (1ull << 21); // records per launch
    DevBuf scratch, drec;
    hipError_t e;
    if ((e = scratch.alloc(6 * (size_t)C * sizeof(v16))) != hipSuccess || (e = drec.alloc((size_t)C * sizeof(DpRecord))) != hipSuccess) {
        (void)hipGetLastError();
        return fail(KNG_E_ALLOC, "audit scratch for %llu records: %s", (unsigned long long)C, hipGetErrorString(e));
    }
    h->last_audit_ms = 0.f;
    uint32_t recorded = 0;
    // own stream: the records need nothing of the herd, so a walk launch in flight is not disturbed
    for (uint64_t c0 = 0; c0 < n; c0 += C) {
        const uint64_t m = n - c0 < C ? n - c0 : C;
        HIP_TRY(hipMemcpyAsync(drec.p, recs + c0, (size_t)m * sizeof(DpRecord), hipMemcpyHostToDevice, h->copy));
        // batch of ~32 per lane, at most two waves per SIMD
        uint64_t lanes = ((m + 31) / 32 + 63) / 64 * 64;
        const uint64_t most = (uint64_t)h->cu_count * 512;
        if (lanes > most) lanes = most;
        AuditArgs a{};
        a.recs = drec.as<DpRecord>();
        v16 *sc = scratch.as<v16>();
        int rc = run_audit<true>(h, h->copy, a, sc, sc + 4 * m, sc + 5 * m, m, (uint32_t)lanes, n_bad, bad_idx, bad_cap, c0, &recorded);
        if (rc != KNG_OK) return rc;
    }
    return KNG_OK;
}

int kng_set_kangaroo(kng_engine *h, uint64_t kidx, const uint64_t x[4], const uint64_t y[4], const uint64_t d[2]) {
    if (!h || !x || !y || !d) return fail(KNG_E_ARG, "null argument");
    if (kidx >= h->n) return fail(KNG_E_ARG, "kIdx %llu out of range", (unsigned long long)kidx);
    HIP_TRY(hipSetDevice(h->dev));
    const fe fx{{x[0], x[1], x[2], x[3]}}, fy{{y[0], y[1], y[2], y[3]}};
    h->products_valid = false; // the next launch starts with its own product pass again
    hipLaunchKernelGGL(kng_patch_kernel, dim3(1), dim3(1), 0, h->walk, plane(h, 0), plane(h, 1), plane(h, 2), plane(h, 3), plane(h, 4),
                       h->n, kidx, fx, fy, make_ulonglong2(d[0], d[1]));
    HIP_TRY(hipGetLastError());
    return KNG_OK;
}

int kng_launch(kng_engine *h) {
    if (!h) return fail(KNG_E_ARG, "null engine");
    if (!h->have_params) return fail(KNG_E_STATE, "kng_set_params has not been called");
    if (!h->have_herd) return fail(KNG_E_STATE, "kng_set_kangaroos has not been called");
    if (h->outstanding) return fail(KNG_E_STATE, "a launch is already outstanding; call kng_wait first");
    HIP_TRY(hipSetDevice(h->dev));
    const int s = h->slot_next;
    WalkArgs a;
    a.x01 = plane(h, 0); a.x23 = plane(h, 1); a.y01 = plane(h, 2); a.y23 = plane(h, 3);
    a.d = plane(h, 4); a.s01 = plane(h, 5); a.s23 = plane(h, 6);
    a.jtab = h->jtab;
    a.dp_mask = h->dp_mask;
    a.dp_count = h->dp_count[s];
    a.dp_items = dp_buffer(h, s);
    a.max_found = h->max_found;
    a.lanes = h->lanes;
    a.group = h->group;
    a.n_kang = h->n;
    a.nsteps = h->nsteps;
    a.resume = h->products_valid ? 1 : 0;
    // Whatever fails from here on, the S planes can no longer be trusted to hold the products of the state the next launch
    // starts from: the flag is raised again only once this launch is known to be queued.
    h->products_valid = false;
    a.asm_args = (uint64_t)(h->asm_args + s);
    HIP_TRY(hipMemsetAsync(h->dp_count[s], 0, 8, h->walk)); // GPUEngine.cu:543 (+ the launch's exact-path exit counter)
    HIP_TRY(hipEventRecord(h->ev_start[s], h->walk));
    const bool ds = h->dsplit_on;
    // herds whose lanes do not give every CU a 512-thread block walk in 256-thread blocks (one inversion per four waves,
    // one wave per SIMD, twice as many CUs busy): option "share" -1 (default) decides per herd, 8 / 4 force the form
    const int share = h->share == -1 ? ((uint64_t)h->lanes < (uint64_t)h->cu_count * 512 ? 4 : 8) : h->share;
    h->share_used = share;
    const uint32_t bthreads = share == 8 ? 512 : 256;
    const dim3 grid2((h->lanes + bthreads - 1) / bthreads);
#define KNG_LAUNCH(SH, DS, AS, GRID, BLOCK) hipLaunchKernelGGL((kng_walk_share_kernel<SH, DS, AS>), GRID, dim3(BLOCK), 0, h->walk, a)
    // four instantiations: the scheduled loop for either distance layout, and the compiler-scheduled pair that serves herds
    // beyond 2^28 kangaroos (and is the exact path behind the scheduled loop).  Round 1-3's share = 1 family is gone.
    if (h->use_asm == 2) { // measurement: the ALU ceiling of the headline kernel (wrong results on purpose)
        if (share != 8 || !ds) return fail(KNG_E_STATE, "\"asm\" 2 exists for the headline form only: share 8, low-word distance streaming");
        hipLaunchKernelGGL(kng_walk_valu_only_kernel, grid2, dim3(512), 0, h->walk, a);
    } else if (share == 8) {
        if (h->use_asm) { if (ds) KNG_LAUNCH(8, true, true, grid2, 512); else KNG_LAUNCH(8, false, true, grid2, 512); }
        else { if (ds) KNG_LAUNCH(8, true, false, grid2, 512); else KNG_LAUNCH(8, false, false, grid2, 512); }
    } else {
        if (h->use_asm) { if (ds) KNG_LAUNCH(4, true, true, grid2, 256); else KNG_LAUNCH(4, false, true, grid2, 256); }
        else { if (ds) KNG_LAUNCH(4, true, false, grid2, 256); else KNG_LAUNCH(4, false, false, grid2, 256); }
    }
#undef KNG_LAUNCH
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(h->ev_stop[s], h->walk));
    HIP_TRY(hipMemcpyAsync(h->h_count[s], h->dp_count[s], 8, hipMemcpyDeviceToHost, h->walk));
    HIP_TRY(hipEventRecord(h->ev_done[s], h->walk));
    h->outstanding = true;
    h->products_valid = (h->nsteps % 2) == 0; // this launch leaves them in ascending order when its passes pair up
    return KNG_OK;
}

int kng_outstanding(const kng_engine *h) { return (h && h->outstanding) ? 1 : 0; }
int kng_undrained(const kng_engine *h) { return (h && h->slot_ready >= 0) ? 1 : 0; }

void *kng_alloc_pinned(size_t size) {
    void *p = nullptr;
    hipError_t e = hipHostMalloc(&p, size, hipHostMallocPortable);
    if (e != hipSuccess) {
        fail(KNG_E_ALLOC, "hipHostMalloc(%zu): %s", size, hipGetErrorString(e));
        return nullptr;
    }
    return p;
}
void kng_free_pinned(void *p) {
    if (p) (void)hipHostFree(p);
}

int kng_wait(kng_engine *h, int spin) {
    if (!h) return fail(KNG_E_ARG, "null engine");
    if (!h->outstanding) return fail(KNG_E_STATE, "no launch outstanding");
    HIP_TRY(hipSetDevice(h->dev));
    const int s = h->slot_next;
    if (spin) {
        hipError_t e;
        while ((e = hipEventQuery(h->ev_done[s])) == hipErrorNotReady) {
        }
        if (e != hipSuccess) return fail(KNG_E_HIP, "walk kernel: %s", hipGetErrorString(e));
    } else {
        HIP_TRY(hipEventSynchronize(h->ev_done[s])); // blocking-sync event: the host thread sleeps
    }
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_start[s], h->ev_stop[s]));
    h->last_ms = ms;
    h->last_exact_exits = h->h_count[s][1];
    h->outstanding = false;
    h->slot_ready = s;
    h->slot_next = s ^ 1;
    return KNG_OK;
}

// copy the points of the most recently waited launch into the pinned landing buffer; *found <= min(max_found, cap)
static int land_points(kng_engine *h, uint32_t cap, uint32_t *found_out, uint32_t *lost_out) {
    *found_out = 0;
    *lost_out = 0;
    if (h->slot_ready < 0) return KNG_OK; // nothing waited yet (first Launch of the reference protocol)
    HIP_TRY(hipSetDevice(h->dev));
    const int s = h->slot_ready;
    uint32_t found = *h->h_count[s];
    uint32_t lost = 0;
    if (found > h->max_found) { // GPUEngine.cu:641-648
        lost = found - h->max_found;
        found = h->max_found;
    }
    if (found > cap) {
        lost += found - cap;
        found = cap;
    }
    if (h->dp_ring) {
        h->view = h->ring[s]; // the kernel wrote them here; complete since the launch's event fired (kng_wait)
    } else {
        if (found) {
            HIP_TRY(hipMemcpyAsync(h->h_items, h->dp_items[s], (size_t)found * sizeof(DpRecord), hipMemcpyDeviceToHost, h->copy));
            HIP_TRY(hipStreamSynchronize(h->copy));
        }
        h->view = h->h_items;
    }
    *found_out = found;
    *lost_out = lost;
    h->slot_ready = -1; // drained
    return KNG_OK;
}

int kng_drain(kng_engine *h, kng_item *items, uint32_t cap, uint32_t *n_items, uint32_t *n_lost) {
    if (!h || !n_items) return fail(KNG_E_ARG, "null argument");
    *n_items = 0;
    if (n_lost) *n_lost = 0;
    if (!items && cap) return fail(KNG_E_ARG, "null items with room for %u", cap);
    uint32_t found = 0, lost = 0;
    const int rc = land_points(h, cap, &found, &lost);
    if (rc != KNG_OK) return rc;
    for (uint32_t i = 0; i < found; i++) {
        memcpy(items[i].x, h->view[i].x, 32);
        items[i].d[0] = h->view[i].d[0];
        items[i].d[1] = h->view[i].d[1];
        items[i].kidx = h->view[i].kidx;
    }
    *n_items = found;
    if (n_lost) *n_lost = lost;
    return KNG_OK;
}

static_assert(sizeof(kng_dp_record) == sizeof(DpRecord), "kng_dp_record is the record the kernel writes");

int kng_drain_view(kng_engine *h, const kng_dp_record **records, uint32_t *n_items, uint32_t *n_lost) {
    if (!h || !records || !n_items) return fail(KNG_E_ARG, "null argument");
    *records = nullptr;
    *n_items = 0;
    if (n_lost) *n_lost = 0;
    uint32_t found = 0, lost = 0;
    const int rc = land_points(h, h->max_found, &found, &lost);
    if (rc != KNG_OK) return rc;
    *records = reinterpret_cast<const kng_dp_record *>(h->view ? h->view : h->h_items);
    *n_items = found;
    if (n_lost) *n_lost = lost;
    return KNG_OK;
}

int kng_last_kernel_ms(const kng_engine *h, float *ms) {
    if (!h || !ms) return fail(KNG_E_ARG, "null argument");
    *ms = h->last_ms;
    return KNG_OK;
}

int kng_test_fieldop(int dev, int op, const uint64_t *a, const uint64_t *b, uint64_t *r, uint64_t n) {
    if (!a || !b || !r) return fail(KNG_E_ARG, "null argument");
    if (op < KNG_OP_MODMUL || op > KNG_OP_MODINV) return fail(KNG_E_ARG, "unknown op %d", op);
    if (n == 0) return KNG_OK;
    if (dev < 0 || dev >= kng_device_count()) return fail(KNG_E_NODEVICE, "invalid device %d (no CPU fallback)", dev);
    HIP_TRY(hipSetDevice(dev));
    DevBuf da, db, dr;
    const size_t bytes = (size_t)n * 32;
    HIP_TRY(da.alloc(bytes));
    HIP_TRY(db.alloc(bytes));
    HIP_TRY(dr.alloc(bytes));
    HIP_TRY(hipMemcpy(da.p, a, bytes, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(db.p, b, bytes, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(kng_fieldop_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, op, da.as<uint64_t>(), db.as<uint64_t>(),
                       dr.as<uint64_t>(), n);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(r, dr.p, bytes, hipMemcpyDeviceToHost));
    return KNG_OK;
}

} // extern "C"
