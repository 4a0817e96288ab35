/*
 * kng_oracle.h -- CPU restatement of the kangaroo jump path.  TEST INFRASTRUCTURE ONLY.
 *
 * This library is the parity checker for the HIP jump engine.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the
 * product path (kangaroo_amd/csrc, kangaroo_amd/host) never links or calls it.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose behaviour it restates.  Parity status: PINNED -- checked against the
 * reference's own SECPK1 objects (oracle/_ref/refprobe, built by
 * oracle/Makefile from the sources where they lie) and against the golden
 * vectors committed under tests/golden/.
 *
 * All big integers are little-endian arrays of uint64_t limbs.
 */
#ifndef KNG_ORACLE_H
#define KNG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_NB_JUMP 32 /* Constants.h:29 */
#define ORC_NB_RUN 64  /* Constants.h:35 */

/* ---- field arithmetic mod p = 2^256 - 0x1000003D1 ------------------------------- */
/* SECPK1/IntMod.cpp:873-950 (ModMulK1) == GPU/GPUMath.h:810-858 (_ModMult):
 * schoolbook 512-bit product, fold hi*0x1000003D1 twice, final carry dropped,
 * no comparison with p.  Result in [0,2^256). */
void orc_modmul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]);
/* SECPK1/IntMod.cpp:1030-1234 (ModSquareK1) == GPU/GPUMath.h:909-1019: same fold on a*a. */
void orc_modsqr(uint64_t r[4], const uint64_t a[4]);
/* SECPK1/IntMod.cpp:95-99 (ModSub) == GPU/GPUMath.h:476-494: a-b, +p on borrow. */
void orc_modsub(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]);
/* SECPK1/IntMod.cpp:368-569 (ModInv): canonical inverse in [0,p); inverse of 0 is 0. */
void orc_modinv(uint64_t r[4], const uint64_t a[4]);
/* SECPK1/IntGroup.cpp:36-57 (IntGroup::ModInv): Montgomery trick, in place over n values. */
void orc_batch_inv(uint64_t (*v)[4], size_t n);

/* ---- scalar arithmetic mod the group order n ----------------------------------- */
/* SECPK1/IntMod.cpp:1245-1263: a+b-n, +n if negative / a-b, +n if negative. 256-bit. */
void orc_add_order(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]);
void orc_sub_order(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]);

/* ---- curve ------------------------------------------------------------------- */
/* SECPK1/SECP256K1.cpp:238-263 (AddDirect, single point): p1 + p2, affine, distinct x. */
void orc_add_direct(uint64_t rx[4], uint64_t ry[4], const uint64_t p1x[4], const uint64_t p1y[4],
                    const uint64_t p2x[4], const uint64_t p2y[4]);
/* SECPK1/SECP256K1.cpp ComputePublicKey semantics: k*G in affine coords (k != 0 mod n).
 * Returns 0 on success, -1 for the point at infinity. */
int orc_pubkey(uint64_t x[4], uint64_t y[4], const uint64_t k[4]);
/* general k*P + optional add, used to build wild herds: (x,y) = k*G + (qx,qy); q may be NULL */
int orc_pubkey_add(uint64_t x[4], uint64_t y[4], const uint64_t k[4], const uint64_t qx[4],
                   const uint64_t qy[4]);

/* ---- RNG (SECPK1/Random.cpp:31-127, MT19937) and Int::Rand (SECPK1/Int.cpp:988-1001) ---- */
void orc_rseed(uint32_t seed);
uint32_t orc_rndl(void);
void orc_int_rand(uint64_t r[4], int nbit);

/* ---- kangaroo parameters ---------------------------------------------------------- */
/* Kangaroo.cpp:154-164 (SetDP): mask of the dp leading bits of limb 3; dp==0 -> 0. */
uint64_t orc_dp_mask(int dp);
/* Kangaroo.cpp:742-832 (CreateJumpTable, non-symmetry build): seed 0x600DCAFE,
 * jumpBit = min(128, rangePower/2+1), 32 distances, retry until the mean is in
 * (2^(jumpBit-1.05), 2^(jumpBit-0.95)); points = d*G.  Returns log2(mean) via *avg_log2. */
void orc_jump_table(int range_power, uint64_t jd[ORC_NB_JUMP][2], uint64_t jx[ORC_NB_JUMP][4],
                    uint64_t jy[ORC_NB_JUMP][4], double *avg_log2);

/* ---- the walk -------------------------------------------------------------------- */
typedef struct {
    uint64_t x[4];
    uint64_t d[2];
    uint64_t kidx;
} orc_dp_t; /* same field content as GPU/GPUEngine.h:31-38 ITEM / GPUMath.h:173-188 record */

/* Device-view walk: restates GPU/GPUCompute.h:45-109 with the arithmetic of
 * Kangaroo.cpp:379-433 (batched inverse over the whole herd each step).
 * x,y: n x 4 limbs; d: n x 2 limbs (128-bit, raw add like GPUMath.h:119-121).
 * Runs nsteps jumps for every kangaroo, updating in place.  Every point whose
 * (x.limb3 & dpmask)==0 after a jump is appended to dps (kidx = array index)
 * while fewer than dp_cap were stored.  Returns the total number of DPs seen. */
size_t orc_walk(uint64_t *x, uint64_t *y, uint64_t *d, size_t n, int nsteps,
                const uint64_t jd[ORC_NB_JUMP][2], const uint64_t jx[ORC_NB_JUMP][4],
                const uint64_t jy[ORC_NB_JUMP][4], uint64_t dpmask, orc_dp_t *dps, size_t dp_cap);

/* Host-view walk of Check.cpp:534-586: per-point AddDirect (one full ModInv per jump),
 * distances are 256-bit and advanced with ModAddK1order.  d: n x 4 limbs. */
size_t orc_walk_direct(uint64_t *x, uint64_t *y, uint64_t *d4, size_t n, int nsteps,
                       const uint64_t jd[ORC_NB_JUMP][2], const uint64_t jx[ORC_NB_JUMP][4],
                       const uint64_t jy[ORC_NB_JUMP][4], uint64_t dpmask, orc_dp_t *dps,
                       size_t dp_cap);

/* Herd builder (Kangaroo.cpp:670-738 CreateHerd semantics, distances supplied by the
 * caller): kangaroo i is tame (i+first_type even) -> P = d*G, or wild -> P = K + d*G.
 * d4: n x 4 limbs (true distance mod n). */
void orc_create_herd(uint64_t *x, uint64_t *y, const uint64_t *d4, size_t n, int first_type,
                     const uint64_t kx[4], const uint64_t ky[4]);

#ifdef __cplusplus
}
#endif
#endif
