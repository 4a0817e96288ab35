/*
 * kng_host.h -- host-side support library of the MI355X kangaroo engine (libkangaroo_host.so).
 *
 * Product code (C++17, no GPU needed): what a host program needs AROUND the jump engine when
 * the reference's own host code is not linked in -- bench.py, the tools and our tests' herds:
 *   - secp256k1 field / scalar / affine-point arithmetic written from scratch,
 *   - the reference-compatible jump table (Kangaroo.cpp:742-832: MT19937 seed 0x600DCAFE,
 *     Int::Rand, mean-distance acceptance window) so work files / DPs stay compatible,
 *   - a multi-threaded batched herd builder (CreateHerd semantics, Kangaroo.cpp:670-738),
 *   - distance bookkeeping mod n (wild offset add/sub, GPUEngine.cu:406-411,477,672).
 * Plain C ABI so Python can bind it with ctypes.  All integers: little-endian uint64 limbs.
 */
#ifndef KNG_HOST_H
#define KNG_HOST_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* k*G in affine coordinates; returns 0, or -1 for the point at infinity (k == 0 mod n) */
int kngh_pubkey(const uint64_t k[4], uint64_t x[4], uint64_t y[4]);
/* (x3,y3) = (x1,y1) + (x2,y2), general affine addition (handles doubling); -1 if infinity */
int kngh_point_add(const uint64_t x1[4], const uint64_t y1[4], const uint64_t x2[4], const uint64_t y2[4],
                   uint64_t x3[4], uint64_t y3[4]);
/* is y^2 == x^3 + 7 (mod p)?  1 / 0 */
int kngh_on_curve(const uint64_t x[4], const uint64_t y[4]);

/* r = a + b mod n ; r = a - b mod n  (SECPK1/IntMod.cpp:1245-1263) */
void kngh_add_order(const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);
void kngh_sub_order(const uint64_t a[4], const uint64_t b[4], uint64_t r[4]);

/* Kangaroo.cpp:154-164 SetDP */
uint64_t kngh_dp_mask(int dp);

/* Kangaroo.cpp:742-832 CreateJumpTable (non-symmetry).  jd[32][2], jx[32][4], jy[32][4].
 * Returns the accepted mean jump distance as log2. */
double kngh_jump_table(int range_power, uint64_t *jd, uint64_t *jx, uint64_t *jy);

/* Kangaroo.cpp:980-993: suggested DP size for a herd of total_kangaroos on a 2^range_power range */
int kngh_suggest_dp(int range_power, double total_kangaroos);

/* Herd builder with CreateHerd semantics (Kangaroo.cpp:670-738): kangaroo i is tame when
 * (i + first_type) is even: P = d*G with d uniform in [0, 2^range_power); otherwise wild:
 * P = K + d*G with d uniform in [-N/2, N/2) represented mod n, N/2 = wild_offset.
 * Outputs x,y: n x 4 limbs, d_true: n x 4 limbs (mod n).  Distances come from a
 * SplitMix64/xoshiro stream seeded by `seed` (the reference seeds MT19937 from /dev/urandom,
 * so no particular herd is part of its contract).  Uses nthreads host threads (0 = all). */
int kngh_create_herd(uint64_t n, int range_power, const uint64_t wild_offset[4], const uint64_t kx[4],
                     const uint64_t ky[4], int first_type, uint64_t seed, int nthreads, uint64_t *x, uint64_t *y,
                     uint64_t *d_true);

/* Inputs of kng_build_herd (device-side herd creation, include/kangaroo_hip.h): the window table
 * table[w][v] = v*256^w*G (windows = ceil(range_power/8), 256 entries of x[4],y[4] each, entry 0 zero),
 * and the offset points base_tame = b*G, base_wild = K - wild_offset*G + b*G, final_add = -b*G for a
 * scalar b derived from seed.  table must hold windows*256*8 uint64.  kx/ky may be NULL (tame only). */
int kngh_herd_params(int range_power, const uint64_t wild_offset[4], const uint64_t kx[4], const uint64_t ky[4],
                     uint64_t seed, uint64_t *table, uint64_t base_tame[8], uint64_t base_wild[8],
                     uint64_t final_add[8]);

/* (n x 4 true distances mod n) <-> (n x 2 device distances): odd indices carry +wild_offset mod n.
 * Returns 0, or -1 when a device distance does not fit 128 bits. */
int kngh_to_device_distances(const uint64_t *d_true, uint64_t n, const uint64_t wild_offset[4], uint64_t *d_dev);
void kngh_to_true_distances(const uint64_t *d_dev, const uint64_t *kidx /* may be NULL: 0..n-1 */, uint64_t n,
                            const uint64_t wild_offset[4], uint64_t *d_true);

#ifdef __cplusplus
}
#endif
#endif
