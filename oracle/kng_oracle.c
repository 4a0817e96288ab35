/*
 * kng_oracle.c -- CPU restatement of the kangaroo jump path.  TEST INFRASTRUCTURE ONLY.
 * See kng_oracle.h for the contract and the reference citations.  Plain C11 + __int128.
 * Parity: PINNED against oracle/_ref/refprobe (reference SECPK1 objects) and tests/golden.
 */
#include <pthread.h>
#include "kng_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

static const uint64_t P[4] = {0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL}; /* GPUMath.h:83-88 */
static const uint64_t ORDER[4] = {0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL,
                                  0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}; /* SECP256K1.cpp:38 */
static const uint64_t GX[4] = {0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL,
                               0x79BE667EF9DCBBACULL}; /* SECP256K1.cpp:35 */
static const uint64_t GY[4] = {0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL,
                               0x483ADA7726A3C465ULL}; /* SECP256K1.cpp:36 */
#define K1C 0x1000003D1ULL

/* ------------------------------------------------------------------ helpers */
static int is_zero4(const uint64_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
static int ge4(const uint64_t a[4], const uint64_t b[4]) {
    for (int i = 3; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static uint64_t add4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a[i] + b[i];
        r[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
static uint64_t sub4(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t borrow = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)t;
        borrow = (uint64_t)(t >> 64) & 1;
    }
    return borrow;
}
/* full reduction of a value in [0,2^256) to [0,p) */
static void canon(uint64_t a[4]) {
    if (ge4(a, P)) sub4(a, a, P);
}

/* ------------------------------------------------------------------ field */
/* 512 -> 320 -> 256 fold of IntMod.cpp:926-942 / GPUMath.h:840-856 */
static void fold512(uint64_t r[4], const uint64_t w[8]) {
    /* t[0..4] = w[4..7] * 0x1000003D1 (320 bit) */
    uint64_t t[5];
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)w[4 + i] * K1C;
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    t[4] = (uint64_t)c;
    /* (lo, carry) = w[0..3] + t[0..3] */
    uint64_t lo[4];
    uint64_t carry = add4(lo, w, t);
    /* second fold: (t[4]+carry) * 0x1000003D1 -> (al, ah) */
    u128 f = (u128)(t[4] + carry) * K1C;
    uint64_t add[4] = {(uint64_t)f, (uint64_t)(f >> 64), 0, 0};
    add4(r, lo, add); /* carry discarded, no comparison with p */
}

void orc_modmul(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t w[8] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a[j] * b[i] + w[i + j];
            w[i + j] = (uint64_t)c;
            c >>= 64;
        }
        w[i + 4] = (uint64_t)c;
    }
    fold512(r, w);
}

void orc_modsqr(uint64_t r[4], const uint64_t a[4]) { orc_modmul(r, a, a); }

void orc_modsub(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    uint64_t t[4];
    if (sub4(t, a, b)) add4(t, t, P);
    memcpy(r, t, sizeof t);
}

static void sqr_n(uint64_t r[4], const uint64_t a[4], int n) {
    uint64_t t[4];
    memcpy(t, a, sizeof t);
    for (int i = 0; i < n; i++) orc_modsqr(t, t);
    memcpy(r, t, sizeof t);
}

/* a^(p-2) by the classic secp256k1 addition chain (255 squarings + 15 multiplications),
 * then full reduction: any algorithm returning the canonical inverse is bit-identical to
 * IntMod.cpp:560-565 / GPUMath.h:795-801. */
void orc_modinv(uint64_t r[4], const uint64_t a_in[4]) {
    uint64_t a[4], x2[4], x3[4], x6[4], x9[4], x11[4], x22[4], x44[4], x88[4], x176[4], x220[4],
        x223[4], t[4];
    memcpy(a, a_in, sizeof a);
    canon(a);
    if (is_zero4(a)) {
        memset(r, 0, 32);
        return;
    }
    orc_modsqr(x2, a);
    orc_modmul(x2, x2, a);
    orc_modsqr(x3, x2);
    orc_modmul(x3, x3, a);
    sqr_n(x6, x3, 3);
    orc_modmul(x6, x6, x3);
    sqr_n(x9, x6, 3);
    orc_modmul(x9, x9, x3);
    sqr_n(x11, x9, 2);
    orc_modmul(x11, x11, x2);
    sqr_n(x22, x11, 11);
    orc_modmul(x22, x22, x11);
    sqr_n(x44, x22, 22);
    orc_modmul(x44, x44, x22);
    sqr_n(x88, x44, 44);
    orc_modmul(x88, x88, x44);
    sqr_n(x176, x88, 88);
    orc_modmul(x176, x176, x88);
    sqr_n(x220, x176, 44);
    orc_modmul(x220, x220, x44);
    sqr_n(x223, x220, 3);
    orc_modmul(x223, x223, x3);
    sqr_n(t, x223, 23);
    orc_modmul(t, t, x22);
    sqr_n(t, t, 5);
    orc_modmul(t, t, a);
    sqr_n(t, t, 3);
    orc_modmul(t, t, x2);
    sqr_n(t, t, 2);
    orc_modmul(t, t, a);
    canon(t);
    memcpy(r, t, sizeof t);
}

/* IntGroup.cpp:36-57: subp[i] = v[0]*..*v[i]; inverse = 1/subp[n-1];
 * v[i] = subp[i-1]*inverse; inverse *= old v[i].  A zero anywhere zeroes everything. */
void orc_batch_inv(uint64_t (*v)[4], size_t n) {
    if (n == 0) return;
    uint64_t(*subp)[4] = malloc(n * sizeof *subp);
    uint64_t inv[4], nv[4];
    memcpy(subp[0], v[0], 32);
    for (size_t i = 1; i < n; i++) orc_modmul(subp[i], subp[i - 1], v[i]);
    orc_modinv(inv, subp[n - 1]);
    for (size_t i = n - 1; i > 0; i--) {
        orc_modmul(nv, subp[i - 1], inv);
        orc_modmul(inv, inv, v[i]);
        memcpy(v[i], nv, 32);
    }
    memcpy(v[0], inv, 32);
    free(subp);
}

/* ------------------------------------------------------------------ order */
void orc_add_order(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    /* IntMod.cpp:1245-1257: Add; Sub(_O); if negative Add(_O) -- on 320-bit signed Ints */
    uint64_t t[4];
    uint64_t c = add4(t, a, b);
    uint64_t br = sub4(t, t, ORDER);
    if (br > c) add4(t, t, ORDER); /* went negative */
    memcpy(r, t, sizeof t);
}
void orc_sub_order(uint64_t r[4], const uint64_t a[4], const uint64_t b[4]) {
    /* IntMod.cpp:1259-1263 */
    uint64_t t[4];
    if (sub4(t, a, b)) add4(t, t, ORDER);
    memcpy(r, t, sizeof t);
}

/* ------------------------------------------------------------------ curve */
void orc_add_direct(uint64_t rx[4], uint64_t ry[4], const uint64_t p1x[4], const uint64_t p1y[4],
                    const uint64_t p2x[4], const uint64_t p2y[4]) {
    /* SECP256K1.cpp:238-263 */
    uint64_t dy[4], dx[4], s[4], p[4], x3[4], y3[4];
    orc_modsub(dy, p2y, p1y);
    orc_modsub(dx, p2x, p1x);
    orc_modinv(dx, dx);
    orc_modmul(s, dy, dx);
    orc_modsqr(p, s);
    orc_modsub(x3, p, p1x);
    orc_modsub(x3, x3, p2x);
    orc_modsub(y3, p2x, x3);
    orc_modmul(y3, y3, s);
    orc_modsub(y3, y3, p2y);
    memcpy(rx, x3, 32);
    memcpy(ry, y3, 32);
}

static void dbl_affine(uint64_t rx[4], uint64_t ry[4], const uint64_t x[4], const uint64_t y[4]) {
    /* s = 3x^2 / 2y */
    uint64_t x2[4], num[4], den[4], s[4], p[4], x3[4], y3[4];
    orc_modsqr(x2, x);
    uint64_t zero[4] = {0};
    uint64_t t[4];
    orc_modsub(t, zero, x2); /* -x^2 */
    orc_modsub(num, x2, t);  /* 2x^2 */
    orc_modsub(num, num, t); /* 3x^2 */
    orc_modsub(t, zero, y);
    orc_modsub(den, y, t); /* 2y */
    orc_modinv(den, den);
    orc_modmul(s, num, den);
    orc_modsqr(p, s);
    orc_modsub(x3, p, x);
    orc_modsub(x3, x3, x);
    orc_modsub(y3, x, x3);
    orc_modmul(y3, y3, s);
    orc_modsub(y3, y3, y);
    canon(x3);
    canon(y3);
    memcpy(rx, x3, 32);
    memcpy(ry, y3, 32);
}

typedef struct {
    uint64_t x[4], y[4];
    int inf;
} apt_t;

static void apt_add(apt_t *r, const apt_t *a, const apt_t *b) {
    if (a->inf) {
        *r = *b;
        return;
    }
    if (b->inf) {
        *r = *a;
        return;
    }
    uint64_t ax[4], bx[4];
    memcpy(ax, a->x, 32);
    memcpy(bx, b->x, 32);
    canon(ax);
    canon(bx);
    if (memcmp(ax, bx, 32) == 0) {
        uint64_t ay[4], by[4];
        memcpy(ay, a->y, 32);
        memcpy(by, b->y, 32);
        canon(ay);
        canon(by);
        if (memcmp(ay, by, 32) == 0 && !is_zero4(ay)) {
            apt_t t;
            t.inf = 0;
            dbl_affine(t.x, t.y, ax, ay);
            *r = t;
        } else {
            memset(r, 0, sizeof *r);
            r->inf = 1;
        }
        return;
    }
    apt_t t;
    t.inf = 0;
    orc_add_direct(t.x, t.y, a->x, a->y, b->x, b->y);
    canon(t.x);
    canon(t.y);
    *r = t;
}

/* 32 x 255 window table of G, the shape of the reference's GTable (SECP256K1.cpp:40-58):
 * gtab[i][j] = (j+1) * 256^i * G */
static apt_t (*gtab)[255];
static pthread_once_t gtab_once = PTHREAD_ONCE_INIT; /* tests walk and verify herds from many threads */
static void gtab_build(void) {
    apt_t(*tab)[255] = malloc(32 * sizeof *tab);
    apt_t base;
    base.inf = 0;
    memcpy(base.x, GX, 32);
    memcpy(base.y, GY, 32);
    for (int i = 0; i < 32; i++) {
        tab[i][0] = base;
        for (int j = 1; j < 255; j++) apt_add(&tab[i][j], &tab[i][j - 1], &base);
        apt_add(&base, &tab[i][254], &base); /* 256 * base */
    }
    gtab = tab;
}
static void gtab_init(void) { pthread_once(&gtab_once, gtab_build); }

int orc_pubkey_add(uint64_t x[4], uint64_t y[4], const uint64_t k[4], const uint64_t qx[4],
                   const uint64_t qy[4]) {
    gtab_init();
    apt_t acc;
    memset(&acc, 0, sizeof acc);
    acc.inf = 1;
    if (qx) {
        acc.inf = 0;
        memcpy(acc.x, qx, 32);
        memcpy(acc.y, qy, 32);
    }
    for (int i = 0; i < 32; i++) {
        unsigned b = (unsigned)(k[i / 8] >> (8 * (i % 8))) & 0xFF;
        if (b) apt_add(&acc, &acc, &gtab[i][b - 1]);
    }
    if (acc.inf) {
        memset(x, 0, 32);
        memset(y, 0, 32);
        return -1;
    }
    memcpy(x, acc.x, 32);
    memcpy(y, acc.y, 32);
    return 0;
}

int orc_pubkey(uint64_t x[4], uint64_t y[4], const uint64_t k[4]) {
    return orc_pubkey_add(x, y, k, NULL, NULL);
}

/* ------------------------------------------------------------------ RNG */
static uint32_t mt[624];
static int mt_pos = 624;

void orc_rseed(uint32_t seed) {
    /* Random.cpp:37-51 */
    for (int pos = 0; pos < 624; pos++) {
        mt[pos] = seed;
        seed = 1812433253U * (seed ^ (seed >> 30)) + (uint32_t)pos + 1U;
    }
    mt_pos = 624;
}

uint32_t orc_rndl(void) {
    /* Random.cpp:66-101 */
    if (mt_pos == 624) {
        int i;
        uint32_t y;
        for (i = 0; i < 624 - 397; i++) {
            y = (mt[i] & 0x80000000U) | (mt[i + 1] & 0x7fffffffU);
            mt[i] = mt[i + 397] ^ (y >> 1) ^ ((0U - (y & 1U)) & 0x9908b0dfU);
        }
        for (; i < 623; i++) {
            y = (mt[i] & 0x80000000U) | (mt[i + 1] & 0x7fffffffU);
            mt[i] = mt[i + (397 - 624)] ^ (y >> 1) ^ ((0U - (y & 1U)) & 0x9908b0dfU);
        }
        y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
        mt[623] = mt[396] ^ (y >> 1) ^ ((0U - (y & 1U)) & 0x9908b0dfU);
        mt_pos = 0;
    }
    uint32_t y = mt[mt_pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680U;
    y ^= (y << 15) & 0xefc60000U;
    y ^= (y >> 18);
    return y;
}

void orc_int_rand(uint64_t r[4], int nbit) {
    /* Int.cpp:988-1001: nbit/32 full draws, then one more draw masked to nbit%32 bits
     * (a draw is consumed even when nbit%32 == 0). */
    uint32_t w[10] = {0};
    uint32_t nb = (uint32_t)nbit / 32, left = (uint32_t)nbit % 32;
    uint32_t mask = ((uint32_t)1 << left) - 1;
    uint32_t i = 0;
    for (; i < nb; i++) w[i] = orc_rndl();
    w[i] = orc_rndl() & mask;
    for (int k = 0; k < 4; k++) r[k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);
}

/* ------------------------------------------------------------------ params */
uint64_t orc_dp_mask(int dp) {
    if (dp == 0) return 0;
    if (dp > 64) dp = 64;
    if (dp == 64) return ~0ULL; /* (1ULL<<0)-1 = 0 -> ~0 */
    return ~((1ULL << (64 - dp)) - 1);
}

static double to_double4(const uint64_t a[4]) {
    return ldexp((double)a[3], 192) + ldexp((double)a[2], 128) + ldexp((double)a[1], 64) + (double)a[0];
}

void orc_jump_table(int range_power, uint64_t jd[ORC_NB_JUMP][2], uint64_t jx[ORC_NB_JUMP][4],
                    uint64_t jy[ORC_NB_JUMP][4], double *avg_log2) {
    int jump_bit = range_power / 2 + 1;
    if (jump_bit > 128) jump_bit = 128;
    int max_retry = 100, ok = 0;
    double max_avg = pow(2.0, (double)jump_bit - 0.95);
    double min_avg = pow(2.0, (double)jump_bit - 1.05);
    double dist_avg = 0;
    uint64_t dist[ORC_NB_JUMP][4];
    orc_rseed(0x600DCAFEU);
    while (!ok && max_retry > 0) {
        uint64_t total[4] = {0};
        for (int i = 0; i < ORC_NB_JUMP; i++) {
            orc_int_rand(dist[i], jump_bit);
            if (is_zero4(dist[i])) dist[i][0] = 1;
            add4(total, total, dist[i]);
        }
        dist_avg = to_double4(total) / (double)ORC_NB_JUMP;
        ok = dist_avg > min_avg && dist_avg < max_avg;
        max_retry--;
    }
    for (int i = 0; i < ORC_NB_JUMP; i++) {
        jd[i][0] = dist[i][0];
        jd[i][1] = dist[i][1];
        orc_pubkey(jx[i], jy[i], dist[i]);
    }
    if (avg_log2) *avg_log2 = log2(dist_avg);
}

/* ------------------------------------------------------------------ walks */
size_t orc_walk(uint64_t *x, uint64_t *y, uint64_t *d, size_t n, int nsteps,
                const uint64_t jd[ORC_NB_JUMP][2], const uint64_t jx[ORC_NB_JUMP][4],
                const uint64_t jy[ORC_NB_JUMP][4], uint64_t dpmask, orc_dp_t *dps, size_t dp_cap) {
    size_t found = 0;
    if (n == 0) return 0;
    uint64_t(*dx)[4] = malloc(n * sizeof *dx);
    for (int run = 0; run < nsteps; run++) {
        /* Kangaroo.cpp:379-391 / GPUCompute.h:52-61 */
        for (size_t g = 0; g < n; g++) {
            unsigned j = (unsigned)x[4 * g] & (ORC_NB_JUMP - 1);
            orc_modsub(dx[g], &x[4 * g], jx[j]);
        }
        /* Kangaroo.cpp:393-394 / GPUCompute.h:63 */
        orc_batch_inv(dx, n);
        /* Kangaroo.cpp:396-433 / GPUCompute.h:67-105 */
        for (size_t g = 0; g < n; g++) {
            uint64_t *px = &x[4 * g], *py = &y[4 * g], *pd = &d[2 * g];
            unsigned j = (unsigned)px[0] & (ORC_NB_JUMP - 1);
            uint64_t dy[4], s[4], p[4], rx[4], ry[4];
            orc_modsub(dy, py, jy[j]);
            orc_modmul(s, dy, dx[g]);
            orc_modsqr(p, s);
            orc_modsub(rx, p, jx[j]);
            orc_modsub(rx, rx, px);
            orc_modsub(ry, px, rx);
            orc_modmul(ry, ry, s);
            orc_modsub(ry, ry, py);
            memcpy(px, rx, 32);
            memcpy(py, ry, 32);
            /* GPUMath.h:119-121 Add128: raw 128-bit add */
            u128 dd = ((u128)pd[1] << 64 | pd[0]) + ((u128)jd[j][1] << 64 | jd[j][0]);
            pd[0] = (uint64_t)dd;
            pd[1] = (uint64_t)(dd >> 64);
            if ((px[3] & dpmask) == 0) {
                if (found < dp_cap && dps) {
                    memcpy(dps[found].x, px, 32);
                    dps[found].d[0] = pd[0];
                    dps[found].d[1] = pd[1];
                    dps[found].kidx = g;
                }
                found++;
            }
        }
    }
    free(dx);
    return found;
}

size_t orc_walk_direct(uint64_t *x, uint64_t *y, uint64_t *d4, size_t n, int nsteps,
                       const uint64_t jd[ORC_NB_JUMP][2], const uint64_t jx[ORC_NB_JUMP][4],
                       const uint64_t jy[ORC_NB_JUMP][4], uint64_t dpmask, orc_dp_t *dps,
                       size_t dp_cap) {
    size_t found = 0;
    for (int run = 0; run < nsteps; run++) {
        for (size_t i = 0; i < n; i++) {
            /* Check.cpp:535-548 */
            uint64_t *px = &x[4 * i], *py = &y[4 * i], *pd = &d4[4 * i];
            unsigned j = (unsigned)(px[0] % ORC_NB_JUMP);
            uint64_t rx[4], ry[4];
            orc_add_direct(rx, ry, px, py, jx[j], jy[j]);
            memcpy(px, rx, 32);
            memcpy(py, ry, 32);
            uint64_t jd4[4] = {jd[j][0], jd[j][1], 0, 0};
            orc_add_order(pd, pd, jd4);
            if ((px[3] & dpmask) == 0) {
                if (found < dp_cap && dps) {
                    memcpy(dps[found].x, px, 32);
                    dps[found].d[0] = pd[0];
                    dps[found].d[1] = pd[1];
                    dps[found].kidx = i;
                }
                found++;
            }
        }
    }
    return found;
}

void orc_create_herd(uint64_t *x, uint64_t *y, const uint64_t *d4, size_t n, int first_type,
                     const uint64_t kx[4], const uint64_t ky[4]) {
    /* Kangaroo.cpp:707-725: S = d*G ; tame -> S, wild -> keyToSearch + S */
    for (size_t j = 0; j < n; j++) {
        int wild = (int)((j + (size_t)first_type) % 2);
        if (wild)
            orc_pubkey_add(&x[4 * j], &y[4 * j], &d4[4 * j], kx, ky);
        else
            orc_pubkey(&x[4 * j], &y[4 * j], &d4[4 * j]);
    }
}
