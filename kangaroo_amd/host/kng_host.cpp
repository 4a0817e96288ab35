// kng_host.cpp -- host-side support library (see kng_host.h).  Product code, C++17 + __int128.
// Written from scratch; behaviour-compatible with the reference where a data format depends on
// it (jump table, DP mask, distance representation), each such place cites the reference.
#include "kng_host.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace {

typedef unsigned __int128 u128;

struct Fe {
    uint64_t v[4];
};

const Fe FP = {{0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL}};
const Fe FN = {{0xBFD25E8CD0364141ULL, 0xBAAEDCE6AF48A03BULL, 0xFFFFFFFFFFFFFFFEULL, 0xFFFFFFFFFFFFFFFFULL}};
const Fe GXc = {{0x59F2815B16F81798ULL, 0x029BFCDB2DCE28D9ULL, 0x55A06295CE870B07ULL, 0x79BE667EF9DCBBACULL}};
const Fe GYc = {{0x9C47D08FFB10D4B8ULL, 0xFD17B448A6855419ULL, 0x5DA4FBFC0E1108A8ULL, 0x483ADA7726A3C465ULL}};
const uint64_t KC = 0x1000003D1ULL;

inline bool is_zero(const Fe &a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
inline bool eq(const Fe &a, const Fe &b) { return !memcmp(a.v, b.v, 32); }
inline bool geq(const Fe &a, const Fe &b) {
    for (int i = 3; i >= 0; i--) {
        if (a.v[i] != b.v[i]) return a.v[i] > b.v[i];
    }
    return true;
}
inline uint64_t add_raw(Fe &r, const Fe &a, const Fe &b) {
    u128 c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)a.v[i] + b.v[i];
        r.v[i] = (uint64_t)c;
        c >>= 64;
    }
    return (uint64_t)c;
}
inline uint64_t sub_raw(Fe &r, const Fe &a, const Fe &b) {
    uint64_t br = 0;
    for (int i = 0; i < 4; i++) {
        u128 t = (u128)a.v[i] - b.v[i] - br;
        r.v[i] = (uint64_t)t;
        br = (uint64_t)(t >> 64) & 1;
    }
    return br;
}

// ---- field: always fully reduced results (host side has no bit-compat constraint on
// intermediate representation; the engine boundary only sees canonical coordinates) ----
inline Fe f_canon(Fe a) {
    if (geq(a, FP)) sub_raw(a, a, FP);
    return a;
}
inline Fe f_sub(const Fe &a, const Fe &b) {
    Fe r;
    if (sub_raw(r, a, b)) add_raw(r, r, FP);
    return r;
}
inline Fe f_add(const Fe &a, const Fe &b) {
    Fe r;
    uint64_t c = add_raw(r, a, b);
    if (c || geq(r, FP)) sub_raw(r, r, FP);
    return r;
}
inline Fe f_mul(const Fe &a, const Fe &b) {
    uint64_t w[8] = {0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)a.v[j] * b.v[i] + w[i + j];
            w[i + j] = (uint64_t)c;
            c >>= 64;
        }
        w[i + 4] = (uint64_t)c;
    }
    // 2^256 = KC (mod p): fold twice, then a final conditional subtraction
    u128 c = 0;
    uint64_t t[5];
    for (int i = 0; i < 4; i++) {
        c += (u128)w[4 + i] * KC;
        t[i] = (uint64_t)c;
        c >>= 64;
    }
    t[4] = (uint64_t)c;
    Fe r;
    c = 0;
    for (int i = 0; i < 4; i++) {
        c += (u128)w[i] + t[i];
        r.v[i] = (uint64_t)c;
        c >>= 64;
    }
    u128 f = (u128)(t[4] + (uint64_t)c) * KC;
    Fe add = {{(uint64_t)f, (uint64_t)(f >> 64), 0, 0}};
    if (add_raw(r, r, add)) { // wrapped past 2^256: add KC once more
        Fe k = {{KC, 0, 0, 0}};
        add_raw(r, r, k);
    }
    return f_canon(r);
}
inline Fe f_sqr(const Fe &a) { return f_mul(a, a); }
Fe f_sqrn(Fe a, int n) {
    while (n--) a = f_sqr(a);
    return a;
}
Fe f_inv(const Fe &a) { // a^(p-2), 255 squarings + 15 multiplications
    Fe x2 = f_mul(f_sqr(a), a), x3 = f_mul(f_sqr(x2), a), x6 = f_mul(f_sqrn(x3, 3), x3);
    Fe x9 = f_mul(f_sqrn(x6, 3), x3), x11 = f_mul(f_sqrn(x9, 2), x2), x22 = f_mul(f_sqrn(x11, 11), x11);
    Fe x44 = f_mul(f_sqrn(x22, 22), x22), x88 = f_mul(f_sqrn(x44, 44), x44), x176 = f_mul(f_sqrn(x88, 88), x88);
    Fe x220 = f_mul(f_sqrn(x176, 44), x44), x223 = f_mul(f_sqrn(x220, 3), x3);
    Fe t = f_mul(f_sqrn(x223, 23), x22);
    t = f_mul(f_sqrn(t, 5), a);
    t = f_mul(f_sqrn(t, 3), x2);
    return f_mul(f_sqrn(t, 2), a);
}

// ---- scalars mod n ----
inline Fe n_add(const Fe &a, const Fe &b) {
    Fe r;
    uint64_t c = add_raw(r, a, b);
    if (c || geq(r, FN)) sub_raw(r, r, FN);
    return r;
}
inline Fe n_sub(const Fe &a, const Fe &b) {
    Fe r;
    if (sub_raw(r, a, b)) add_raw(r, r, FN);
    return r;
}

// ---- affine points ----
struct Pt {
    Fe x, y;
    bool inf;
};
const Pt PINF = {{{0, 0, 0, 0}}, {{0, 0, 0, 0}}, true};

Pt pt_add(const Pt &a, const Pt &b) {
    if (a.inf) return b;
    if (b.inf) return a;
    Fe s;
    if (eq(a.x, b.x)) {
        if (!eq(a.y, b.y) || is_zero(a.y)) return PINF;
        Fe x2 = f_sqr(a.x);
        Fe num = f_add(f_add(x2, x2), x2);
        s = f_mul(num, f_inv(f_add(a.y, a.y)));
    } else {
        s = f_mul(f_sub(b.y, a.y), f_inv(f_sub(b.x, a.x)));
    }
    Pt r;
    r.inf = false;
    r.x = f_sub(f_sub(f_sqr(s), a.x), b.x);
    r.y = f_sub(f_mul(s, f_sub(a.x, r.x)), a.y);
    return r;
}

// 32 x 255 window table: gtab[w][b-1] = b * 256^w * G
std::vector<Pt> gtab;
std::once_flag gtab_once;
void gtab_build() {
    gtab.resize(32 * 255);
    Pt base = {GXc, GYc, false};
    for (int w = 0; w < 32; w++) {
        gtab[w * 255] = base;
        for (int b = 1; b < 255; b++) gtab[w * 255 + b] = pt_add(gtab[w * 255 + b - 1], base);
        base = pt_add(gtab[w * 255 + 254], base);
    }
}
const Pt &gt(int w, unsigned byte) {
    std::call_once(gtab_once, gtab_build);
    return gtab[w * 255 + (byte - 1)];
}
inline unsigned byte_of(const Fe &k, int w) { return (unsigned)(k.v[w / 8] >> (8 * (w % 8))) & 0xFF; }

Pt scalar_mul_g(const Fe &k, const Pt &start) {
    Pt acc = start;
    for (int w = 0; w < 32; w++) {
        unsigned b = byte_of(k, w);
        if (b) acc = pt_add(acc, gt(w, b));
    }
    return acc;
}

// ---- MT19937 as used by the reference (SECPK1/Random.cpp:31-101) ----
struct MT {
    uint32_t s[624];
    int pos;
    void seed(uint32_t v) {
        for (int i = 0; i < 624; i++) {
            s[i] = v;
            v = 1812433253U * (v ^ (v >> 30)) + (uint32_t)i + 1U;
        }
        pos = 624;
    }
    uint32_t next() {
        if (pos == 624) {
            for (int i = 0; i < 624; i++) {
                uint32_t y = (s[i] & 0x80000000U) | (s[(i + 1) % 624] & 0x7fffffffU);
                s[i] = s[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1U) ? 0x9908b0dfU : 0U);
            }
            pos = 0;
        }
        uint32_t y = s[pos++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680U;
        y ^= (y << 15) & 0xefc60000U;
        y ^= y >> 18;
        return y;
    }
    // Int::Rand(nbit), SECPK1/Int.cpp:988-1001: nbit/32 words + one masked word (always drawn)
    Fe rand_bits(int nbit) {
        uint32_t w[10] = {0};
        int nb = nbit / 32, left = nbit % 32;
        int i = 0;
        for (; i < nb; i++) w[i] = next();
        w[i] = next() & (uint32_t)(((uint64_t)1 << left) - 1);
        Fe r;
        for (int k = 0; k < 4; k++) r.v[k] = (uint64_t)w[2 * k] | ((uint64_t)w[2 * k + 1] << 32);
        return r;
    }
};

// ---- fast non-cryptographic stream for herd distances ----
struct Xo {
    uint64_t s[4];
    static uint64_t splitmix(uint64_t &z) {
        z += 0x9E3779B97F4A7C15ULL;
        uint64_t r = z;
        r = (r ^ (r >> 30)) * 0xBF58476D1CE4E5B9ULL;
        r = (r ^ (r >> 27)) * 0x94D049BB133111EBULL;
        return r ^ (r >> 31);
    }
    explicit Xo(uint64_t seed) {
        for (auto &w : s) w = splitmix(seed);
    }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() { // xoshiro256**
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0];
        s[3] ^= s[1];
        s[1] ^= s[2];
        s[0] ^= s[3];
        s[2] ^= t;
        s[3] = rotl(s[3], 45);
        return r;
    }
};

inline Fe load(const uint64_t *p) {
    Fe r;
    memcpy(r.v, p, 32);
    return r;
}
inline void store(uint64_t *p, const Fe &a) { memcpy(p, a.v, 32); }

// batched "acc[i] += T_i" over one chunk: one field inversion for the whole chunk
void batch_add(std::vector<Pt> &acc, const std::vector<const Pt *> &add, std::vector<Fe> &dx, std::vector<Fe> &pre) {
    const size_t m = acc.size();
    // special cases first (infinity / equal x): resolved one by one, marked by add[i] == nullptr
    std::vector<const Pt *> todo(add);
    for (size_t i = 0; i < m; i++) {
        if (!todo[i]) continue;
        if (acc[i].inf || eq(acc[i].x, todo[i]->x)) {
            acc[i] = pt_add(acc[i], *todo[i]);
            todo[i] = nullptr;
        }
    }
    Fe run = {{1, 0, 0, 0}};
    for (size_t i = 0; i < m; i++) {
        if (!todo[i]) continue;
        dx[i] = f_sub(todo[i]->x, acc[i].x);
        pre[i] = run;
        run = f_mul(run, dx[i]);
    }
    Fe inv = f_inv(run);
    for (size_t i = m; i-- > 0;) {
        if (!todo[i]) continue;
        Fe idx = f_mul(inv, pre[i]);
        inv = f_mul(inv, dx[i]);
        const Pt &b = *todo[i];
        Fe s = f_mul(f_sub(b.y, acc[i].y), idx);
        Fe x3 = f_sub(f_sub(f_sqr(s), acc[i].x), b.x);
        acc[i].y = f_sub(f_mul(s, f_sub(acc[i].x, x3)), acc[i].y);
        acc[i].x = x3;
    }
}

} // namespace

extern "C" {

int kngh_pubkey(const uint64_t k[4], uint64_t x[4], uint64_t y[4]) {
    Pt r = scalar_mul_g(load(k), PINF);
    if (r.inf) {
        memset(x, 0, 32);
        memset(y, 0, 32);
        return -1;
    }
    store(x, r.x);
    store(y, r.y);
    return 0;
}

int kngh_point_add(const uint64_t x1[4], const uint64_t y1[4], const uint64_t x2[4], const uint64_t y2[4], uint64_t x3[4],
                   uint64_t y3[4]) {
    Pt a = {f_canon(load(x1)), f_canon(load(y1)), false}, b = {f_canon(load(x2)), f_canon(load(y2)), false};
    Pt r = pt_add(a, b);
    if (r.inf) {
        memset(x3, 0, 32);
        memset(y3, 0, 32);
        return -1;
    }
    store(x3, r.x);
    store(y3, r.y);
    return 0;
}

int kngh_on_curve(const uint64_t x[4], const uint64_t y[4]) {
    Fe fx = f_canon(load(x)), fy = f_canon(load(y));
    Fe seven = {{7, 0, 0, 0}};
    return eq(f_sqr(fy), f_add(f_mul(f_sqr(fx), fx), seven)) ? 1 : 0;
}

void kngh_add_order(const uint64_t a[4], const uint64_t b[4], uint64_t r[4]) { store(r, n_add(load(a), load(b))); }
void kngh_sub_order(const uint64_t a[4], const uint64_t b[4], uint64_t r[4]) { store(r, n_sub(load(a), load(b))); }

uint64_t kngh_dp_mask(int dp) {
    if (dp <= 0) return 0;
    if (dp >= 64) return ~0ULL;
    return ~((1ULL << (64 - dp)) - 1);
}

double kngh_jump_table(int range_power, uint64_t *jd, uint64_t *jx, uint64_t *jy) {
    int jump_bit = range_power / 2 + 1;
    if (jump_bit > 128) jump_bit = 128;
    const double max_avg = std::pow(2.0, jump_bit - 0.95), min_avg = std::pow(2.0, jump_bit - 1.05);
    MT mt;
    mt.seed(0x600DCAFEU); // constant seed "for compatibility of workfiles" (Kangaroo.cpp:759-761)
    Fe dist[32];
    double avg = 0;
    bool ok = false;
    for (int retry = 100; !ok && retry > 0; retry--) {
        long double total = 0;
        for (int i = 0; i < 32; i++) {
            dist[i] = mt.rand_bits(jump_bit);
            if (is_zero(dist[i])) dist[i].v[0] = 1;
            total += std::ldexp((long double)dist[i].v[3], 192) + std::ldexp((long double)dist[i].v[2], 128) +
                     std::ldexp((long double)dist[i].v[1], 64) + (long double)dist[i].v[0];
        }
        // the reference sums exactly in an Int and converts once (Int::ToDouble); 32 values of
        // <= 128 bits summed in long double agree with that to far better than the window width
        avg = (double)(total / 32.0L);
        ok = avg > min_avg && avg < max_avg;
    }
    for (int i = 0; i < 32; i++) {
        jd[2 * i] = dist[i].v[0];
        jd[2 * i + 1] = dist[i].v[1];
        Pt p = scalar_mul_g(dist[i], PINF);
        store(jx + 4 * i, p.x);
        store(jy + 4 * i, p.y);
    }
    return std::log2(avg);
}

int kngh_suggest_dp(int range_power, double total_kangaroos) {
    // Kangaroo.cpp:980-988 with ComputeExpected (:836-873): overhead = (1 + k*2^dp/sqrt(N))^(1/3)
    int dp = (int)((double)range_power / 2.0 - std::log2(total_kangaroos));
    if (dp < 0) dp = 0;
    const double sqrtN = std::pow(2.0, range_power / 2.0);
    auto overhead = [&](int d) { return std::cbrt(1.0 + total_kangaroos * std::pow(2.0, d) / sqrtN); };
    while (overhead(dp) > 1.05 && dp > 0) dp--;
    return dp;
}

int kngh_create_herd(uint64_t n, int range_power, const uint64_t wild_offset[4], const uint64_t kx[4], const uint64_t ky[4],
                     int first_type, uint64_t seed, int nthreads, uint64_t *x, uint64_t *y, uint64_t *d_true) {
    if (range_power < 1 || range_power > 128 || !x || !y || !d_true) return -1;
    std::call_once(gtab_once, gtab_build);
    const Fe woff = load(wild_offset);
    // wild kangaroo: K + (dd - N/2)*G = (K - (N/2)*G) + dd*G with dd uniform in [0, 2^range_power)
    Pt wild_base = PINF;
    if (kx && ky) {
        Pt K = {f_canon(load(kx)), f_canon(load(ky)), false};
        Pt off = scalar_mul_g(woff, PINF);
        if (!off.inf) off.y = f_sub(Fe{{0, 0, 0, 0}}, off.y);
        wild_base = pt_add(K, off);
    }
    const int windows = (range_power + 7) / 8;
    unsigned hw = std::thread::hardware_concurrency();
    int T = nthreads > 0 ? nthreads : (hw ? (int)hw : 1);
    const uint64_t CH = 2048;
    const uint64_t nchunks = (n + CH - 1) / CH;
    if ((uint64_t)T > nchunks) T = (int)std::max<uint64_t>(1, nchunks);
    auto worker = [&](int tid) {
        std::vector<Pt> acc;
        std::vector<const Pt *> add;
        std::vector<Fe> dd, dx, pre;
        for (uint64_t c = tid; c < nchunks; c += T) {
            const uint64_t c0 = c * CH, m = std::min<uint64_t>(CH, n - c0);
            acc.assign(m, PINF);
            add.assign(m, nullptr);
            dd.resize(m);
            dx.resize(m);
            pre.resize(m);
            Xo rng(seed ^ (0xA24BAED4963EE407ULL * (c + 1)));
            for (uint64_t i = 0; i < m; i++) {
                Fe d = {{rng.next(), rng.next(), 0, 0}};
                if (range_power < 64) {
                    d.v[0] &= (1ULL << range_power) - 1;
                    d.v[1] = 0;
                } else if (range_power < 128) {
                    d.v[1] &= (range_power == 64) ? 0 : ((1ULL << (range_power - 64)) - 1);
                }
                dd[i] = d;
                const bool wild = ((c0 + i + (uint64_t)first_type) & 1) != 0;
                acc[i] = wild ? wild_base : PINF;
                store(d_true + 4 * (c0 + i), wild ? n_sub(d, woff) : d);
            }
            for (int w = 0; w < windows; w++) {
                for (uint64_t i = 0; i < m; i++) {
                    unsigned b = byte_of(dd[i], w);
                    add[i] = b ? &gt(w, b) : nullptr;
                }
                batch_add(acc, add, dx, pre);
            }
            for (uint64_t i = 0; i < m; i++) {
                // a zero distance on a tame kangaroo would be the point at infinity: bump it to 1*G
                if (acc[i].inf) {
                    acc[i] = gt(0, 1);
                    Fe one = {{1, 0, 0, 0}};
                    store(d_true + 4 * (c0 + i), one);
                }
                store(x + 4 * (c0 + i), acc[i].x);
                store(y + 4 * (c0 + i), acc[i].y);
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++) th.emplace_back(worker, t);
    worker(0);
    for (auto &t : th) t.join();
    return 0;
}

int kngh_herd_params(int range_power, const uint64_t wild_offset[4], const uint64_t kx[4], const uint64_t ky[4],
                     uint64_t seed, uint64_t *table, uint64_t base_tame[8], uint64_t base_wild[8],
                     uint64_t final_add[8]) {
    if (range_power < 1 || range_power > 128 || !table || !base_tame || !base_wild || !final_add) return -1;
    std::call_once(gtab_once, gtab_build);
    const int windows = (range_power + 7) / 8;
    memset(table, 0, (size_t)windows * 256 * 8 * sizeof(uint64_t));
    for (int w = 0; w < windows; w++)
        for (unsigned v = 1; v < 256; v++) {
            const Pt &p = gt(w, v);
            store(table + ((size_t)w * 256 + v) * 8, p.x);
            store(table + ((size_t)w * 256 + v) * 8 + 4, p.y);
        }
    // b: 127 random bits (never 0), far above any distance so b + dd never wraps or hits 0 mod n
    Xo rng(seed ^ 0x5851F42D4C957F2DULL);
    Fe b = {{rng.next() | 1, rng.next() >> 1, 0, 0}};
    Pt bg = scalar_mul_g(b, PINF);
    Pt wild = PINF;
    if (kx && ky) {
        Pt K = {f_canon(load(kx)), f_canon(load(ky)), false};
        Pt off = scalar_mul_g(load(wild_offset), PINF);
        if (!off.inf) off.y = f_sub(Fe{{0, 0, 0, 0}}, off.y);
        wild = pt_add(pt_add(K, off), bg);
    } else {
        wild = bg;
    }
    if (wild.inf || bg.inf) return -1;
    store(base_tame, bg.x);
    store(base_tame + 4, bg.y);
    store(base_wild, wild.x);
    store(base_wild + 4, wild.y);
    store(final_add, bg.x);
    store(final_add + 4, f_sub(Fe{{0, 0, 0, 0}}, bg.y));
    return 0;
}

int kngh_to_device_distances(const uint64_t *d_true, uint64_t n, const uint64_t wild_offset[4], uint64_t *d_dev) {
    const Fe woff = load(wild_offset);
    for (uint64_t i = 0; i < n; i++) {
        Fe d = load(d_true + 4 * i);
        if (i & 1) d = n_add(d, woff); // GPUEngine.cu:409
        if (d.v[2] | d.v[3]) return -1;
        d_dev[2 * i] = d.v[0];
        d_dev[2 * i + 1] = d.v[1];
    }
    return 0;
}

void kngh_to_true_distances(const uint64_t *d_dev, const uint64_t *kidx, uint64_t n, const uint64_t wild_offset[4],
                            uint64_t *d_true) {
    const Fe woff = load(wild_offset);
    for (uint64_t i = 0; i < n; i++) {
        Fe d = {{d_dev[2 * i], d_dev[2 * i + 1], 0, 0}};
        const uint64_t k = kidx ? kidx[i] : i;
        if (k & 1) d = n_sub(d, woff); // GPUEngine.cu:477,672
        store(d_true + 4 * i, d);
    }
}

} // extern "C"
