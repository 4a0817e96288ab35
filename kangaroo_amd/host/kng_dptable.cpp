// kng_dptable.cpp -- see kng_dptable.h.  Product code (host, no GPU needed).
//
// Storage.  A bucket of the file format (18 bits of x.limb2) is kept sorted by (x.limb1, x.limb0).  Eight GPUs at the
// DP size the reference suggests for them deliver ~85 M points per second and 2^30 points per solved 80-bit key:
// 4096 entries per bucket, where inserting into ONE sorted array would move 64 KB per point.  Each bucket is
// therefore a row of 2^k "fine" arrays selected by the TOP k bits of x.limb1 -- the most significant bits of the sort
// key, so the fine arrays, read in index order, ARE the sorted bucket -- and k grows by 2 whenever the bucket holds
// more than 32 entries per fine array (a local re-split by the one thread that owns the bucket).  An insertion then
// moves ~0.3 KB whatever the size of the table; the serialised form is unchanged.
//
// Memory.  The table owns its memory (kng_arena.h says why): every bucket belongs to one of 64 arenas (by a scramble of its
// index, the same one the solver uses to assign buckets to consumers, so an arena is shared by at most two threads; a spin
// lock covers that).  Nothing goes back to the OS before kngt_reset.
#include "kng_dptable.h"

#include <sys/stat.h>

#include <atomic>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "kng_arena.h"
#include "kng_host.h"

namespace {

struct Fine {
    kngt_entry *e = nullptr;
    uint32_t n = 0, cap = 0;
};

using namespace kng_arena;
constexpr unsigned ARENA_BITS = 6, N_ARENAS = 1u << ARENA_BITS;

struct Bucket {
    Fine *fine = nullptr; // 1 << k arrays, ordered by the top k bits of x[1]
    uint32_t n = 0;       // entries in the whole bucket (nbItem)
    uint32_t ref_max = 0; // the reference's maxItem bookkeeping (file compatibility only)
    uint32_t split_retry = 0; // a re-split that failed (no memory for the finer layout) is tried again once n reaches this
    uint8_t k = 0;
};

constexpr uint64_t D_MASK = 0x3FFFFFFFFFFFFFFFULL;
constexpr uint64_t D_SIGN = 1ULL << 63, D_TYPE = 1ULL << 62;
// entries per fine array that trigger a re-split into four times as many: runs then hold 2..8 entries (64..256 B).  Round 3 used
// 32 (runs of 8..32): at 60-80 M entries an insertion cost 68-72 ns against 53-61 ns with 8, and 60 against 45 bytes of memory
// per entry (less slack in short runs outweighs more 16-byte run headers); 16 table threads took 150 -> 182 M points/s
// (profiles/r04_dp_probe_split.txt).  KNGT_SPLIT_AVG overrides it for measurements.
static uint32_t split_avg_knob() {
    const char *e = getenv("KNGT_SPLIT_AVG");
    const long v = e ? atol(e) : 8;
    return (uint32_t)(v < 2 ? 2 : v > 4096 ? 4096 : v); // k_for() halves it: below 2 every bucket would split without end
}
static const uint32_t SPLIT_AVG = split_avg_knob();
static const bool GROW2 = getenv("KNGT_GROW2") && atoi(getenv("KNGT_GROW2")); // runs grow by doubling instead of by size class (measurement knob)
constexpr uint8_t K_MAX = 24;

inline int cmp_x(const uint64_t a[2], const uint64_t b[2]) {
    if (a[1] != b[1]) return a[1] > b[1] ? 1 : -1;
    if (a[0] != b[0]) return a[0] > b[0] ? 1 : -1;
    return 0;
}
inline size_t fine_index(uint8_t k, uint64_t x1) { return k ? (size_t)(x1 >> (64 - k)) : 0; }

inline uint32_t cap_of_class(int c) { return (uint32_t)(class_bytes(c) / sizeof(kngt_entry)); }

bool reserve(Arena &a, Fine &f, uint32_t want) {
    if (want <= f.cap) return true;
    const int c = class_of((size_t)want * sizeof(kngt_entry));
    kngt_entry *p = static_cast<kngt_entry *>(arena_alloc(a, c));
    if (!p) return false;
    if (f.n) std::memcpy(p, f.e, (size_t)f.n * sizeof(kngt_entry));
    if (f.e) arena_free(a, f.e, class_of((size_t)f.cap * sizeof(kngt_entry)));
    f.e = p;
    f.cap = cap_of_class(c);
    return true;
}

inline int fine_class(uint8_t k) { return class_of(((size_t)1 << k) * sizeof(Fine)); }

void free_bucket(Arena &a, Bucket &b) {
    if (b.fine) {
        const size_t nf = (size_t)1 << b.k;
        for (size_t i = 0; i < nf; i++)
            if (b.fine[i].e) arena_free(a, b.fine[i].e, class_of((size_t)b.fine[i].cap * sizeof(kngt_entry)));
        arena_free(a, b.fine, fine_class(b.k));
    }
    b.fine = nullptr;
    b.k = 0;
}

// lay `n` entries (sorted) out over 1 << k fine arrays; the bucket must be empty of storage
bool build(Arena &a, Bucket &b, uint8_t k, const kngt_entry *sorted, uint32_t n) {
    const size_t nf = (size_t)1 << k;
    Fine *fine = static_cast<Fine *>(arena_alloc(a, fine_class(k)));
    if (!fine) return false;
    for (size_t q = 0; q < nf; q++) fine[q] = Fine();
    b.fine = fine;
    b.k = k;
    uint32_t i = 0;
    while (i < n) {
        const size_t fi = fine_index(k, sorted[i].x[1]);
        uint32_t j = i + 1;
        while (j < n && fine_index(k, sorted[j].x[1]) == fi) j++;
        Fine &f = fine[fi];
        if (!reserve(a, f, j - i + 2)) {
            free_bucket(a, b);
            return false;
        }
        std::memcpy(f.e, sorted + i, (size_t)(j - i) * sizeof(kngt_entry));
        f.n = j - i;
        i = j;
    }
    return true;
}

// all entries of the bucket, in order
void gather(const Bucket &b, kngt_entry *out) {
    if (!b.fine) return;
    const size_t nf = (size_t)1 << b.k;
    for (size_t i = 0; i < nf; i++) {
        if (b.fine[i].n) std::memcpy(out, b.fine[i].e, (size_t)b.fine[i].n * sizeof(kngt_entry));
        out += b.fine[i].n;
    }
}

bool resplit(Arena &a, Bucket &b, uint8_t k, std::vector<kngt_entry> &scratch) {
    scratch.resize(b.n);
    gather(b, scratch.data());
    Bucket nb;
    if (!build(a, nb, k, scratch.data(), b.n)) return false; // keep the old layout: still correct, only slower
    nb.n = b.n;
    nb.ref_max = b.ref_max;
    nb.split_retry = 0;
    free_bucket(a, b);
    b = nb;
    return true;
}

uint8_t k_for(uint32_t n) {
    uint8_t k = 0;
    while (k < K_MAX && n > ((SPLIT_AVG / 2) << k)) k += 2;
    return k;
}

} // namespace

struct kngt_table {
    // Bucket headers are stored in SCRAMBLED order, slot(h) = the bijection of the 18-bit bucket index that kng_solver's
    // consumer_of() cuts into equal ranges: a consumer's buckets are then one contiguous stretch of this array (190 KB of
    // 24-byte headers at 32 consumers: they live in its L2) instead of being interleaved line by line with every other
    // consumer's.  Round 3 stored them in file order: each 64-byte line held headers of two or three different owners,
    // every insertion wrote one (n, ref_max), and the lines bounced between cores and sockets -- consumers slowed from
    // 75 ns per point alone to 130 / 240 / 370 ns at 16 / 32 / 64 threads (profiles/r04_dp_host_before.txt).
    Bucket b[KNGT_BUCKETS];
    Arena arena[N_ARENAS];
};

namespace {
inline uint32_t slot_of(uint32_t h) { return ((h & (KNGT_BUCKETS - 1)) * 0x9E3779B1u) & (KNGT_BUCKETS - 1); }
inline Bucket &bucket_of(kngt_table *t, uint32_t h) { return t->b[slot_of(h)]; }
inline const Bucket &bucket_of(const kngt_table *t, uint32_t h) { return t->b[slot_of(h)]; }
// the arenas follow the same order: the buckets of an arena belong to one consumer, or to two neighbours
inline Arena &arena_of(kngt_table *t, uint32_t h) { return t->arena[slot_of(h) >> (KNGT_HASH_BITS - ARENA_BITS)]; }
} // namespace

extern "C" {

kngt_table *kngt_create(void) { return new (std::nothrow) kngt_table(); }

void kngt_reset(kngt_table *t) {
    if (!t) return;
    for (Bucket &b : t->b) b = Bucket();
    for (Arena &a : t->arena) arena_release(a);
}

void kngt_destroy(kngt_table *t) {
    if (!t) return;
    kngt_reset(t);
    delete t;
}

void kngt_encode(const uint64_t x[4], const uint64_t d_true[4], uint32_t type, uint32_t *bucket, kngt_entry *e) {
    e->x[0] = x[0];
    e->x[1] = x[1];
    uint64_t sign = 0;
    if (d_true[3] > 0x7FFFFFFFFFFFFFFFULL) { // "negative": store n - d with the sign bit
        const uint64_t zero[4] = {0, 0, 0, 0};
        uint64_t neg[4];
        kngh_sub_order(zero, d_true, neg);
        e->d[0] = neg[0];
        e->d[1] = neg[1] & D_MASK;
        sign = D_SIGN;
    } else {
        e->d[0] = d_true[0];
        e->d[1] = d_true[1] & D_MASK;
    }
    e->d[1] |= sign | ((uint64_t)(type & 1) << 62);
    *bucket = (uint32_t)(x[2] & (KNGT_BUCKETS - 1));
}

void kngt_encode_device(const uint64_t x[4], const uint64_t d_dev[2], const uint64_t wild_offset[2], uint64_t kidx,
                        uint32_t *bucket, kngt_entry *e) {
    typedef unsigned __int128 u128;
    const uint64_t type = kidx & 1;
    u128 d = ((u128)d_dev[1] << 64) | d_dev[0];
    uint64_t sign = 0;
    if (type) { // (d - offset) mod n is "negative" exactly when d < offset; its magnitude is then offset - d
        const u128 off = ((u128)wild_offset[1] << 64) | wild_offset[0];
        if (d >= off) {
            d -= off;
        } else {
            d = off - d;
            sign = D_SIGN;
        }
    }
    e->x[0] = x[0];
    e->x[1] = x[1];
    e->d[0] = (uint64_t)d;
    e->d[1] = ((uint64_t)(d >> 64) & D_MASK) | sign | (type << 62);
    *bucket = (uint32_t)(x[2] & (KNGT_BUCKETS - 1));
}

void kngt_decode(const uint64_t d_word[2], uint64_t d_true[4], uint32_t *type) {
    if (type) *type = (d_word[1] & D_TYPE) ? 1 : 0;
    uint64_t v[4] = {d_word[0], d_word[1] & D_MASK, 0, 0};
    if (d_word[1] & D_SIGN) {
        const uint64_t zero[4] = {0, 0, 0, 0};
        kngh_sub_order(zero, v, d_true);
    } else {
        std::memcpy(d_true, v, 32);
    }
}

void kngt_prefetch(const kngt_table *t, uint32_t h, uint64_t x1, int stage) {
    const Bucket &b = bucket_of(t, h);
    if (stage == 0) {
        __builtin_prefetch(&b);
    } else if (b.fine) {
        const Fine *f = &b.fine[fine_index(b.k, x1)];
        if (stage == 1) {
            __builtin_prefetch(f);
        } else if (f->e) { // the whole run: the search reads a few of its lines, the insertion shifts the rest
            static const int lines = getenv("KNGT_PF_LINES") ? atoi(getenv("KNGT_PF_LINES")) : 12; // (measurement knob)
            const char *p = reinterpret_cast<const char *>(f->e), *end = p + (size_t)(f->n + 1) * sizeof(kngt_entry);
            for (int i = 0; i < lines && p < end; i++, p += 64) __builtin_prefetch(p, 1);
        }
    }
}

int kngt_add_entry(kngt_table *t, uint32_t h, const kngt_entry *e, kngt_entry *other) {
    Bucket &b = bucket_of(t, h);
    // the reference's allocation bookkeeping, reproduced for the file format: first use -> 16, and a
    // +4 step whenever the bucket is within one slot of full at the START of an add (even one that
    // ends as DUPLICATE/COLLISION)
    if (b.ref_max == 0) b.ref_max = 16;
    Arena &ar = arena_of(t, h);
    if (!b.fine && !build(ar, b, 0, nullptr, 0)) return -1;
    if (b.n && b.n >= b.ref_max - 1) b.ref_max += 4;

    Fine &f = b.fine[fine_index(b.k, e->x[1])];
    uint32_t lo = 0, hi = f.n; // first position with x >= e->x
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (cmp_x(f.e[mid].x, e->x) < 0) lo = mid + 1;
        else hi = mid;
    }
    if (lo < f.n && cmp_x(f.e[lo].x, e->x) == 0) {
        if (f.e[lo].d[0] == e->d[0] && f.e[lo].d[1] == e->d[1]) return KNGT_ADD_DUPLICATE;
        if (other) *other = f.e[lo];
        return KNGT_ADD_COLLISION;
    }
    if (b.n == 0xFFFFFFFFu) return -1; // nbItem is a 32-bit word of the file format
    if (!reserve(ar, f, (GROW2 && f.n + 1 > f.cap && f.cap >= 4) ? 2 * f.cap : f.n + 1)) return -1;
    std::memmove(f.e + lo + 1, f.e + lo, (size_t)(f.n - lo) * sizeof(kngt_entry));
    f.e[lo] = *e;
    f.n++;
    b.n++;
    if (b.k < K_MAX && b.n > (SPLIT_AVG << b.k) && b.n >= b.split_retry) {
        thread_local std::vector<kngt_entry> scratch;
        // a failed re-split keeps the old (correct, slower) layout; gathering the whole bucket again on every insert
        // would make each one cost O(n): try again when the bucket has doubled
        if (resplit(ar, b, (uint8_t)(b.k + 2), scratch)) b.split_retry = 0;
        else b.split_retry = b.n > 0x7FFFFFFFu ? 0xFFFFFFFFu : 2 * b.n;
    }
    return KNGT_ADD_OK;
}

int kngt_add(kngt_table *t, const uint64_t x[4], const uint64_t d_true[4], uint32_t type, uint64_t other_d[4],
             uint32_t *other_type) {
    uint32_t h;
    kngt_entry e, o;
    kngt_encode(x, d_true, type, &h, &e);
    const int st = kngt_add_entry(t, h, &e, &o);
    if (st == KNGT_ADD_COLLISION && other_d) kngt_decode(o.d, other_d, other_type);
    return st;
}

uint64_t kngt_count(const kngt_table *t) {
    uint64_t c = 0;
    for (const Bucket &b : t->b) c += b.n;
    return c;
}

uint32_t kngt_bucket_count(const kngt_table *t, uint32_t bucket) { return bucket_of(t, bucket).n; }

uint32_t kngt_bucket_entries(const kngt_table *t, uint32_t bucket, kngt_entry *out, uint32_t cap) {
    const Bucket &b = bucket_of(t, bucket);
    if (b.n <= cap) {
        gather(b, out);
        return b.n;
    }
    std::vector<kngt_entry> all(b.n);
    gather(b, all.data());
    if (cap) std::memcpy(out, all.data(), (size_t)cap * sizeof(kngt_entry));
    return cap;
}

uint64_t kngt_serialised_size(const kngt_table *t) { return (uint64_t)KNGT_BUCKETS * 8 + kngt_count(t) * 32; }

uint64_t kngt_memory_bytes(const kngt_table *t) {
    uint64_t m = sizeof(kngt_table);
    for (const Arena &a : t->arena) m += arena_touched(a);
    return m;
}

int kngt_write(const kngt_table *t, FILE *f) {
    for (uint32_t h = 0; h < KNGT_BUCKETS; h++) { // file order = bucket index order (HashTable.cpp:375-396)
        const Bucket &b = bucket_of(t, h);
        const uint32_t head[2] = {b.n, b.ref_max};
        if (std::fwrite(head, 4, 2, f) != 2) return -1;
        if (!b.fine) continue;
        const size_t nf = (size_t)1 << b.k;
        for (size_t i = 0; i < nf; i++)
            if (b.fine[i].n && std::fwrite(b.fine[i].e, sizeof(kngt_entry), b.fine[i].n, f) != b.fine[i].n) return -1;
    }
    return 0;
}

static int read_buckets(kngt_table *t, FILE *f) {
    // a bucket cannot announce more entries than the file has bytes left (a truncated or corrupt file must fail,
    // not allocate); unknown size (a pipe) falls back to the 32-bit word itself
    uint64_t remaining = UINT64_MAX;
    struct stat sb;
    const off_t pos = ftello(f);
    if (pos >= 0 && fstat(fileno(f), &sb) == 0 && S_ISREG(sb.st_mode) && (uint64_t)sb.st_size >= (uint64_t)pos)
        remaining = (uint64_t)sb.st_size - (uint64_t)pos;
    std::vector<kngt_entry> buf;
    for (uint32_t h = 0; h < KNGT_BUCKETS; h++) {
        Bucket &b = bucket_of(t, h);
        uint32_t head[2];
        if (std::fread(head, 4, 2, f) != 2) return -1;
        if (remaining != UINT64_MAX) remaining -= remaining < 8 ? remaining : 8;
        const uint32_t n = head[0];
        if ((uint64_t)n * sizeof(kngt_entry) > remaining) return -1;
        b.ref_max = head[1];
        if (!n) continue;
        buf.resize(n);
        if (std::fread(buf.data(), sizeof(kngt_entry), n, f) != n) return -1;
        if (remaining != UINT64_MAX) remaining -= (uint64_t)n * sizeof(kngt_entry);
        // Add() binary-searches the bucket: it must be strictly ascending in (x.limb1, x.limb0) as the reference writes it
        for (uint32_t i = 1; i < n; i++)
            if (cmp_x(buf[i - 1].x, buf[i].x) >= 0) return -1;
        if (!build(arena_of(t, h), b, k_for(n), buf.data(), n)) return -1;
        b.n = n;
    }
    return 0;
}

int kngt_read(kngt_table *t, FILE *f) {
    kngt_reset(t);
    const int rc = read_buckets(t, f);
    if (rc != 0) kngt_reset(t); // never leave half a table behind a failed load
    return rc;
}

} // extern "C"
