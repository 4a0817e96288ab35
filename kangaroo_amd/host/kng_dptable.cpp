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
#include "kng_bucket.h"
#include "kng_host.h"

namespace {
using namespace kng_arena;
using namespace kng_bucket;
constexpr unsigned ARENA_BITS = 6, N_ARENAS = 1u << ARENA_BITS;
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

void kngt_prefetch(const kngt_table *t, uint32_t h, uint64_t x1, int stage) { kng_bucket::prefetch(bucket_of(t, h), x1, stage); }

int kngt_add_entry(kngt_table *t, uint32_t h, const kngt_entry *e, kngt_entry *other) {
    Bucket &b = bucket_of(t, h);
    // the reference's allocation bookkeeping, reproduced for the file format: first use -> 16, and a
    // +4 step whenever the bucket is within one slot of full at the START of an add (even one that
    // ends as DUPLICATE/COLLISION)
    if (b.ref_max == 0) b.ref_max = 16;
    if (b.n && b.n >= b.ref_max - 1) b.ref_max += 4;
    return add_entry(arena_of(t, h), b, e, other);
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
