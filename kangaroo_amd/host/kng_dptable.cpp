// kng_dptable.cpp -- see kng_dptable.h.  Product code (host, no GPU needed).
#include "kng_dptable.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include "kng_host.h"

namespace {

struct Bucket {
    kngt_entry *e = nullptr;
    uint32_t n = 0;       // entries stored (nbItem)
    uint32_t cap = 0;     // entries allocated
    uint32_t ref_max = 0; // the reference's maxItem bookkeeping (file compatibility only)
};

constexpr uint64_t D_MASK = 0x3FFFFFFFFFFFFFFFULL;
constexpr uint64_t D_SIGN = 1ULL << 63, D_TYPE = 1ULL << 62;

inline int cmp_x(const uint64_t a[2], const uint64_t b[2]) {
    if (a[1] != b[1]) return a[1] > b[1] ? 1 : -1;
    if (a[0] != b[0]) return a[0] > b[0] ? 1 : -1;
    return 0;
}

bool reserve(Bucket &b, uint32_t want) {
    if (want <= b.cap) return true;
    uint32_t cap = b.cap ? b.cap : 8;
    while (cap < want) cap += cap / 2 + 4;
    void *p = std::realloc(b.e, (size_t)cap * sizeof(kngt_entry));
    if (!p) return false;
    b.e = static_cast<kngt_entry *>(p);
    b.cap = cap;
    return true;
}

} // namespace

struct kngt_table {
    Bucket b[KNGT_BUCKETS];
};

extern "C" {

kngt_table *kngt_create(void) { return new (std::nothrow) kngt_table(); }

void kngt_reset(kngt_table *t) {
    if (!t) return;
    for (Bucket &b : t->b) {
        std::free(b.e);
        b = Bucket();
    }
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

int kngt_add_entry(kngt_table *t, uint32_t h, const kngt_entry *e, kngt_entry *other) {
    Bucket &b = t->b[h & (KNGT_BUCKETS - 1)];
    // the reference's allocation bookkeeping, reproduced for the file format: first use -> 16, and a
    // +4 step whenever the bucket is within one slot of full at the START of an add (even one that
    // ends as DUPLICATE/COLLISION)
    if (b.ref_max == 0) b.ref_max = 16;
    if (b.n == 0) {
        if (!reserve(b, 1)) return -1;
        b.e[0] = *e;
        b.n = 1;
        return KNGT_ADD_OK;
    }
    if (b.n >= b.ref_max - 1) b.ref_max += 4;

    uint32_t lo = 0, hi = b.n; // first position with x >= e->x
    while (lo < hi) {
        const uint32_t mid = lo + (hi - lo) / 2;
        if (cmp_x(b.e[mid].x, e->x) < 0) lo = mid + 1;
        else hi = mid;
    }
    if (lo < b.n && cmp_x(b.e[lo].x, e->x) == 0) {
        if (b.e[lo].d[0] == e->d[0] && b.e[lo].d[1] == e->d[1]) return KNGT_ADD_DUPLICATE;
        if (other) *other = b.e[lo];
        return KNGT_ADD_COLLISION;
    }
    if (!reserve(b, b.n + 1)) return -1;
    std::memmove(b.e + lo + 1, b.e + lo, (size_t)(b.n - lo) * sizeof(kngt_entry));
    b.e[lo] = *e;
    b.n++;
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

uint32_t kngt_bucket_count(const kngt_table *t, uint32_t bucket) { return t->b[bucket & (KNGT_BUCKETS - 1)].n; }

uint32_t kngt_bucket_entries(const kngt_table *t, uint32_t bucket, kngt_entry *out, uint32_t cap) {
    const Bucket &b = t->b[bucket & (KNGT_BUCKETS - 1)];
    const uint32_t n = b.n < cap ? b.n : cap;
    if (n) std::memcpy(out, b.e, (size_t)n * sizeof(kngt_entry));
    return n;
}

uint64_t kngt_serialised_size(const kngt_table *t) { return (uint64_t)KNGT_BUCKETS * 8 + kngt_count(t) * 32; }

int kngt_write(const kngt_table *t, FILE *f) {
    for (const Bucket &b : t->b) {
        const uint32_t head[2] = {b.n, b.ref_max};
        if (std::fwrite(head, 4, 2, f) != 2) return -1;
        if (b.n && std::fwrite(b.e, sizeof(kngt_entry), b.n, f) != b.n) return -1;
    }
    return 0;
}

int kngt_read(kngt_table *t, FILE *f) {
    kngt_reset(t);
    for (Bucket &b : t->b) {
        uint32_t head[2];
        if (std::fread(head, 4, 2, f) != 2) return -1;
        b.ref_max = head[1];
        if (head[0]) {
            if (!reserve(b, head[0])) return -1;
            if (std::fread(b.e, sizeof(kngt_entry), head[0], f) != head[0]) return -1;
            b.n = head[0];
        }
    }
    return 0;
}

} // extern "C"
