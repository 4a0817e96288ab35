// kng_bucket.h -- how both distinguished-point tables keep ONE bucket of the file format (kng_dptable.cpp: the repo's own
// table; HashTable_kng.cpp: `class HashTable` of the reference, HashTable.h:66-108, for link-time replacement).
//
// A bucket (18 bits of x.limb2, HashTable.h:27-56) is sorted by (x.limb1, x.limb0) in the file and in the reference's memory,
// where it is an array of pointers to malloc'ed 32-byte entries: an insertion is a binary search through cold heap lines and a
// memmove of half the pointer array (HashTable.cpp:262-324).  Here the ENTRIES THEMSELVES lie in short sorted runs: a bucket
// is a row of 2^k "fine" arrays selected by the TOP k bits of x.limb1 -- the most significant bits of the sort key, so the
// fine arrays read in index order ARE the sorted bucket -- and k grows by 2 whenever the bucket holds more than SPLIT_AVG
// entries per fine array (a local re-split by whoever holds the bucket).  An insertion touches one run header and one run of
// 2..8 entries (64..256 B) whatever the size of the table; the serialised form is unchanged.  No pointer array, no per-entry
// allocation: 45-47 bytes of memory per point.
//
// Nothing in here locks a bucket: the callers do (kng_dptable: a bucket belongs to one consumer thread; HashTable_kng: 1024
// stripe locks).  The arena takes its own lock for allocations.
#ifndef KNG_BUCKET_H
#define KNG_BUCKET_H

#include <cstdlib>
#include <cstring>
#include <vector>

#include "kng_arena.h"
#include "kng_dptable.h" /* kngt_entry = the 32 bytes of the file format, KNGT_ADD_* = HashTable.h:33-35 */

namespace kng_bucket {

struct Fine {
    kngt_entry *e = nullptr;
    uint32_t n = 0, cap = 0;
};

using namespace kng_arena;

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
inline uint32_t split_avg_knob() {
    const char *e = getenv("KNGT_SPLIT_AVG");
    const long v = e ? atol(e) : 8;
    return (uint32_t)(v < 2 ? 2 : v > 4096 ? 4096 : v); // k_for() halves it: below 2 every bucket would split without end
}
inline const uint32_t SPLIT_AVG = split_avg_knob(); // (C++17 inline variables: one object for every translation unit, no guard in the hot path)
inline const bool GROW2 = getenv("KNGT_GROW2") && atoi(getenv("KNGT_GROW2")); // runs grow by doubling instead of by size class (measurement knob)
inline uint32_t split_avg() { return SPLIT_AVG; }
inline bool grow2() { return GROW2; }
constexpr uint8_t K_MAX = 24;

inline int cmp_x(const uint64_t a[2], const uint64_t b[2]) {
    if (a[1] != b[1]) return a[1] > b[1] ? 1 : -1;
    if (a[0] != b[0]) return a[0] > b[0] ? 1 : -1;
    return 0;
}
inline size_t fine_index(uint8_t k, uint64_t x1) { return k ? (size_t)(x1 >> (64 - k)) : 0; }

inline uint32_t cap_of_class(int c) { return (uint32_t)(class_bytes(c) / sizeof(kngt_entry)); }

inline bool reserve(Arena &a, Fine &f, uint32_t want) {
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

inline void free_bucket(Arena &a, Bucket &b) {
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
inline bool build(Arena &a, Bucket &b, uint8_t k, const kngt_entry *sorted, uint32_t n) {
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
inline void gather(const Bucket &b, kngt_entry *out) {
    if (!b.fine) return;
    const size_t nf = (size_t)1 << b.k;
    for (size_t i = 0; i < nf; i++) {
        if (b.fine[i].n) std::memcpy(out, b.fine[i].e, (size_t)b.fine[i].n * sizeof(kngt_entry));
        out += b.fine[i].n;
    }
}

inline bool resplit(Arena &a, Bucket &b, uint8_t k, std::vector<kngt_entry> &scratch) {
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

inline uint8_t k_for(uint32_t n) {
    uint8_t k = 0;
    while (k < K_MAX && n > ((split_avg() / 2) << k)) k += 2;
    return k;
}


// HashTable::Add(h, e) (HashTable.cpp:262-307) on this layout, without the maxItem bookkeeping (the callers keep that word,
// it belongs to the file format): KNGT_ADD_OK / _DUPLICATE / _COLLISION (*other = the entry already stored), -1 = out of
// memory or the 32-bit nbItem word is full.
inline int add_entry(Arena &ar, Bucket &b, const kngt_entry *e, kngt_entry *other) {
    if (!b.fine && !build(ar, b, 0, nullptr, 0)) return -1;
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
    if (!reserve(ar, f, (grow2() && f.n + 1 > f.cap && f.cap >= 4) ? 2 * f.cap : f.n + 1)) return -1;
    std::memmove(f.e + lo + 1, f.e + lo, (size_t)(f.n - lo) * sizeof(kngt_entry));
    f.e[lo] = *e;
    f.n++;
    b.n++;
    if (b.k < K_MAX && b.n > (split_avg() << b.k) && b.n >= b.split_retry) {
        thread_local std::vector<kngt_entry> scratch;
        // a failed re-split keeps the old (correct, slower) layout; gathering the whole bucket again on every insert
        // would make each one cost O(n): try again when the bucket has doubled
        if (resplit(ar, b, (uint8_t)(b.k + 2), scratch)) b.split_retry = 0;
        else b.split_retry = b.n > 0x7FFFFFFFu ? 0xFFFFFFFFu : 2 * b.n;
    }
    return KNGT_ADD_OK;
}

// hints for batched adds: stage 0 touches the bucket header, 1 the header of the run the key falls into, 2 the run itself (the
// search reads a few of its lines, the insertion shifts the rest).  They read words another thread may be changing: a stale
// word only makes a hint useless -- run headers and runs stay mapped until the table is reset.
__attribute__((no_sanitize("thread"))) inline void prefetch(const Bucket &b, uint64_t x1, int stage) {
    if (stage == 0) {
        __builtin_prefetch(&b);
        return;
    }
    const Fine *fine = __atomic_load_n(&b.fine, __ATOMIC_RELAXED);
    const uint8_t k = __atomic_load_n(&b.k, __ATOMIC_RELAXED);
    if (!fine) return;
    const Fine *f = &fine[fine_index(k, x1)];
    if (stage == 1) {
        __builtin_prefetch(f);
        return;
    }
    kngt_entry *e = __atomic_load_n(&f->e, __ATOMIC_RELAXED);
    const uint32_t n = __atomic_load_n(&f->n, __ATOMIC_RELAXED);
    if (e) {
        static const int lines = getenv("KNGT_PF_LINES") ? atoi(getenv("KNGT_PF_LINES")) : 12; // (measurement knob)
        const char *p = reinterpret_cast<const char *>(e), *end = p + (size_t)((n < 4096 ? n : 4096) + 1) * sizeof(kngt_entry);
        for (int i = 0; i < lines && p < end; i++, p += 64) __builtin_prefetch(p, 1);
    }
}

} // namespace kng_bucket
#endif
