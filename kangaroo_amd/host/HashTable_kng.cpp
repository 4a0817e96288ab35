// HashTable_kng.cpp -- `class HashTable` of the reference (HashTable.h:66-108) re-implemented for LINK-TIME replacement:
// this file is compiled against the reference's own HashTable.h (-I<reference root>) and its object takes the place of
// HashTable.o in the reference's link line.  No reference source is edited; Kangaroo.cpp, Backup.cpp, Check.cpp, Merge.cpp,
// PartMerge.cpp, Network.cpp and Thread.cpp keep calling the class exactly as before.
//
// What stays as in the reference (HashTable.cpp), because callers or files depend on it:
//   * every public member and its result: Add -> ADD_OK / ADD_DUPLICATE / ADD_COLLISION with kDist / kType decoded from the
//     entry ALREADY stored (HashTable.cpp:262-307), Convert (:85-113), CalcDistAndType (:249-260), MergeH (:119-221);
//   * E[h].nbItem, E[h].maxItem (the reference's 16, +4, +4 ... bookkeeping word for word -- it is written to work files,
//     :374-375) and E[h].items as an array of nbItem pointers to 32-byte ENTRYs (Check.cpp:47,88 reads it after LoadTable);
//   * SaveTable / LoadTable / SeekNbItem bytes (:369-458), the strings of GetSizeInfo and PrintInfo.
// What is different, because it is where the reference program's time went once a MI355X feeds it (at the program's own DP 14
// on an 80-bit range one GPU delivers 1.5 M points/s; the reference table took 0.65 us per point when small and 1.4 us after
// three minutes: malloc per entry, a realloc + memcpy of the pointer array every fourth insertion, a binary search whose
// every probe dereferences a pointer into a cold heap line, a memmove of half the bucket -- 19 then 12 GK/s of a 25 GK/s
// kernel, profiles/r03_reference_program_on_engine_80bit.txt):
//   * entries come from bump areas of 64 arenas (kng_arena.h; regions up to 256 MiB, huge pages where the system grants
//     them): no malloc, no per-entry header, neighbours in time are neighbours in memory;
//   * a bucket is ONE block {header, ENTRY *items[cap], uint64_t keys[cap]} that doubles when full; the block it leaves is
//     not returned to a free list -- uniform points make every bucket outgrow a size at about the same time, so nobody would
//     ever ask for that size again -- but cut into ENTRYs: the 16 B per entry of capacity a doubling frees is half of what the
//     entries of the next doubling need, so all of it is used again (50-53 B of memory per point against the reference's
//     56-60 with malloc's chunk headers).  keys[i]
//     mirrors items[i]->x.i64[1], the most significant word of the sort key, so a search reads a few adjacent words instead
//     of chasing pointers, and starts at the interpolated position (x is uniform): two or three cache lines per lookup
//     whatever the bucket size;
//   * items[0 .. sorted) is ascending, items[sorted .. nbItem) is a second ascending run of at most `tail` (64) entries that
//     new points go into; when it is full it is folded into the main run.  An insertion moves ~16 B x (n / 64 + 32) instead of
//     8 B x n / 2: 4096-entry buckets (eight GPUs, DP 11) cost what 64-entry buckets cost.  Nothing outside this file can
//     observe the difference: the only readers of items[] (Check.cpp:47,88 after LoadTable; SaveTable) get a folded bucket.
//     kng_ht_set_tail(ht, 0) keeps every bucket folded after every Add;
//   * 1024 stripe locks make Add and kng_ht_ingest (kng_hashtable_ext.h) safe from several threads at once; the reference
//     relies on the program's ghMutex (Kangaroo.cpp:594), which its own callers still take;
//   * a failed allocation ends the process with a message (the reference dereferences malloc's NULL).
#include "HashTable.h"

#include <inttypes.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <new>
#include <vector>

#include "kng_arena.h"
#include "kng_hashtable_ext.h"

using namespace kng_arena;

namespace {

constexpr unsigned STRIPE_BITS = 10, N_STRIPES = 1u << STRIPE_BITS;
constexpr unsigned STRIPE_SHIFT = HASH_SIZE_BIT - STRIPE_BITS; // a stripe = 256 consecutive buckets
constexpr uint64_t D_MASK = 0x3FFFFFFFFFFFFFFFULL, D_SIGN = 1ULL << 63, D_TYPE = 1ULL << 62;
constexpr uint32_t TAIL_DEFAULT = 64, TAIL_MAX = 4096;
constexpr unsigned ARENA_BITS = 6, N_ARENAS = 1u << ARENA_BITS; // 16 stripes share an arena (and its lock, for allocations only)
constexpr int FIRST_CLASS = 2;                       // 128 B: 7 entries; then 256, 512, ... (every second class: doubling)
constexpr size_t ENTRY_SLAB = (size_t)16 << 10;      // fresh entries are carved from 16 KiB blocks of the stripe's arena

// block = Hdr | ENTRY *items[cap] | uint64_t keys[cap]; E[h].items points at items
struct Hdr {
    uint32_t cap;    // entries the block has room for
    uint32_t sorted; // items[0 .. sorted) = main run, items[sorted .. nbItem) = tail run (both ascending)
    int32_t cls;     // arena size class of the block
    uint32_t pad;
};
static_assert(sizeof(Hdr) == 16, "header must keep items[] 16-byte aligned");

inline Hdr *hdr_of(ENTRY **items) { return reinterpret_cast<Hdr *>(items) - 1; }
inline uint64_t *keys_of(ENTRY **items, uint32_t cap) { return reinterpret_cast<uint64_t *>(items + cap); }
inline uint32_t cap_of_class(int c) { return (uint32_t)((class_bytes(c) - sizeof(Hdr)) / 16); }

struct Spare { // a bucket block that was outgrown, waiting to be cut into ENTRYs
    Spare *next;
    size_t bytes;
};

struct alignas(64) Stripe {
    std::atomic_flag lock = ATOMIC_FLAG_INIT;
    char *ecur = nullptr, *eend = nullptr; // bump area for ENTRYs
    Spare *spare = nullptr;
    Arena *arena = nullptr;
    uint64_t merges = 0, grows = 0, recycled = 0; // (written under the stripe lock)
    std::atomic<uint64_t> spins{0};                // failed attempts on the lock: counted by threads that do NOT hold it
};

struct Impl {
    HashTable *owner = nullptr;
    uint32_t tail = TAIL_DEFAULT;
    Stripe stripe[N_STRIPES];
    Arena arena[N_ARENAS];
    Impl() {
        for (unsigned i = 0; i < N_STRIPES; i++) stripe[i].arena = &arena[i >> (STRIPE_BITS - ARENA_BITS)];
    }
};

// ---- which Impl belongs to which HashTable object: the class has no spare member (HashTable.h:87-90) -----------------
constexpr int MAX_TABLES = 16;
std::atomic<HashTable *> g_owner[MAX_TABLES];
Impl *g_impl[MAX_TABLES];
std::mutex g_registry;

Impl *find_impl(HashTable *ht) {
    for (int i = 0; i < MAX_TABLES; i++)
        if (g_owner[i].load(std::memory_order_acquire) == ht) return g_impl[i];
    return nullptr;
}

[[noreturn]] void die(const char *what) {
    fprintf(stderr, "\nHashTable: %s\n", what);
    abort();
}

Impl *impl_of(HashTable *ht) {
    if (Impl *p = find_impl(ht)) return p;
    std::lock_guard<std::mutex> g(g_registry);
    if (Impl *p = find_impl(ht)) return p;
    for (int i = 0; i < MAX_TABLES; i++)
        if (g_owner[i].load(std::memory_order_relaxed) == nullptr) {
            Impl *p = new (std::nothrow) Impl();
            if (!p) die("out of memory");
            p->owner = ht;
            if (const char *e = getenv("KNG_HT_TAIL")) {
                const long v = atol(e);
                p->tail = (uint32_t)(v < 0 ? 0 : v > (long)TAIL_MAX ? TAIL_MAX : v);
            }
            g_impl[i] = p;
            g_owner[i].store(ht, std::memory_order_release);
            return p;
        }
    die("more than 16 HashTable objects alive");
}

void release_memory(Impl *p) {
    for (Stripe &s : p->stripe) {
        s.ecur = s.eend = nullptr;
        s.spare = nullptr;
    }
    for (Arena &a : p->arena) arena_release(a);
}

struct StripeLock {
    Stripe &s;
    explicit StripeLock(Stripe &st) : s(st) {
        while (s.lock.test_and_set(std::memory_order_acquire)) s.spins.fetch_add(1, std::memory_order_relaxed);
    }
    ~StripeLock() { s.lock.clear(std::memory_order_release); }
};

inline Stripe &stripe_of(Impl *p, uint64_t h) { return p->stripe[(h & HASH_MASK) >> STRIPE_SHIFT]; }

// ---- storage ---------------------------------------------------------------------------------------------------------
// `bytes` (a multiple of 32, at most half a slab) of the stripe's bump area; outgrown bucket blocks are used up first
char *slab_take(Stripe &s, size_t bytes) {
    while ((size_t)(s.eend - s.ecur) < bytes) { // what is left of the old area stays unused: less than one request
        if (Spare *sp = s.spare) {
            s.spare = sp->next;
            s.ecur = reinterpret_cast<char *>(sp);
            s.eend = s.ecur + sp->bytes;
            s.recycled += sp->bytes;
            continue;
        }
        char *b = static_cast<char *>(arena_alloc(*s.arena, class_of(ENTRY_SLAB)));
        if (!b) die("out of memory in the distinguished-point table");
        s.ecur = b;
        s.eend = b + ENTRY_SLAB;
    }
    char *r = s.ecur;
    s.ecur += bytes;
    return r;
}
inline ENTRY *new_entry(Stripe &s) { return reinterpret_cast<ENTRY *>(slab_take(s, sizeof(ENTRY))); }

// make room for at least `want` entries, keeping what is there
void reserve(Stripe &s, HASH_ENTRY &b, uint32_t want) {
    uint32_t cap = b.items ? hdr_of(b.items)->cap : 0;
    if (want <= cap) return;
    int c = b.items ? hdr_of(b.items)->cls + 2 : FIRST_CLASS;
    while (cap_of_class(c) < want) c += 2;
    char *blk = static_cast<char *>(arena_alloc(*s.arena, c));
    if (!blk) die("out of memory in the distinguished-point table");
    Hdr *nh = reinterpret_cast<Hdr *>(blk);
    nh->cap = cap_of_class(c);
    nh->cls = c;
    nh->pad = 0;
    ENTRY **nitems = reinterpret_cast<ENTRY **>(nh + 1);
    if (b.items) {
        Hdr *oh = hdr_of(b.items);
        nh->sorted = oh->sorted;
        memcpy(nitems, b.items, (size_t)b.nbItem * sizeof(ENTRY *));
        memcpy(keys_of(nitems, nh->cap), keys_of(b.items, oh->cap), (size_t)b.nbItem * sizeof(uint64_t));
        const size_t old_bytes = class_bytes(oh->cls);
        Spare *sp = reinterpret_cast<Spare *>(oh); // the old block: future ENTRYs (see the file header)
        sp->bytes = old_bytes;
        sp->next = s.spare;
        s.spare = sp;
        s.grows++;
    } else {
        nh->sorted = 0;
    }
    b.items = nitems;
}

// ---- search ----------------------------------------------------------------------------------------------------------
// first index in [lo, hi) whose key is >= key; starts at the interpolated position and gallops, so uniform keys cost O(1)
// probes next to each other and any other distribution O(log distance)
inline uint32_t lower_bound_key(const uint64_t *k, uint32_t lo, uint32_t hi, uint64_t key) {
    const uint32_t len = hi - lo;
    if (!len) return lo;
    const uint32_t g = lo + (uint32_t)(((unsigned __int128)key * len) >> 64);
    uint32_t a, b;
    if (k[g] < key) {
        a = g + 1;
        b = hi;
        for (uint32_t step = 1, p = g;; step <<= 1) {
            const uint64_t q = (uint64_t)p + step;
            if (q >= hi) break;
            if (k[q] >= key) {
                b = (uint32_t)q;
                break;
            }
            a = (uint32_t)q + 1;
            p = (uint32_t)q;
        }
    } else {
        a = lo;
        b = g;
        for (uint32_t step = 1, p = g;; step <<= 1) {
            if (p - lo < step) break;
            const uint32_t q = p - step;
            if (k[q] < key) {
                a = q + 1;
                break;
            }
            b = q;
            p = q;
        }
    }
    while (a < b) {
        const uint32_t m = a + (b - a) / 2;
        if (k[m] < key) a = m + 1;
        else b = m;
    }
    return a;
}

// position of x = (x1, x0) in the ascending run [lo, hi): *found = its index when present, else the return value is where
// it would be inserted
inline uint32_t locate(ENTRY **items, const uint64_t *keys, uint32_t lo, uint32_t hi, uint64_t x1, uint64_t x0, bool *found) {
    uint32_t i = lower_bound_key(keys, lo, hi, x1);
    *found = false;
    while (i < hi && keys[i] == x1) { // the 64-bit key ties: order by the low word, which only the entry has
        const uint64_t lowi = items[i]->x.i64[0];
        if (lowi == x0) {
            *found = true;
            return i;
        }
        if (lowi > x0) break;
        i++;
    }
    return i;
}

inline bool entry_greater(ENTRY *a, uint64_t ka, ENTRY *b, uint64_t kb) {
    if (ka != kb) return ka > kb;
    return a->x.i64[0] > b->x.i64[0];
}

// fold the tail run into the main run, in place, from the top
void fold(Stripe &s, HASH_ENTRY &b) {
    Hdr *h = hdr_of(b.items);
    const uint32_t n = b.nbItem, m = h->sorted;
    if (m >= n) return;
    if (m == 0) { // the tail is the whole bucket and is ascending
        h->sorted = n;
        return;
    }
    const uint32_t t = n - m;
    ENTRY *tp_small[128];
    uint64_t tk_small[128];
    std::vector<ENTRY *> tp_big;
    std::vector<uint64_t> tk_big;
    ENTRY **tp = tp_small;
    uint64_t *tk = tk_small;
    if (t > 128) {
        tp_big.resize(t);
        tk_big.resize(t);
        tp = tp_big.data();
        tk = tk_big.data();
    }
    uint64_t *keys = keys_of(b.items, h->cap);
    memcpy(tp, b.items + m, (size_t)t * sizeof(ENTRY *));
    memcpy(tk, keys + m, (size_t)t * sizeof(uint64_t));
    int64_t i = (int64_t)m - 1, j = (int64_t)t - 1, w = (int64_t)n - 1;
    while (j >= 0) {
        if (i >= 0 && entry_greater(b.items[i], keys[i], tp[j], tk[j])) {
            b.items[w] = b.items[i];
            keys[w] = keys[i];
            i--;
        } else {
            b.items[w] = tp[j];
            keys[w] = tk[j];
            j--;
        }
        w--;
    }
    h->sorted = n;
    s.merges++;
}

// The reference's Add(h, ENTRY*) (HashTable.cpp:262-307) on our layout.  `ext` = an entry the caller allocated (the public
// Add(h, ENTRY*)); otherwise the entry is created here, and only when the point is new.  On ADD_COLLISION *stored = the d word
// of the entry found.
int insert(Impl *p, HashTable *ht, uint64_t h, uint64_t x0, uint64_t x1, uint64_t d0, uint64_t d1, ENTRY *ext, int128_t *stored) {
    h &= HASH_MASK;
    Stripe &s = stripe_of(p, h);
    StripeLock g(s);
    HASH_ENTRY &b = ht->E[h];
    if (b.maxItem == 0) b.maxItem = 16;
    uint32_t n = b.nbItem;
    if (n == 0) {
        reserve(s, b, 1);
        hdr_of(b.items)->sorted = 0;
    } else {
        if (n >= b.maxItem - 1) b.maxItem += 4; // ReAllocate(h, 4): the word that goes to the work file
        if (!b.items) die("Add on a table that holds counts only (SeekNbItem)");
    }
    Hdr *hd = hdr_of(b.items);
    uint64_t *keys = keys_of(b.items, hd->cap);
    bool found = false;
    uint32_t at = 0;
    if (hd->sorted) at = locate(b.items, keys, 0, hd->sorted, x1, x0, &found);
    uint32_t pos = at;
    if (!found) pos = locate(b.items, keys, hd->sorted, n, x1, x0, &found); // the tail run: where new points go
    if (found) {
        const ENTRY *old = b.items[pos];
        if (old->d.i64[0] == d0 && old->d.i64[1] == d1) return ADD_DUPLICATE; // same point twice, or same herd
        if (stored) *stored = old->d;
        return ADD_COLLISION;
    }
    if (n == hd->cap) {
        reserve(s, b, n + 1);
        hd = hdr_of(b.items);
        keys = keys_of(b.items, hd->cap);
    }
    ENTRY *e = ext;
    if (!e) {
        e = new_entry(s);
        e->x.i64[0] = x0;
        e->x.i64[1] = x1;
        e->d.i64[0] = d0;
        e->d.i64[1] = d1;
    }
    if (pos < n) {
        memmove(b.items + pos + 1, b.items + pos, (size_t)(n - pos) * sizeof(ENTRY *));
        memmove(keys + pos + 1, keys + pos, (size_t)(n - pos) * sizeof(uint64_t));
    }
    b.items[pos] = e;
    keys[pos] = x1;
    b.nbItem = ++n;
    if (n - hd->sorted > p->tail || hd->sorted == 0) fold(s, b);
    return ADD_OK;
}

// The look-ahead of kng_ht_ingest: prefetch hints computed from words read WITHOUT the stripe lock.  Racy by design -- a stale
// or half-updated word only makes a hint useless (the addresses stay inside mappings that live until Reset) -- so the function
// is excluded from ThreadSanitizer instead of pretending the reads are ordered; everything that decides anything happens under
// the lock in insert().
__attribute__((no_sanitize("thread"))) inline void hint_stages(HashTable *ht, const kng_dp_record *recs, uint32_t n, uint32_t i, uint32_t A,
                                                               uint32_t B, uint32_t C) {
    if (i < n) __builtin_prefetch(&ht->E[recs[i].x[2] & HASH_MASK], 0, 1);
    if (i >= A - B && i - (A - B) < n) {
        const kng_dp_record &r = recs[i - (A - B)];
        ENTRY **items = __atomic_load_n(&ht->E[r.x[2] & HASH_MASK].items, __ATOMIC_RELAXED);
        if (items) __builtin_prefetch(hdr_of(items), 0, 1);
    }
    if (i >= A - C && i - (A - C) < n) {
        const kng_dp_record &r = recs[i - (A - C)];
        const HASH_ENTRY &b = ht->E[r.x[2] & HASH_MASK];
        ENTRY **items = __atomic_load_n(&b.items, __ATOMIC_RELAXED);
        if (items) {
            const Hdr *hd = hdr_of(items);
            const uint32_t cap = __atomic_load_n(&hd->cap, __ATOMIC_RELAXED), m = __atomic_load_n(&hd->sorted, __ATOMIC_RELAXED);
            const uint32_t nb = __atomic_load_n(&b.nbItem, __ATOMIC_RELAXED);
            if (m <= nb && nb <= cap && nb - m <= TAIL_MAX + 8) { // stale words can say anything: bound the hint loop
                const uint64_t *keys = keys_of(items, cap);
                if (m) __builtin_prefetch(keys + (uint32_t)(((unsigned __int128)r.x[1] * m) >> 64), 0, 1);
                for (uint32_t q = m; q <= nb; q += 8) { // the tail run: searched, then shifted by the insertion
                    __builtin_prefetch(keys + q, 1, 1);
                    __builtin_prefetch(items + q, 1, 1);
                }
            }
        }
    }
}

} // namespace

// ---- the class -------------------------------------------------------------------------------------------------------

HashTable::HashTable() {
    memset(E, 0, sizeof(E));
    kType = 0;
    // an object constructed where a destroyed one was (the class has no destructor to tell us): start from nothing
    if (Impl *p = find_impl(this)) release_memory(p);
}

void HashTable::Reset() {
    if (Impl *p = find_impl(this)) release_memory(p);
    memset(E, 0, sizeof(E));
}

uint64_t HashTable::GetNbItem() {
    uint64_t total = 0;
    for (uint32_t h = 0; h < HASH_SIZE; h++) total += E[h].nbItem;
    return total;
}

ENTRY *HashTable::CreateEntry(int128_t *x, int128_t *d) {
    // only reachable through this file; entries live in the arena of the bucket they go to (see insert)
    ENTRY *e = static_cast<ENTRY *>(malloc(sizeof(ENTRY)));
    if (!e) die("out of memory");
    e->x = *x;
    e->d = *d;
    return e;
}

void HashTable::Convert(Int *x, Int *d, uint32_t type, uint64_t *h, int128_t *X, int128_t *D) {
    X->i64[0] = x->bits64[0];
    X->i64[1] = x->bits64[1];
    uint64_t flags = (uint64_t)type << 62;
    if (d->bits64[3] > 0x7FFFFFFFFFFFFFFFULL) { // negative mod n: store |d| and the sign (HashTable.cpp:96-102)
        Int neg(d);
        neg.ModNegK1order();
        D->i64[0] = neg.bits64[0];
        D->i64[1] = neg.bits64[1] & D_MASK;
        flags |= D_SIGN;
    } else {
        D->i64[0] = d->bits64[0];
        D->i64[1] = d->bits64[1] & D_MASK;
    }
    D->i64[1] |= flags;
    *h = x->bits64[2] & HASH_MASK;
}

void HashTable::CalcDistAndType(int128_t d, Int *kDist, uint32_t *kType) {
    *kType = (d.i64[1] & D_TYPE) != 0;
    const bool negative = (d.i64[1] & D_SIGN) != 0;
    kDist->SetInt32(0);
    kDist->bits64[0] = d.i64[0];
    kDist->bits64[1] = d.i64[1] & D_MASK;
    if (negative) kDist->ModNegK1order();
}

int HashTable::Add(Int *x, Int *d, uint32_t type) {
    int128_t X, D, stored;
    uint64_t h;
    Convert(x, d, type, &h, &X, &D);
    const int st = insert(impl_of(this), this, h, X.i64[0], X.i64[1], D.i64[0], D.i64[1], nullptr, &stored);
    if (st == ADD_COLLISION) CalcDistAndType(stored, &kDist, &kType);
    return st;
}

int HashTable::Add(uint64_t h, int128_t *x, int128_t *d) {
    int128_t stored;
    const int st = insert(impl_of(this), this, h, x->i64[0], x->i64[1], d->i64[0], d->i64[1], nullptr, &stored);
    if (st == ADD_COLLISION) CalcDistAndType(stored, &kDist, &kType);
    return st;
}

int HashTable::Add(uint64_t h, ENTRY *e) {
    int128_t stored;
    const int st = insert(impl_of(this), this, h, e->x.i64[0], e->x.i64[1], e->d.i64[0], e->d.i64[1], e, &stored);
    if (st == ADD_COLLISION) CalcDistAndType(stored, &kDist, &kType);
    return st;
}

void HashTable::ReAllocate(uint64_t h, uint32_t add) {
    Impl *p = impl_of(this);
    h &= HASH_MASK;
    Stripe &s = stripe_of(p, h);
    StripeLock g(s);
    E[h].maxItem += add;
    reserve(s, E[h], E[h].maxItem);
}

int HashTable::compare(int128_t *i1, int128_t *i2) {
    if (i1->i64[1] != i2->i64[1]) return i1->i64[1] > i2->i64[1] ? 1 : -1;
    if (i1->i64[0] != i2->i64[0]) return i1->i64[0] > i2->i64[0] ? 1 : -1;
    return 0;
}

std::string HashTable::GetStr(int128_t *i) {
    char tmp[40];
    snprintf(tmp, sizeof tmp, "%08X%08X%08X%08X", i->i32[3], i->i32[2], i->i32[1], i->i32[0]);
    return std::string(tmp);
}

// "used/total" like HashTable.cpp:325-357: used = what a file of the table takes, total = what the process holds for it
// (here: the bucket array plus every page the arenas have touched)
std::string HashTable::GetSizeInfo() {
    uint64_t used = (uint64_t)HASH_SIZE * 2 * sizeof(uint32_t) + GetNbItem() * sizeof(ENTRY);
    uint64_t total = sizeof(E);
    if (Impl *p = find_impl(this))
        for (const Arena &a : p->arena) total += arena_touched(a);
    else
        for (uint32_t h = 0; h < HASH_SIZE; h++) total += (uint64_t)E[h].nbItem * sizeof(ENTRY); // counts only (SeekNbItem)
    const char *unit = "MB";
    double totalMB = (double)total / (1024.0 * 1024.0), usedMB = (double)used / (1024.0 * 1024.0);
    if (totalMB > 1024) {
        totalMB /= 1024;
        usedMB /= 1024;
        unit = "GB";
    }
    if (totalMB > 1024) {
        totalMB /= 1024;
        usedMB /= 1024;
        unit = "TB";
    }
    char ret[256];
    snprintf(ret, sizeof ret, "%.1f/%.1f%s", usedMB, totalMB, unit);
    return std::string(ret);
}

void HashTable::PrintInfo() {
    uint16_t max = 0, min = 65535;
    uint32_t maxH = 0, minH = 0;
    const uint64_t count = GetNbItem();
    const double avg = (double)count / (double)HASH_SIZE;
    double var = 0;
    for (uint32_t h = 0; h < HASH_SIZE; h++) {
        if (E[h].nbItem > max) {
            max = (uint16_t)E[h].nbItem;
            maxH = h;
        }
        if (E[h].nbItem < min) {
            min = (uint16_t)E[h].nbItem;
            minH = h;
        }
        var += (avg - (double)E[h].nbItem) * (avg - (double)E[h].nbItem);
    }
    const double sdev = sqrt(var / (double)HASH_SIZE);
    ::printf("DP Size   : %s\n", GetSizeInfo().c_str());
    ::printf("DP Count  : %" PRId64 " 2^%.3f\n", (int64_t)count, log2((double)count));
    ::printf("HT Max    : %d [@ %06X]\n", max, maxH);
    ::printf("HT Min    : %d [@ %06X]\n", min, minH);
    ::printf("HT Avg    : %.2f \n", avg);
    ::printf("HT SDev   : %.2f \n", sdev);
}

void HashTable::SaveTable(FILE *f) { SaveTable(f, 0, HASH_SIZE, true); }

void HashTable::SaveTable(FILE *f, uint32_t from, uint32_t to, bool printPoint) {
    Impl *p = find_impl(this);
    const uint64_t point = GetNbItem() / 16; // a dot every point + 1 entries (HashTable.cpp:367-385)
    uint64_t pending = 0;
    std::vector<ENTRY> row;
    for (uint32_t h = from; h < to; h++) {
        HASH_ENTRY &b = E[h];
        fwrite(&b.nbItem, sizeof(uint32_t), 1, f);
        fwrite(&b.maxItem, sizeof(uint32_t), 1, f);
        if (!b.nbItem || !b.items) continue;
        if (p) {
            Stripe &s = stripe_of(p, h);
            StripeLock g(s);
            fold(s, b);
        }
        row.resize(b.nbItem);
        for (uint32_t i = 0; i < b.nbItem; i++) row[i] = *b.items[i];
        fwrite(row.data(), sizeof(ENTRY), b.nbItem, f);
        if (printPoint) {
            pending += b.nbItem;
            while (pending > point) {
                ::printf(".");
                pending -= point + 1;
            }
        }
    }
}

void HashTable::SeekNbItem(FILE *f, bool restorePos) {
    Reset();
    const off_t org = ftello(f);
    SeekNbItem(f, 0, HASH_SIZE);
    if (restorePos) fseeko(f, org, SEEK_SET);
}

// counts only: nbItem / maxItem of every bucket, entries skipped, items left NULL (HashTable.cpp:410-427; `-winfo`)
void HashTable::SeekNbItem(FILE *f, uint32_t from, uint32_t to) {
    for (uint32_t h = from; h < to; h++) {
        if (fread(&E[h].nbItem, sizeof(uint32_t), 1, f) != 1 || fread(&E[h].maxItem, sizeof(uint32_t), 1, f) != 1) return;
        fseeko(f, (off_t)32 * E[h].nbItem, SEEK_CUR);
    }
}

void HashTable::LoadTable(FILE *f) { LoadTable(f, 0, HASH_SIZE); }

void HashTable::LoadTable(FILE *f, uint32_t from, uint32_t to) {
    Reset();
    Impl *p = impl_of(this);
    for (uint32_t h = from; h < to; h++) {
        HASH_ENTRY &b = E[h];
        uint32_t nb = 0, mx = 0;
        if (fread(&nb, sizeof(uint32_t), 1, f) != 1 || fread(&mx, sizeof(uint32_t), 1, f) != 1) return;
        b.maxItem = mx;
        if (!nb) continue;
        Stripe &s = stripe_of(p, h);
        StripeLock g(s);
        reserve(s, b, nb > mx ? nb : mx);
        // a bucket's entries are read with one call into one block of the stripe's arena
        const size_t bytes = (size_t)nb * sizeof(ENTRY);
        ENTRY *row;
        if (bytes <= ENTRY_SLAB / 2) {
            row = reinterpret_cast<ENTRY *>(slab_take(s, bytes));
        } else {
            row = static_cast<ENTRY *>(arena_alloc(*s.arena, class_of(bytes)));
            if (!row) die("out of memory in the distinguished-point table");
        }
        const size_t got = fread(row, sizeof(ENTRY), nb, f);
        Hdr *hd = hdr_of(b.items);
        uint64_t *keys = keys_of(b.items, hd->cap);
        bool ascending = true;
        for (uint32_t i = 0; i < (uint32_t)got; i++) {
            b.items[i] = row + i;
            keys[i] = row[i].x.i64[1];
            if (i && !entry_greater(row + i, keys[i], row + i - 1, keys[i - 1])) ascending = false;
        }
        b.nbItem = (uint32_t)got;
        hd->sorted = b.nbItem;
        if (!ascending) { // not a file the reference wrote; keep what it holds, in the order the searches need
            for (uint32_t i = 1; i < b.nbItem; i++) {
                ENTRY *e = b.items[i];
                const uint64_t k = keys[i];
                uint32_t j = i;
                while (j && entry_greater(b.items[j - 1], keys[j - 1], e, k)) {
                    b.items[j] = b.items[j - 1];
                    keys[j] = keys[j - 1];
                    j--;
                }
                b.items[j] = e;
                keys[j] = k;
            }
        }
        if (got != nb) return;
    }
}

// Merge bucket h of two work files into fd (HashTable.cpp:119-221): both rows ascending, equal x keeps the entry of f1,
// counting a duplicate when the distances agree and reporting a collision (the last one of the row) when they do not; the
// output row carries maxItem = entries rounded up to a multiple of four.
int HashTable::MergeH(uint32_t h, FILE *f1, FILE *f2, FILE *fd, uint32_t *nbDP, uint32_t *duplicate, Int *d1, uint32_t *k1, Int *d2,
                      uint32_t *k2) {
    (void)h;
    uint32_t head1[2] = {0, 0}, head2[2] = {0, 0};
    *duplicate = 0;
    *nbDP = 0;
    if (fread(head1, sizeof(uint32_t), 2, f1) != 2) head1[0] = 0;
    if (fread(head2, sizeof(uint32_t), 2, f2) != 2) head2[0] = 0;
    const uint32_t n1 = head1[0], n2 = head2[0];
    if (n1 + n2 == 0) {
        const uint32_t zero[2] = {0, 0};
        fwrite(zero, sizeof(uint32_t), 2, fd);
        return ADD_OK;
    }
    std::vector<ENTRY> a(n1), b(n2), out;
    if (n1 && fread(a.data(), sizeof(ENTRY), n1, f1) != n1) die("MergeH: short read");
    if (n2 && fread(b.data(), sizeof(ENTRY), n2, f2) != n2) die("MergeH: short read");
    out.reserve((size_t)n1 + n2);
    bool collision = false;
    uint32_t i = 0, j = 0;
    while (i < n1 && j < n2) {
        const int c = compare(&a[i].x, &b[j].x);
        if (c < 0) {
            out.push_back(a[i++]);
        } else if (c > 0) {
            out.push_back(b[j++]);
        } else {
            if (a[i].d.i64[0] == b[j].d.i64[0] && a[i].d.i64[1] == b[j].d.i64[1]) {
                *duplicate += 1;
            } else {
                CalcDistAndType(a[i].d, d1, k1);
                CalcDistAndType(b[j].d, d2, k2);
                collision = true;
            }
            out.push_back(a[i]);
            i++;
            j++;
        }
    }
    while (i < n1) out.push_back(a[i++]);
    while (j < n2) out.push_back(b[j++]);
    const uint32_t nbd = (uint32_t)out.size();
    const uint32_t head[2] = {nbd, (nbd + 3) / 4 * 4};
    fwrite(head, sizeof(uint32_t), 2, fd);
    fwrite(out.data(), sizeof(ENTRY), nbd, fd);
    *nbDP = nbd;
    return collision ? ADD_COLLISION : ADD_OK;
}

// ---- the batch interface (kng_hashtable_ext.h) ----------------------------------------------------------------------

extern "C" {

int kng_ht_ingest(HashTable *ht, const kng_dp_record *recs, uint32_t n, const uint64_t wild_off[2], kng_ht_event *ev, uint32_t ev_cap,
                  uint32_t *n_ev) {
    Impl *p = impl_of(ht);
    uint32_t events = 0;
    const unsigned __int128 off = ((unsigned __int128)wild_off[1] << 64) | wild_off[0];
    // three look-ahead stages turn the dependent misses of one insertion (bucket word -> block header -> keys at the
    // interpolated position and the tail run) into independent misses of different insertions.  The reads ahead of the
    // lock are hints only: a block another thread is replacing stays mapped (arenas never unmap before Reset).
    constexpr uint32_t A = 24, B = 16, C = 8;
    for (uint32_t i = 0; i < n + A; i++) {
        hint_stages(ht, recs, n, i, A, B, C);
        if (i < A) continue;
        const uint32_t at = i - A;
        const kng_dp_record &r = recs[at];
        // GPUEngine.cu:672 (true distance of a wild kangaroo = device distance - offset mod n) followed by
        // HashTable::Convert: with |true distance| < 2^127 the mod-n negation of the reference is this 128-bit subtraction
        unsigned __int128 d = ((unsigned __int128)r.d[1] << 64) | r.d[0];
        uint64_t flags = 0;
        if (r.kidx & 1) {
            flags = D_TYPE;
            if (d >= off) {
                d -= off;
            } else {
                d = off - d;
                flags |= D_SIGN;
            }
        }
        const uint64_t d0 = (uint64_t)d, d1 = ((uint64_t)(d >> 64) & D_MASK) | flags;
        int128_t stored;
        const int st = insert(p, ht, r.x[2] & HASH_MASK, r.x[0], r.x[1], d0, d1, nullptr, &stored);
        if (st != ADD_OK) {
            if (ev && events < ev_cap) {
                ev[events].index = at;
                ev[events].status = (uint32_t)st;
                ev[events].stored_d[0] = st == ADD_COLLISION ? stored.i64[0] : 0;
                ev[events].stored_d[1] = st == ADD_COLLISION ? stored.i64[1] : 0;
            }
            events++;
        }
    }
    if (n_ev) *n_ev = events;
    return 0;
}

void kng_ht_normalize(HashTable *ht) {
    Impl *p = find_impl(ht);
    if (!p) return;
    for (uint32_t h = 0; h < HASH_SIZE; h++) {
        if (!ht->E[h].items) continue;
        Stripe &s = stripe_of(p, h);
        StripeLock g(s);
        if (ht->E[h].items) fold(s, ht->E[h]);
    }
}

void kng_ht_set_tail(HashTable *ht, uint32_t tail) {
    Impl *p = impl_of(ht);
    p->tail = tail > TAIL_MAX ? TAIL_MAX : tail;
    if (tail == 0) kng_ht_normalize(ht);
}

void kng_ht_stats(HashTable *ht, kng_ht_stats_t *out) {
    memset(out, 0, sizeof *out);
    out->entries = ht->GetNbItem();
    if (Impl *p = find_impl(ht)) {
        for (const Arena &a : p->arena) {
            out->bytes_mapped += a.bytes;
            out->bytes_touched += arena_touched(a);
        }
        for (const Stripe &s : p->stripe) {
            out->merges += s.merges;
            out->grows += s.grows;
            out->lock_spins += s.spins.load(std::memory_order_relaxed);
            out->bytes_recycled += s.recycled;
        }
    }
}
}
