// HashTable_kng.cpp -- `class HashTable` of the reference (HashTable.h:66-108) re-implemented for LINK-TIME replacement:
// this file is compiled against the reference's own HashTable.h (-I<reference root>) and its object takes the place of
// HashTable.o in the reference's link line.  No reference source is edited; Kangaroo.cpp, Backup.cpp, Check.cpp, Merge.cpp,
// PartMerge.cpp, Network.cpp and Thread.cpp keep calling the class exactly as before.
//
// What stays as in the reference (HashTable.cpp), because callers or files depend on it:
//   * every public member and its result: Add -> ADD_OK / ADD_DUPLICATE / ADD_COLLISION with kDist / kType decoded from the
//     entry ALREADY stored (HashTable.cpp:262-307), Convert (:85-113), CalcDistAndType (:249-260), MergeH (:119-221);
//   * E[h].nbItem and E[h].maxItem at all times (the reference's 16, +4, +4 ... bookkeeping word for word -- it is written to
//     work files, :374-375 -- and the status line reads nbItem through GetNbItem);
//   * E[h].items as an array of nbItem pointers to 32-byte ENTRYs WHERE SOMEBODY READS IT: the only readers outside the class
//     are Check.cpp:47,88 (-wcheck), after LoadTable.  LoadTable builds that view; the first insertion into a loaded table
//     drops all of it (a restored search, -i, never looks at it);
//   * SaveTable / LoadTable / SeekNbItem bytes (:369-458), the strings of GetSizeInfo and PrintInfo.
// What is different, because it is where the reference program's time went once a MI355X feeds it (at the program's own DP 14
// on an 80-bit range one GPU delivers 1.5 M points/s, eight GPUs at their DP 11 98 M/s; the reference table took 0.65 us per
// point when small and 1.4 us after three minutes: malloc per entry, a realloc + memcpy of the pointer array every fourth
// insertion, a binary search whose every probe dereferences a pointer into a cold heap line, a memmove of half the bucket):
//   * round 6: the storage is kng_bucket.h, shared with the repo's own table (kng_dptable.cpp) -- the ENTRIES THEMSELVES in
//     short sorted runs, 2^k runs per bucket by the top bits of the sort key, k growing with the bucket: an insertion reads one
//     run header and moves at most a few hundred bytes whatever the table holds; no pointer array, no per-entry allocation,
//     45-47 B of memory per point.  (Round 5 kept the reference-visible ENTRY*[] array live -- pointer array + mirrored key
//     array + separately allocated entries, two memmoves per insertion: 280-380 ns per point and thread at 16 threads, three
//     times the repo's own table, profiles/r05_htbench_threads_gpu_host.txt.)
//   * memory comes from 64 arenas (kng_arena.h; regions up to 256 MiB, huge pages where the system grants them);
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
#include "kng_bucket.h"
#include "kng_hashtable_ext.h"

using namespace kng_arena;
using kng_bucket::Bucket;
using kng_bucket::Fine;

static_assert(sizeof(ENTRY) == sizeof(kngt_entry) && sizeof(ENTRY) == 32, "ENTRY is the 32-byte record of the file format");

namespace {

constexpr unsigned STRIPE_BITS = 10, N_STRIPES = 1u << STRIPE_BITS;
constexpr unsigned STRIPE_SHIFT = HASH_SIZE_BIT - STRIPE_BITS; // a stripe = 256 consecutive buckets
constexpr uint64_t D_MASK = 0x3FFFFFFFFFFFFFFFULL, D_SIGN = 1ULL << 63, D_TYPE = 1ULL << 62;
constexpr unsigned ARENA_BITS = 6, N_ARENAS = 1u << ARENA_BITS; // 16 stripes share an arena (and its lock, for allocations only)

struct alignas(64) Stripe {
    std::atomic_flag lock = ATOMIC_FLAG_INIT;
    Arena *arena = nullptr;
    uint64_t resplits = 0;          // (written under the stripe lock)
    std::atomic<uint64_t> spins{0}; // failed attempts on the lock: counted by threads that do NOT hold it
};

struct Impl {
    HashTable *owner = nullptr;
    Bucket bk[HASH_SIZE]; // bk[h].n == E[h].nbItem whenever the bucket has storage
    Stripe stripe[N_STRIPES];
    Arena arena[N_ARENAS];
    // the ENTRY*[] views LoadTable builds for Check.cpp: one arena of their own, dropped as a whole by the first insertion
    Arena view_arena;
    std::atomic<bool> views_live{false};
    std::mutex view_lock;
    Impl() {
        for (unsigned i = 0; i < N_STRIPES; i++) stripe[i].arena = &arena[i >> (STRIPE_BITS - ARENA_BITS)];
    }
};

// ---- which Impl belongs to which HashTable object: the class has no spare member (HashTable.h:87-90) and no destructor.
// A table of slots, claimed at the first insertion / load (not at construction) and given back by kng_ht_release; an object
// constructed at an address a slot remembers takes that slot over with its memory released.  VERDICT r5 weak 7: 16 slots made
// the 17th distinct table abort; 1024 now, behind a per-thread one-entry cache so that the per-point Add path does not scan.
constexpr int MAX_TABLES = 1024;
std::atomic<HashTable *> g_owner[MAX_TABLES];
Impl *g_impl[MAX_TABLES];
std::mutex g_registry;
std::atomic<uint64_t> g_registry_gen{1}; // bumped whenever a slot changes hands: invalidates the per-thread caches

Impl *find_impl(HashTable *ht) {
    thread_local HashTable *c_ht = nullptr;
    thread_local Impl *c_impl = nullptr;
    thread_local uint64_t c_gen = 0;
    const uint64_t gen = g_registry_gen.load(std::memory_order_acquire);
    if (c_ht == ht && c_gen == gen) return c_impl;
    for (int i = 0; i < MAX_TABLES; i++)
        if (g_owner[i].load(std::memory_order_acquire) == ht) {
            c_ht = ht;
            c_impl = g_impl[i];
            c_gen = gen;
            return c_impl;
        }
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
            Impl *p = g_impl[i] ? g_impl[i] : new (std::nothrow) Impl(); // (a released slot keeps its Impl, empty)
            if (!p) die("out of memory");
            p->owner = ht;
            g_impl[i] = p;
            g_owner[i].store(ht, std::memory_order_release);
            g_registry_gen.fetch_add(1, std::memory_order_acq_rel);
            return p;
        }
    die("more than 1024 HashTable objects alive (kng_ht_release gives a slot back)");
}

void release_memory(Impl *p) {
    for (Bucket &b : p->bk) b = Bucket();
    for (Arena &a : p->arena) arena_release(a);
    arena_release(p->view_arena);
    p->views_live.store(false, std::memory_order_relaxed);
}

struct StripeLock {
    Stripe &s;
    explicit StripeLock(Stripe &st) : s(st) {
        while (s.lock.test_and_set(std::memory_order_acquire)) s.spins.fetch_add(1, std::memory_order_relaxed);
    }
    ~StripeLock() { s.lock.clear(std::memory_order_release); }
};

inline Stripe &stripe_of(Impl *p, uint64_t h) { return p->stripe[(h & HASH_MASK) >> STRIPE_SHIFT]; }

// the views are all or nothing: whoever changes a loaded table first takes them all away (E[h].items = NULL everywhere),
// so that nobody can ever read a view that is out of date
void drop_views(Impl *p, HashTable *ht) {
    std::lock_guard<std::mutex> g(p->view_lock);
    if (!p->views_live.load(std::memory_order_acquire)) return;
    for (uint32_t h = 0; h < HASH_SIZE; h++) ht->E[h].items = nullptr;
    arena_release(p->view_arena);
    p->views_live.store(false, std::memory_order_release);
}

// items[i] -> the i-th entry of bucket h, where it lies (view_lock and the stripe lock taken)
void build_view(Impl *p, HashTable *ht, uint32_t h) {
    const Bucket &b = p->bk[h];
    if (!b.n || !b.fine) {
        ht->E[h].items = nullptr;
        return;
    }
    ENTRY **view = static_cast<ENTRY **>(arena_alloc(p->view_arena, class_of((size_t)b.n * sizeof(ENTRY *))));
    if (!view) die("out of memory in the distinguished-point table");
    size_t at = 0;
    const size_t nf = (size_t)1 << b.k;
    for (size_t q = 0; q < nf; q++)
        for (uint32_t i = 0; i < b.fine[q].n; i++) view[at++] = reinterpret_cast<ENTRY *>(b.fine[q].e + i);
    ht->E[h].items = view;
}

// The reference's Add(h, ENTRY*) (HashTable.cpp:262-307) on our layout.  On ADD_COLLISION *stored = the d word of the entry found.
int insert(Impl *p, HashTable *ht, uint64_t h, uint64_t x0, uint64_t x1, uint64_t d0, uint64_t d1, int128_t *stored) {
    h &= HASH_MASK;
    if (p->views_live.load(std::memory_order_acquire)) drop_views(p, ht);
    Stripe &s = stripe_of(p, h);
    StripeLock g(s);
    HASH_ENTRY &e = ht->E[h];
    Bucket &b = p->bk[h];
    if (e.nbItem != b.n) die("Add on a table that holds counts only (SeekNbItem)");
    if (e.maxItem == 0) e.maxItem = 16;
    if (b.n && b.n >= e.maxItem - 1) e.maxItem += 4; // ReAllocate(h, 4): the word that goes to the work file
    const kngt_entry ne = {{x0, x1}, {d0, d1}};
    kngt_entry other;
    const uint8_t k0 = b.k;
    const int st = kng_bucket::add_entry(*s.arena, b, &ne, &other);
    if (st < 0) die("out of memory in the distinguished-point table");
    if (b.k != k0) s.resplits++;
    if (st == KNGT_ADD_COLLISION && stored) {
        stored->i64[0] = other.d[0];
        stored->i64[1] = other.d[1];
    }
    // (the status line's GetNbItem reads this word without the lock, as it does in the reference: a relaxed atomic store)
    __atomic_store_n(&e.nbItem, b.n, __ATOMIC_RELAXED);
    return st; // KNGT_ADD_* == ADD_* (HashTable.h:33-35)
}
static_assert(KNGT_ADD_OK == ADD_OK && KNGT_ADD_DUPLICATE == ADD_DUPLICATE && KNGT_ADD_COLLISION == ADD_COLLISION, "status codes");

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
    for (uint32_t h = 0; h < HASH_SIZE; h++) total += __atomic_load_n(&E[h].nbItem, __ATOMIC_RELAXED); // (table threads may be inserting)
    return total;
}

ENTRY *HashTable::CreateEntry(int128_t *x, int128_t *d) {
    // (the reference's Add overloads call this; ours store the 32 bytes themselves.  Kept for callers outside the class.)
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
    const int st = insert(impl_of(this), this, h, X.i64[0], X.i64[1], D.i64[0], D.i64[1], &stored);
    if (st == ADD_COLLISION) CalcDistAndType(stored, &kDist, &kType);
    return st;
}

int HashTable::Add(uint64_t h, int128_t *x, int128_t *d) {
    int128_t stored;
    const int st = insert(impl_of(this), this, h, x->i64[0], x->i64[1], d->i64[0], d->i64[1], &stored);
    if (st == ADD_COLLISION) CalcDistAndType(stored, &kDist, &kType);
    return st;
}

// the reference keeps the caller's pointer in items[]; here the 32 bytes are copied into the bucket and the caller keeps
// its entry (nothing in the reference calls this overload from outside the class)
int HashTable::Add(uint64_t h, ENTRY *e) {
    int128_t stored;
    const int st = insert(impl_of(this), this, h, e->x.i64[0], e->x.i64[1], e->d.i64[0], e->d.i64[1], &stored);
    if (st == ADD_COLLISION) CalcDistAndType(stored, &kDist, &kType);
    return st;
}

void HashTable::ReAllocate(uint64_t h, uint32_t add) {
    Impl *p = impl_of(this);
    h &= HASH_MASK;
    StripeLock g(stripe_of(p, h));
    E[h].maxItem += add; // the bookkeeping word; the runs grow by themselves
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
// (here: the bucket arrays plus every page the arenas have touched)
std::string HashTable::GetSizeInfo() {
    uint64_t used = (uint64_t)HASH_SIZE * 2 * sizeof(uint32_t) + GetNbItem() * sizeof(ENTRY);
    uint64_t total = sizeof(E);
    if (Impl *p = find_impl(this)) {
        total += sizeof(p->bk);
        for (const Arena &a : p->arena) total += arena_touched(a);
        total += arena_touched(p->view_arena);
    } else {
        for (uint32_t h = 0; h < HASH_SIZE; h++) total += (uint64_t)E[h].nbItem * sizeof(ENTRY); // counts only (SeekNbItem)
    }
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

// bytes of HashTable.cpp:369-396 -- per bucket u32 nbItem, u32 maxItem, nbItem x 32 B in ascending x -- assembled in 8 MB
// pieces: the runs of a bucket, in index order, are the sorted bucket
void HashTable::SaveTable(FILE *f, uint32_t from, uint32_t to, bool printPoint) {
    Impl *p = find_impl(this);
    const uint64_t point = GetNbItem() / 16; // a dot every point + 1 entries (HashTable.cpp:367-385)
    uint64_t pending = 0;
    std::vector<char> out;
    out.reserve((size_t)8 << 20);
    auto put = [&](const void *src, size_t n) {
        if (out.size() + n > out.capacity() && !out.empty()) {
            fwrite(out.data(), 1, out.size(), f);
            out.clear();
        }
        if (n > out.capacity()) {
            fwrite(src, 1, n, f);
            return;
        }
        out.insert(out.end(), static_cast<const char *>(src), static_cast<const char *>(src) + n);
    };
    for (uint32_t h = from; h < to; h++) {
        HASH_ENTRY &e = E[h];
        const uint32_t head[2] = {e.nbItem, e.maxItem};
        put(head, sizeof head);
        if (!e.nbItem || !p || !p->bk[h].fine) continue; // (counts only: nothing to write, as in the reference with items == NULL ... it would crash there)
        {
            StripeLock g(stripe_of(p, h));
            const Bucket &b = p->bk[h];
            const size_t nf = (size_t)1 << b.k;
            for (size_t i = 0; i < nf; i++)
                if (b.fine[i].n) put(b.fine[i].e, (size_t)b.fine[i].n * sizeof(kngt_entry));
        }
        if (printPoint) {
            pending += e.nbItem;
            while (pending > point) {
                ::printf(".");
                pending -= point + 1;
            }
        }
    }
    if (!out.empty()) fwrite(out.data(), 1, out.size(), f);
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
    std::vector<kngt_entry> row;
    std::lock_guard<std::mutex> vg(p->view_lock);
    for (uint32_t h = from; h < to; h++) {
        HASH_ENTRY &e = E[h];
        uint32_t nb = 0, mx = 0;
        if (fread(&nb, sizeof(uint32_t), 1, f) != 1 || fread(&mx, sizeof(uint32_t), 1, f) != 1) break;
        e.maxItem = mx;
        if (!nb) continue;
        row.resize(nb);
        const size_t got = fread(row.data(), sizeof(kngt_entry), nb, f);
        // a file the reference wrote is strictly ascending; anything else is kept, in the order the searches need
        bool ascending = true;
        for (size_t i = 1; i < got && ascending; i++) ascending = kng_bucket::cmp_x(row[i - 1].x, row[i].x) < 0;
        if (!ascending) {
            for (size_t i = 1; i < got; i++) { // stable insertion sort: files are (almost) sorted
                const kngt_entry v = row[i];
                size_t j = i;
                while (j && kng_bucket::cmp_x(row[j - 1].x, v.x) > 0) {
                    row[j] = row[j - 1];
                    j--;
                }
                row[j] = v;
            }
        }
        Stripe &s = stripe_of(p, h);
        StripeLock g(s);
        Bucket &b = p->bk[h];
        if (!kng_bucket::build(*s.arena, b, kng_bucket::k_for((uint32_t)got), row.data(), (uint32_t)got)) die("out of memory in the distinguished-point table");
        b.n = (uint32_t)got;
        e.nbItem = (uint32_t)got;
        build_view(p, this, h); // what Check.cpp:47,88 reads
        if (got != nb) break;
    }
    p->views_live.store(true, std::memory_order_release);
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
    // three look-ahead stages turn the dependent misses of one insertion (bucket header -> run header -> run) into
    // independent misses of different insertions.  The reads ahead of the lock are hints only (kng_bucket::prefetch).
    // (KNG_HT_LOOKAHEAD="A,B,C": measurement knob, points of look-ahead per stage)
    static const struct Look {
        uint32_t a = 24, b = 16, c = 8;
        Look() {
            unsigned x, y, z;
            const char *e = getenv("KNG_HT_LOOKAHEAD");
            if (e && sscanf(e, "%u,%u,%u", &x, &y, &z) == 3 && x >= y && y >= z && x <= 4096) a = x, b = y, c = z;
        }
    } look;
    const uint32_t A = look.a, B = look.b, C = look.c;
    for (uint32_t i = 0; i < n + A; i++) {
        if (i < n) kng_bucket::prefetch(p->bk[recs[i].x[2] & HASH_MASK], recs[i].x[1], 0);
        if (i >= A - B && i - (A - B) < n) kng_bucket::prefetch(p->bk[recs[i - (A - B)].x[2] & HASH_MASK], recs[i - (A - B)].x[1], 1);
        if (i >= A - C && i - (A - C) < n) kng_bucket::prefetch(p->bk[recs[i - (A - C)].x[2] & HASH_MASK], recs[i - (A - C)].x[1], 2);
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
        const int st = insert(p, ht, r.x[2] & HASH_MASK, r.x[0], r.x[1], d0, d1, &stored);
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

// E[h].items for every bucket, as the reference keeps it after every Add: nbItem pointers to the entries in ascending x.
// Valid until the next insertion (which drops all views): for a reader of the public array between insertions -- the
// differential probes; the reference's own readers come after LoadTable, which builds the views itself.
void kng_ht_normalize(HashTable *ht) {
    Impl *p = find_impl(ht);
    if (!p) return;
    std::lock_guard<std::mutex> vg(p->view_lock);
    arena_release(p->view_arena);
    for (uint32_t h = 0; h < HASH_SIZE; h++) {
        StripeLock g(stripe_of(p, h));
        build_view(p, ht, h);
    }
    p->views_live.store(true, std::memory_order_release);
}
// (rounds 4-5: the length of the unsorted tail a bucket could carry; the runs of kng_bucket.h are sorted at all times)
void kng_ht_set_tail(HashTable *ht, uint32_t tail) {
    (void)ht;
    (void)tail;
}

// gives the table's memory AND its registry slot back (the class has no destructor: a program that creates and deletes
// tables calls this before `delete`; one that does not simply keeps up to 1024 of them registered)
void kng_ht_release(HashTable *ht) {
    std::lock_guard<std::mutex> g(g_registry);
    for (int i = 0; i < MAX_TABLES; i++)
        if (g_owner[i].load(std::memory_order_acquire) == ht) {
            release_memory(g_impl[i]);
            g_impl[i]->owner = nullptr;
            g_owner[i].store(nullptr, std::memory_order_release);
            g_registry_gen.fetch_add(1, std::memory_order_acq_rel);
            return;
        }
}

void kng_ht_stats(HashTable *ht, kng_ht_stats_t *out) {
    memset(out, 0, sizeof *out);
    out->entries = ht->GetNbItem();
    if (Impl *p = find_impl(ht)) {
        for (const Arena &a : p->arena) {
            out->bytes_mapped += arena_mapped(a);
            out->bytes_touched += arena_touched(a);
        }
        for (const Stripe &s : p->stripe) {
            out->grows += s.resplits;
            out->lock_spins += s.spins.load(std::memory_order_relaxed);
        }
    }
}
}
