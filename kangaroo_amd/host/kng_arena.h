// kng_arena.h -- the memory the two distinguished-point tables own (kng_dptable.cpp: the repo's table;
// HashTable_kng.cpp: `class HashTable` of the reference, HashTable.h:66-108, re-implemented for link-time replacement).
//
// Why the tables do not use malloc.  Dozens of threads growing small arrays through malloc, in one address space that
// gains gigabytes per second, serialise on the kernel's mmap lock (heap growth + page faults): measured on the 256-thread
// GPU host, 16 consumers inserted 30 M points/s and 96 consumers 11 M (profiles/r02_dp_ingest_*.txt).  An arena takes
// regions of 64 KiB doubling to 256 MiB from the OS (huge pages where the system grants them), carves blocks of
// 64, 96, 128, 192, ... bytes from them and recycles freed blocks through per-size free lists.  Nothing goes back to the OS
// before arena_release.  Large regions on purpose: address space is free (pages arrive on first touch; a mapping the kernel
// refuses is asked for again with MAP_NORESERVE, then in smaller pieces), but every mmap takes
// the process's mm lock for writing and has to wait for the page faults in flight on the neighbouring mapping it merges
// with.  With 2 MiB regions the table threads of an 8-GPU run issued ~4000 mmaps per second and spent two thirds of their
// time blocked behind one another (profiles/r04_dp_probe.txt: 16 consumers 138 M points/s, 32 consumers 117).
#ifndef KNG_ARENA_H
#define KNG_ARENA_H

#include <sys/mman.h>

#include <atomic>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace kng_arena {

// size classes 64, 96, 128, 192, 256, 384, ... bytes (class c: 64 << c/2, times 1.5 when c is odd): 64 B .. 1 GiB.
// Blocks above the region size get a mapping of their own (arena_alloc: want = sz): a skewed bucket must not end as
// "out of memory" while memory is there.
constexpr int MIN_CLASS = 0, MAX_CLASS = 48;
constexpr size_t REGION_MIN = (size_t)64 << 10; // an arena's first region; each further one doubles ...
constexpr int REGION_DOUBLINGS = 12;            // ... up to 256 MiB (a small table must not cost N arenas x 2 MiB)

inline size_t class_bytes(int c) { return ((size_t)(c & 1 ? 96 : 64)) << (c >> 1); }

inline int class_of(size_t bytes) {
    int c = MIN_CLASS;
    while (class_bytes(c) < bytes) c++;
    return c;
}

struct Arena {
    std::atomic_flag lock = ATOMIC_FLAG_INIT;
    char *cur = nullptr, *end = nullptr; // bump area of the current region
    void *free_list[MAX_CLASS + 1] = {};
    std::vector<std::pair<void *, size_t>> regions; // for munmap
    uint64_t bytes = 0;                             // taken from the OS (under the lock; readers elsewhere use the two atomics)
    std::atomic<uint64_t> mapped{0}, touched{0};    // = bytes / bytes less the untouched tail of the current region, for status lines
};

struct Locked {
    Arena &a;
    explicit Locked(Arena &ar) : a(ar) {
        while (a.lock.test_and_set(std::memory_order_acquire)) {
        }
    }
    ~Locked() { a.lock.clear(std::memory_order_release); }
};

// A plain mapping first: under the default overcommit policy running out of memory then shows HERE, as a refused mmap the
// callers turn into an error return or a message -- not later as SIGBUS / an OOM kill at first touch.  Only when
// that is refused is the same size tried with MAP_NORESERVE (an accounting limit may refuse a 256 MiB region of which a few
// pages will ever be touched); the callers' shrinking-region retry comes after both.
inline void *map_region(size_t want) {
    void *m = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED) m = mmap(nullptr, want, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return nullptr;
    if (want >= ((size_t)2 << 20)) (void)madvise(m, want, MADV_HUGEPAGE);
    return m;
}

inline void *arena_alloc(Arena &a, int c) {
    if (c > MAX_CLASS) return nullptr;
    Locked g(a);
    if (void *p = a.free_list[c]) {
        a.free_list[c] = *static_cast<void **>(p);
        return p;
    }
    const size_t sz = class_bytes(c);
    if ((size_t)(a.end - a.cur) < sz) {
        // the tail of the old region is recycled as smaller blocks
        for (int k = MAX_CLASS; k >= MIN_CLASS; k--)
            while ((size_t)(a.end - a.cur) >= class_bytes(k)) {
                *reinterpret_cast<void **>(a.cur) = a.free_list[k];
                a.free_list[k] = a.cur;
                a.cur += class_bytes(k);
            }
        size_t want = REGION_MIN << (a.regions.size() < (size_t)REGION_DOUBLINGS ? a.regions.size() : (size_t)REGION_DOUBLINGS);
        if (want < sz) want = sz;
        // an address-space limit (RLIMIT_AS, a strict overcommit policy) may refuse a 256 MiB mapping long before memory
        // is short: retry with smaller regions down to what this block needs
        void *m = map_region(want);
        while (!m && want > sz) {
            want = want / 2 > sz ? want / 2 : sz;
            m = map_region(want);
        }
        if (!m) return nullptr;
        a.regions.emplace_back(m, want);
        a.bytes += want;
        a.cur = static_cast<char *>(m);
        a.end = a.cur + want;
        a.mapped.store(a.bytes, std::memory_order_relaxed);
    }
    void *p = a.cur;
    a.cur += sz;
    a.touched.store(a.bytes - (uint64_t)(a.end - a.cur), std::memory_order_relaxed);
    return p;
}

inline void arena_free(Arena &a, void *p, int c) {
    if (!p) return;
    Locked g(a);
    *static_cast<void **>(p) = a.free_list[c];
    a.free_list[c] = p;
}

inline void arena_release(Arena &a) {
    for (auto &r : a.regions) munmap(r.first, r.second);
    a.regions.clear();
    a.cur = a.end = nullptr;
    for (void *&f : a.free_list) f = nullptr;
    a.bytes = 0;
    a.mapped.store(0, std::memory_order_relaxed);
    a.touched.store(0, std::memory_order_relaxed);
}

// mapped less the untouched tail of the current region; safe to read while other threads allocate (the status
// line's GetSizeInfo used to combine `bytes`, `cur` and `end` of three different moments)
inline uint64_t arena_touched(const Arena &a) { return a.touched.load(std::memory_order_relaxed); }
inline uint64_t arena_mapped(const Arena &a) { return a.mapped.load(std::memory_order_relaxed); }

} // namespace kng_arena
#endif
