// kng_solver.cpp -- see kng_solver.h.  Product code: host pipeline over the C ABI of the engine.
#include "kng_solver.h"

#include <sched.h>
#include <unistd.h>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <sys/mman.h>
#include <sys/resource.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kangaroo_hip.h"
#include "kng_cpus.h"
#include "kng_dptable.h"
#include "kng_host.h"
#include "kng_placement.h"
#include "kng_workfile.h"

namespace {

thread_local std::string g_err;

int fail(const char *fmt, ...) {
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return -1;
}

using Clock = std::chrono::steady_clock;
double seconds_since(Clock::time_point t0) { return std::chrono::duration<double>(Clock::now() - t0).count(); }

struct U256 {
    uint64_t v[4];
};
bool is_zero(const uint64_t a[4]) { return (a[0] | a[1] | a[2] | a[3]) == 0; }
bool eq4(const uint64_t a[4], const uint64_t b[4]) { return std::memcmp(a, b, 32) == 0; }
// a - b over 256 bits (caller guarantees a >= b); shift right by one; bit length
U256 sub256(const uint64_t a[4], const uint64_t b[4]) {
    U256 r;
    unsigned __int128 br = 0;
    for (int i = 0; i < 4; i++) {
        const unsigned __int128 t = (unsigned __int128)a[i] - b[i] - br;
        r.v[i] = (uint64_t)t;
        br = (t >> 64) & 1;
    }
    return r;
}
U256 shr1(const U256 &a) {
    U256 r;
    for (int i = 0; i < 4; i++) r.v[i] = (a.v[i] >> 1) | (i < 3 ? a.v[i + 1] << 63 : 0);
    return r;
}
int bit_length(const U256 &a) {
    for (int i = 3; i >= 0; i--)
        if (a.v[i]) return 64 * i + 64 - __builtin_clzll(a.v[i]);
    return 0;
}
const uint64_t P_FIELD[4] = {0xFFFFFFFEFFFFFC2FULL, ~0ULL, ~0ULL, ~0ULL};
const uint64_t ZERO4[4] = {0, 0, 0, 0};

constexpr uint64_t INFLIGHT_LIMIT = 96ull << 20; // points queued for the table (x 48 B = 4.8 GB) beyond which GPU threads wait

// one distinguished point on its way to the table
struct DpMsg {
    kngt_entry e;
    uint32_t bucket;
    uint32_t gpu;
    uint64_t kidx;
};

// Messages travel in fixed-size chunks that come from -- and go back to -- a pool which only ever grows (slabs of huge pages,
// released when the solver is destroyed).  Round 3 moved one std::vector per consumer and launch and let the heap recycle them:
// at the 8-GPU rate that is ~6 GB/s of 4 KiB page faults on the GPU threads and, whenever the consumers fall behind and the
// vectors pile up, munmap + TLB shootdowns across all 40 threads -- the whole path stalled at ~100 M points/s whatever the
// number of consumers (profiles/r04_dp_host_before.txt).  In steady state a chunk now costs two short critical sections.
constexpr uint32_t CHUNK_MSGS = 2044; // 64-byte header + 2044 x 48 B: just under 96 KiB; a multiple of the staging group (ingest)
struct alignas(64) Chunk {
    Chunk *next;
    uint32_t n, pad;
    alignas(64) DpMsg m[CHUNK_MSGS];
};
static_assert(sizeof(DpMsg) == 48 && sizeof(Chunk) <= 96 * 1024, "64 chunks fit three huge pages");
struct ChunkPool {
    std::mutex m;
    Chunk *free_list = nullptr;
    std::vector<std::pair<void *, size_t>> slabs;
    uint64_t made = 0;
    Chunk *get() {
        {
            std::lock_guard<std::mutex> g(m);
            if (Chunk *c = free_list) {
                free_list = c->next;
                c->n = 0;
                return c;
            }
        }
        // a new slab: 64 chunks = 6 MiB, mapped outside the lock (several GPU threads may grow the pool at once)
        const size_t per = 64, bytes = (per * sizeof(Chunk) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
        void *mem = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        if (mem == MAP_FAILED) return nullptr;
        (void)madvise(mem, bytes, MADV_HUGEPAGE);
        Chunk *c = static_cast<Chunk *>(mem);
        std::lock_guard<std::mutex> g(m);
        slabs.emplace_back(mem, bytes);
        made += per;
        for (size_t i = 1; i < per; i++) {
            c[i].next = free_list;
            free_list = &c[i];
        }
        c[0].n = 0;
        return &c[0];
    }
    void put(Chunk *c) {
        std::lock_guard<std::mutex> g(m);
        c->next = free_list;
        free_list = c;
    }
    ~ChunkPool() {
        for (auto &sl : slabs) munmap(sl.first, sl.second);
    }
};

// whole cache lines, written around the caches (dst 64-byte aligned, bytes a multiple of 16)
inline void stream_copy(void *dst, const void *src, size_t bytes) {
#if defined(__SSE2__)
    __m128i *d = static_cast<__m128i *>(dst);
    const __m128i *sp = static_cast<const __m128i *>(src);
    for (size_t i = 0; i < bytes / 16; i++) _mm_stream_si128(d + i, _mm_loadu_si128(sp + i));
#else
    std::memcpy(dst, src, bytes);
#endif
}
inline void stream_fence() {
#if defined(__SSE2__)
    _mm_sfence();
#endif
}

struct Consumer {
    std::mutex m;
    std::condition_variable cv;
    std::deque<Chunk *> q;
    std::thread th;
    std::atomic<uint64_t> handled{0}; // points taken off the queue (load balance across consumers)
    std::atomic<uint64_t> busy_ns{0}; // time spent inserting (the rest of the run it slept on its queue)
    // the kernel's view of the thread, read when it ends: on-CPU time, time runnable but waiting for a CPU, context switches
    std::atomic<uint64_t> cpu_ns{0}, runq_ns{0}, nvcsw{0}, nivcsw{0};
    bool pin = false;                 // confine the thread to one NUMA node's CPUs (start_consumers)
    cpu_set_t cpus;
    uint64_t queued = 0;              // points waiting in q (under m)
    std::atomic<uint64_t> queued_high{0};
};

struct Worker {
    int index = 0, dev = 0, grid_x = 0, grid_y = 0;
    kng_engine *eng = nullptr;
    uint64_t n = 0;
    std::thread th;
    std::mutex reset_m;
    std::vector<uint64_t> resets; // kangaroos to replace (same-herd collisions), filled by the consumers
    std::atomic<uint64_t> launches{0};
    std::atomic<uint64_t> kernel_us_sum{0}; // walk-kernel time of all launches, microseconds
    // host work between two kng_wait calls (launch, drain, ingest, replacements): while it stays below the kernel time the
    // GPU never waits for its host thread.  `late` counts the launches where it did not.
    std::atomic<uint64_t> host_us_sum{0}, host_us_max{0}, ingest_us_max{0}, late{0};
    bool ended = false, paused = false; // guarded by kngs_solver::ctl_m
    uint64_t reset_seq = 0;
    bool pin = false;  // confine this GPU's host thread to the NUMA node its device hangs off (kngs_prepare)
    cpu_set_t cpus;
    int numa_node = -1;
    std::vector<Chunk *> out; // per-consumer chunk being filled by ingest()
    std::vector<DpMsg> stage;    // per-consumer staging groups of ingest() (software write-combining)
    std::vector<uint8_t> staged; // messages waiting in each group
};

} // namespace

struct kngs_solver {
    kngs_config cfg;
    unsigned instance = 0; // how many solvers this process created before this one (rotates KNGS_PIN=core's choice of cores)
    // derived (InitRange / InitSearchKey / CreateJumpTable)
    int range_power = 0, dp = 0;
    U256 wild_offset{};
    uint64_t skx[4], sky[4]; // keyToSearch = key - rangeStart*G
    uint64_t jd[32 * 2], jx[32 * 4], jy[32 * 4];
    uint64_t dp_mask = 0;

    kngt_table *table = nullptr;
    std::vector<Worker *> workers;
    std::vector<Consumer *> consumers;
    std::atomic<uint64_t> inflight{0}; // DP messages queued but not yet in the table
    ChunkPool pool; // message chunks: GPU threads take, consumers return

    // control
    std::mutex ctl_m;
    std::condition_variable ctl_cv;
    std::atomic<bool> stop{false}, solved{false}, failed{false}, consumers_quit{false};
    std::atomic<int> pause_req{0};
    std::mutex save_m;
    std::string error;
    bool started = false, joined = false;
    uint64_t priv[4] = {0, 0, 0, 0};

    // counters
    std::atomic<uint64_t> dps{0}, dps_lost{0}, same_herd{0}, wrong{0};
    uint64_t offset_count = 0;
    uint64_t warmup_jumps = 0; // part of offset_count: jumps of the discarded warm-up launches (kngs_prepare)
    double offset_seconds = 0;
    Clock::time_point t_start;
    double run_seconds = 0; // frozen at stop

    // restored herds
    kngw_file *herd_file = nullptr;
    uint64_t herd_left = 0, herd_total = 0;
    bool herd_spent = false; // a failed kngs_prepare had already taken kangaroos from the work file
    uint64_t herd_loaded = 0, herd_created = 0;
    uint64_t audits = 0, audited_kangaroos = 0, audit_mismatches = 0; // kngs_audit, cumulative
    uint64_t seed_used = 0;
    int numa_nodes = 0;       // nodes the consumers were spread over (0/1 = not pinned)
    double cpus = 0;          // CPUs the process may use (affinity, cgroup quota)
    bool prepared = false;    // engines created, herds in place (kngs_prepare)
    bool ingest_only = false; // kngs_start_ingest: consumers without engines (host-path measurements)
};

namespace {

// the flags are part of the predicates kngs_wait / the parked workers sleep on: change them under ctl_m, or a
// waiter that has just evaluated its predicate misses the notification
void set_error(kngs_solver *s, const std::string &msg) {
    std::lock_guard<std::mutex> g(s->ctl_m);
    if (s->error.empty()) s->error = msg;
    s->failed = true;
    s->stop = true;
    s->ctl_cv.notify_all();
}

// Kangaroo::CheckKey (Kangaroo.cpp:233-268) for the four sign combinations of (Td, Wd)
bool resolve(const kngs_solver *s, const uint64_t td[4], const uint64_t wd[4], uint64_t priv[4]) {
    for (int type = 0; type < 4; type++) {
        uint64_t d1[4], d2[4], pk[4], px[4], py[4];
        if (type & 1) kngh_sub_order(ZERO4, td, d1); else std::memcpy(d1, td, 32);
        if (type & 2) kngh_sub_order(ZERO4, wd, d2); else std::memcpy(d2, wd, 32);
        kngh_add_order(d1, d2, pk);
        if (kngh_pubkey(pk, px, py) != 0) continue;
        if (!eq4(px, s->skx)) continue;
        if (!eq4(py, s->sky)) kngh_sub_order(ZERO4, pk, pk); // P == -keyToSearch
        kngh_add_order(pk, s->cfg.range_start, priv);
        // Kangaroo::Output (:175-217): the answer must reproduce the public key
        if (kngh_pubkey(priv, px, py) == 0 && eq4(px, s->cfg.key_x) && eq4(py, s->cfg.key_y)) return true;
    }
    return false;
}

void request_reset(kngs_solver *s, const DpMsg &m) {
    Worker *w = s->workers[m.gpu];
    std::lock_guard<std::mutex> g(w->reset_m);
    w->resets.push_back(m.kidx);
}

using namespace kng_placement; // NUMA placement of GPU and table threads (kng_placement.h)

void consumer_main(kngs_solver *s, Consumer *c) {
    if (c->pin) (void)pin_this_thread(c->cpus, "table");
    struct AtExit {
        Consumer *c;
        ~AtExit() {
            if (FILE *f = fopen("/proc/thread-self/schedstat", "r")) {
                unsigned long long cpu = 0, runq = 0;
                if (fscanf(f, "%llu %llu", &cpu, &runq) == 2) {
                    c->cpu_ns = cpu;
                    c->runq_ns = runq;
                }
                fclose(f);
            }
            struct rusage ru;
            if (getrusage(RUSAGE_THREAD, &ru) == 0) {
                c->nvcsw = (uint64_t)ru.ru_nvcsw;
                c->nivcsw = (uint64_t)ru.ru_nivcsw;
            }
        }
    } at_exit{c};
    for (;;) {
        Chunk *chunk;
        {
            std::unique_lock<std::mutex> lk(c->m);
            c->cv.wait(lk, [&] { return !c->q.empty() || s->consumers_quit.load(); });
            if (c->q.empty()) return;
            chunk = c->q.front();
            c->q.pop_front();
            c->queued -= chunk->n;
        }
        const auto busy0 = Clock::now();
        const DpMsg *batch = chunk->m;
        const size_t nb = chunk->n;
        c->handled += nb;
        for (size_t bi = 0; bi < nb; bi++) {
            // the table is far larger than the caches: touch bucket header, run header and run a few points ahead
            if (bi + 12 < nb) kngt_prefetch(s->table, batch[bi + 12].bucket, batch[bi + 12].e.x[1], 0);
            if (bi + 8 < nb) kngt_prefetch(s->table, batch[bi + 8].bucket, batch[bi + 8].e.x[1], 1);
            if (bi + 4 < nb) kngt_prefetch(s->table, batch[bi + 4].bucket, batch[bi + 4].e.x[1], 2);
            const DpMsg &m = batch[bi];
            if (s->solved) break;
            kngt_entry other;
            const int st = kngt_add_entry(s->table, m.bucket, &m.e, &other);
            if (st == KNGT_ADD_OK) continue;
            if (st < 0) {
                set_error(s, "out of memory in the distinguished-point table");
                break;
            }
            bool replace = true; // AddToTable() == false (Kangaroo.cpp:599-606)
            if (st == KNGT_ADD_COLLISION) {
                uint64_t d_new[4], d_old[4];
                uint32_t t_new, t_old;
                kngt_decode(m.e.d, d_new, &t_new);
                kngt_decode(other.d, d_old, &t_old);
                if (t_new != t_old) {
                    uint64_t priv[4];
                    const uint64_t *td = t_new == 0 ? d_new : d_old, *wd = t_new == 0 ? d_old : d_new;
                    if (resolve(s, td, wd, priv)) {
                        {
                            std::lock_guard<std::mutex> g(s->ctl_m);
                            std::memcpy(s->priv, priv, 32);
                            s->solved = true;
                            s->stop = true;
                            s->ctl_cv.notify_all();
                        }
                        replace = false;
                    } else {
                        s->wrong++;
                    }
                }
            }
            if (replace) {
                s->same_herd++;
                request_reset(s, m);
            }
        }
        s->inflight -= nb;
        s->pool.put(chunk);
        c->busy_ns += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - busy0).count();
    }
}

// replace kangaroo kidx of worker w by a fresh one of the same type (CreateHerd(1,..) + SetKangaroo)
int replace_kangaroo(kngs_solver *s, Worker *w, uint64_t kidx) {
    uint64_t x[4], y[4], d[4], dd[4];
    const int type = (int)(kidx & 1);
    const uint64_t seed = s->seed_used ^ (0x9E3779B97F4A7C15ULL * (++w->reset_seq)) ^ ((uint64_t)w->index << 56) ^ kidx;
    if (kngh_create_herd(1, s->range_power, s->wild_offset.v, s->skx, s->sky, type, seed, 1, x, y, d) != 0)
        return fail("kngh_create_herd failed");
    if (type) kngh_add_order(d, s->wild_offset.v, dd); else std::memcpy(dd, d, 32);
    if (dd[2] | dd[3]) return fail("replacement distance does not fit 128 bits");
    if (kng_set_kangaroo(w->eng, kidx, x, y, dd) != KNG_OK) return fail("kng_set_kangaroo: %s", kng_last_error());
    return 0;
}

// The distinguished points of one launch, straight from the engine's pinned records: one pass turns each record
// into its table entry (GPUEngine.cu:672 + HashTable::Convert in one step) and appends it to the batch of the
// consumer that owns its bucket (the c-th of nc equal ranges of a scrambled bucket index: two multiplies, no division).
// Batch buffers come back from the consumers through a pool, so a launch touches no fresh pages.  At the 8-GPU DP
// size (262 144 points per 25 ms launch) this must stay well under 95 ns per point: tools/dp_ingest_bench.
inline size_t consumer_of(uint32_t bucket, size_t nc) {
    const uint32_t h = (bucket * 0x9E3779B1u) & (KNGT_BUCKETS - 1); // a bijection of the 18-bit bucket index that spreads neighbours
    return ((size_t)h * nc) >> KNGT_HASH_BITS;
}

void ingest(kngs_solver *s, Worker *w, const kng_dp_record *rec, uint32_t n) {
    struct Timed { // every exit path
        Worker *w;
        Clock::time_point t0 = Clock::now();
        ~Timed() {
            const uint64_t us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - t0).count();
            if (us > w->ingest_us_max.load(std::memory_order_relaxed)) w->ingest_us_max.store(us, std::memory_order_relaxed);
        }
    } timed{w};
    const size_t nc = s->consumers.size();
    if (w->out.size() != nc) w->out.assign(nc, nullptr);
    // back-pressure instead of unbounded memory: a host whose table threads cannot keep up makes its GPU threads wait here
    // (and shows it: late_launches, queue_high_points), it does not pile up gigabytes of messages
    while (s->inflight.load(std::memory_order_relaxed) > INFLIGHT_LIMIT && !s->stop && !s->failed) std::this_thread::sleep_for(std::chrono::microseconds(200));
    auto hand_over = [&](size_t c) {
        Chunk *ch = w->out[c];
        w->out[c] = nullptr;
        s->inflight += ch->n;
        Consumer *cs = s->consumers[c];
        bool wake;
        {
            std::lock_guard<std::mutex> g(cs->m);
            cs->queued += ch->n;
            if (cs->queued > cs->queued_high.load(std::memory_order_relaxed)) cs->queued_high.store(cs->queued, std::memory_order_relaxed);
            wake = cs->q.empty();
            cs->q.push_back(ch);
        }
        if (wake) cs->cv.notify_one();
    };
    // Software write-combining (the radix-partitioning idiom): a point's message first goes to a small per-consumer staging
    // group in this thread's L1; a full group -- four messages = three whole cache lines -- is written to the consumer's chunk
    // with streaming stores.  Without it every one of the nc output streams costs a read-for-ownership per line, of a line a
    // consumer on another core (or socket) read last: the hand-over slowed from 26 to 59 ns per point between an idle table and
    // 32 busy consumers, and the consumers then fetched every message line out of this core's cache
    // (profiles/r04_dp_host_chunks.txt).  Streamed lines bypass the caches both ways.
    constexpr uint32_t GROUP = 4;
    static_assert(CHUNK_MSGS % GROUP == 0 && (GROUP * sizeof(DpMsg)) % 64 == 0, "a staging group is whole cache lines");
    if (w->stage.size() != nc * GROUP) {
        w->stage.assign(nc * GROUP, DpMsg());
        w->staged.assign(nc, 0);
    }
    DpMsg *const stage = w->stage.data();
    uint8_t *const staged = w->staged.data();
    auto chunk_for = [&](size_t c) -> Chunk * {
        Chunk *ch = w->out[c];
        if (!ch) ch = w->out[c] = s->pool.get();
        return ch;
    };
    for (uint32_t i = 0; i < n; i++) {
        uint32_t bucket;
        kngt_entry e;
        kngt_encode_device(rec[i].x, rec[i].d, s->wild_offset.v, rec[i].kidx, &bucket, &e);
        const size_t c = consumer_of(bucket, nc);
        DpMsg &m = stage[c * GROUP + staged[c]];
        m.e = e;
        m.bucket = bucket;
        m.gpu = (uint32_t)w->index;
        m.kidx = rec[i].kidx;
        if (++staged[c] < GROUP) continue;
        staged[c] = 0;
        Chunk *ch = chunk_for(c);
        if (!ch) {
            set_error(s, "out of memory for distinguished-point messages");
            return;
        }
        stream_copy(&ch->m[ch->n], &stage[c * GROUP], GROUP * sizeof(DpMsg));
        ch->n += GROUP;
        if (ch->n == CHUNK_MSGS) {
            stream_fence();
            hand_over(c);
        }
    }
    // the launch's points reach the table within the launch: the staged remainders and the partly filled chunks go too
    stream_fence();
    for (size_t c = 0; c < nc; c++) {
        if (staged[c]) {
            Chunk *ch = chunk_for(c);
            if (!ch) {
                set_error(s, "out of memory for distinguished-point messages");
                return;
            }
            std::memcpy(&ch->m[ch->n], &stage[c * GROUP], staged[c] * sizeof(DpMsg)); // (n <= CHUNK_MSGS - GROUP here)
            ch->n += staged[c];
            staged[c] = 0;
        }
        if (w->out[c] && w->out[c]->n) hand_over(c);
    }
}

void worker_main(kngs_solver *s, Worker *w) {
    auto bail = [&](const std::string &msg) {
        set_error(s, msg);
        std::lock_guard<std::mutex> g(s->ctl_m);
        w->ended = true;
        s->ctl_cv.notify_all();
    };

    if (w->pin) (void)pin_this_thread(w->cpus, "GPU"); // next to its GPU: the ring it reads was written over that node's PCIe root
    if (kng_launch(w->eng) != KNG_OK) return bail(std::string("kng_launch: ") + kng_last_error());
    Clock::time_point host_t0{};
    bool have_host_t0 = false;
    for (;;) {
        uint64_t host_us = 0;
        if (have_host_t0) { // the host's share of the last cycle: everything between two kng_wait calls
            host_us = (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(Clock::now() - host_t0).count();
            w->host_us_sum += host_us;
            if (host_us > w->host_us_max.load(std::memory_order_relaxed)) w->host_us_max.store(host_us, std::memory_order_relaxed);
        }
        if (kng_wait(w->eng, 0) != KNG_OK) return bail(std::string("kng_wait: ") + kng_last_error());
        float ms = 0;
        kng_last_kernel_ms(w->eng, &ms);
        // The kernel this wait returned from was started FIRST in that host share; while the share is shorter than the
        // kernel the device never waits for this thread.  (Measured against the kernel's own duration, not the time to the
        // wait's return: on a device shared with other engines that includes their kernels.)
        if (have_host_t0 && (double)host_us > (double)ms * 1000.0) w->late++;
        host_t0 = Clock::now();
        have_host_t0 = true;
        w->kernel_us_sum += (uint64_t)(ms * 1000.0f + 0.5f);
        const uint64_t done = ++w->launches;
        const bool last = s->cfg.max_launches && done >= s->cfg.max_launches;
        const bool go_on = !s->stop && !s->pause_req && !last;
        // launch k+1 first, then look at the points of launch k (DP buffers are double-buffered)
        if (go_on && kng_launch(w->eng) != KNG_OK) return bail(std::string("kng_launch: ") + kng_last_error());

        uint32_t n_items = 0, n_lost = 0;
        const kng_dp_record *rec = nullptr;
        if (kng_drain_view(w->eng, &rec, &n_items, &n_lost) != KNG_OK) return bail(std::string("kng_drain_view: ") + kng_last_error());
        s->dps += n_items;
        s->dps_lost += n_lost;
        if (n_items) ingest(s, w, rec, n_items);
        // kangaroos the consumers asked to replace; stream-ordered behind the launch in flight
        std::vector<uint64_t> todo;
        {
            std::lock_guard<std::mutex> g(w->reset_m);
            todo.swap(w->resets);
        }
        for (uint64_t k : todo)
            if (replace_kangaroo(s, w, k) != 0) return bail(g_err);

        if (go_on) continue;
        // ---- launch boundary without a kernel in flight: park for a save, or end
        std::unique_lock<std::mutex> lk(s->ctl_m);
        if (!s->stop && !last) {
            if (s->pause_req) {
                w->paused = true;
                s->ctl_cv.notify_all();
                s->ctl_cv.wait(lk, [&] { return !s->pause_req || s->stop; });
                w->paused = false;
            }
            if (!s->stop) {
                lk.unlock();
                if (kng_launch(w->eng) != KNG_OK) return bail(std::string("kng_launch: ") + kng_last_error());
                continue;
            }
        }
        w->ended = true;
        s->ctl_cv.notify_all();
        return;
    }
}

// a pinned buffer of `piece` work-file records for streaming between a file and an engine's snapshot buffer (pageable when
// the driver refuses: slower copies, same result)
struct RecordStage {
    static constexpr uint64_t piece = 1u << 18; // 24 MB
    uint8_t *buf = nullptr;
    bool pinned = false;
    RecordStage() {
        buf = static_cast<uint8_t *>(kng_alloc_pinned(piece * 96));
        pinned = buf != nullptr;
        if (!buf) buf = static_cast<uint8_t *>(std::malloc(piece * 96));
    }
    ~RecordStage() {
        if (pinned) kng_free_pinned(buf);
        else std::free(buf);
    }
};

// upload the first `count` kangaroos of worker w from the open work file: the file's 96-byte records go to the device as they
// are and ONE kernel turns them into herd state, the wild offset added mod n (kng_snapshot_write / kng_snapshot_restore;
// rounds 3-5 converted every kangaroo on the host and uploaded x, y, d planes through kng_set_kangaroos_range)
int upload_from_file(kngs_solver *s, Worker *w, uint64_t count) {
    RecordStage st;
    if (!st.buf) return fail("out of memory");
    for (uint64_t c0 = 0; c0 < count; c0 += RecordStage::piece) {
        const uint64_t m = count - c0 < RecordStage::piece ? count - c0 : RecordStage::piece;
        if (kngw_get_records(s->herd_file, st.buf, m) != 0) return fail("%s", kngw_last_error());
        if (kng_snapshot_write(w->eng, c0, m, st.buf) != KNG_OK) return fail("kng_snapshot_write: %s", kng_last_error());
    }
    uint64_t bad = 0;
    if (kng_snapshot_restore(w->eng, 0, count, s->wild_offset.v, &bad) != KNG_OK) return fail("restored distance does not fit 128 bits: %s", kng_last_error());
    (void)kng_snapshot_release(w->eng); // a save brings the buffer back
    s->herd_left -= count;
    s->herd_loaded += count;
    return 0;
}

// the herd of worker w as the engine's last snapshot holds it -> the kangaroo section of f.  May run while the worker walks
// on (kng_snapshot_read is the one call of the engine's C ABI another host thread may make meanwhile).
int dump_herd(Worker *w, kngw_file *f) {
    RecordStage st;
    if (!st.buf) return fail("out of memory");
    for (uint64_t c0 = 0; c0 < w->n; c0 += RecordStage::piece) {
        const uint64_t m = w->n - c0 < RecordStage::piece ? w->n - c0 : RecordStage::piece;
        if (kng_snapshot_read(w->eng, c0, m, st.buf) != KNG_OK) return fail("kng_snapshot_read: %s", kng_last_error());
        if (kngw_put_records(f, st.buf, m) != 0) return fail("%s", kngw_last_error());
    }
    return 0;
}

void join_all(kngs_solver *s) {
    if (!s->started || s->joined) return;
    for (Worker *w : s->workers)
        if (w->th.joinable()) w->th.join();
    // let the consumers finish what is queued, then quit
    while (s->inflight.load() && !s->solved && !s->failed) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    s->consumers_quit = true;
    for (Consumer *c : s->consumers) {
        c->cv.notify_all();
        if (c->th.joinable()) c->th.join();
    }
    s->run_seconds = seconds_since(s->t_start);
    s->joined = true;
}

} // namespace

extern "C" {

const char *kngs_last_error(void) { return g_err.c_str(); }

int kngs_create(const kngs_config *cfg, kngs_solver **out) {
    if (!cfg || !out) return fail("null argument");
    *out = nullptr;
    if (cfg->n_gpus < 1 || cfg->n_gpus > KNGS_MAX_GPUS) return fail("n_gpus must be 1..%d", KNGS_MAX_GPUS);
    if (std::memcmp(cfg->range_end, cfg->range_start, 32) == 0) return fail("empty range");
    for (int i = 3; i >= 0; i--) {
        if (cfg->range_end[i] > cfg->range_start[i]) break;
        if (cfg->range_end[i] < cfg->range_start[i]) return fail("range end below range start");
    }
    if (!kngh_on_curve(cfg->key_x, cfg->key_y)) return fail("the public key does not lie on the curve");
    kngs_solver *s = new kngs_solver();
    s->cfg = *cfg;
    s->instance = g_solver_instances.fetch_add(1);
    // InitRange (Kangaroo.cpp:877-890)
    const U256 width = sub256(cfg->range_end, cfg->range_start);
    s->range_power = bit_length(width);
    if (s->range_power > 125) { // device distances are 128-bit (README: 125-bit interval limit)
        const int rp = s->range_power;
        delete s;
        return fail("range width 2^%d exceeds the 125-bit limit", rp);
    }
    s->wild_offset = shr1(width);
    // InitSearchKey (:892-912): keyToSearch = key - rangeStart*G
    if (is_zero(cfg->range_start)) {
        std::memcpy(s->skx, cfg->key_x, 32);
        std::memcpy(s->sky, cfg->key_y, 32);
    } else {
        uint64_t rx[4], ry[4], nry[4];
        if (kngh_pubkey(cfg->range_start, rx, ry) != 0) {
            delete s;
            return fail("range start is a multiple of the group order");
        }
        const U256 neg = sub256(P_FIELD, ry);
        std::memcpy(nry, neg.v, 32);
        if (kngh_point_add(cfg->key_x, cfg->key_y, rx, nry, s->skx, s->sky) != 0) {
            delete s;
            return fail("the key is the start of the range"); // trivial: priv = range_start
        }
    }
    kngh_jump_table(s->range_power, s->jd, s->jx, s->jy);
    s->table = kngt_create();
    if (!s->table) {
        delete s;
        return fail("out of memory");
    }
    *out = s;
    return 0;
}

void kngs_destroy(kngs_solver *s) {
    if (!s) return;
    if (s->started && !s->joined) kngs_stop(s);
    for (Worker *w : s->workers) {
        if (w->eng) kng_destroy(w->eng);
        delete w;
    }
    for (Consumer *c : s->consumers) delete c;
    if (s->herd_file) kngw_close(s->herd_file);
    kngt_destroy(s->table);
    delete s;
}

int kngs_load(kngs_solver *s, const char *path) {
    if (!s || !path) return fail("null argument");
    if (s->started) return fail("kngs_load must precede kngs_start");
    kngw_header h;
    uint64_t n = 0;
    kngw_file *f = kngw_open(path, &h, s->table, &n);
    if (!f) return fail("%s", kngw_last_error());
    if (h.magic != KNGW_HEADW) {
        kngw_close(f);
        return fail("%s is a kangaroo-only file; a full work file is needed", path);
    }
    if (!eq4(h.range_start, s->cfg.range_start) || !eq4(h.range_end, s->cfg.range_end) || !eq4(h.key_x, s->cfg.key_x) ||
        !eq4(h.key_y, s->cfg.key_y)) {
        kngw_close(f);
        kngt_reset(s->table);
        return fail("%s was made for another range or key", path);
    }
    if (s->cfg.dp < 0) s->cfg.dp = (int32_t)h.dp_size; // LoadWork (Backup.cpp:163): the file's DP unless forced
    s->offset_count = h.total_count;
    s->offset_seconds = h.total_seconds;
    if (s->herd_file) kngw_close(s->herd_file);
    s->herd_file = f;
    s->herd_left = s->herd_total = n;
    s->herd_spent = false;
    return 0;
}

namespace {

double effective_cpus();

// everything kngs_start and kngs_start_ingest share: consumers, clock, state
void start_consumers(kngs_solver *s, int nc) {
    for (int c = 0; c < nc; c++) s->consumers.push_back(new Consumer());
    const std::vector<cpu_set_t> nodes = (s->cfg.flags & KNGS_FLAG_NO_PIN) ? std::vector<cpu_set_t>() : numa_node_cpus();
    const char *mode = getenv("KNGS_PIN");
    const bool per_core = mode && std::string(mode) == "core";
    const unsigned salt = s->instance + (unsigned)getpid();
    const std::vector<Placement> plan = plan_consumers(nodes, nc, per_core, salt);
    int used_nodes = 0;
    {
        std::vector<int> seen;
        for (int c = 0; c < nc; c++) {
            Consumer *cs = s->consumers[(size_t)c];
            cs->pin = plan[(size_t)c].pin;
            cs->cpus = plan[(size_t)c].cpus;
            if (plan[(size_t)c].node >= 0 && std::find(seen.begin(), seen.end(), plan[(size_t)c].node) == seen.end()) seen.push_back(plan[(size_t)c].node);
        }
        used_nodes = (int)seen.size();
    }
    s->numa_nodes = used_nodes;
    s->cpus = effective_cpus();
    { // the first launch must not pay for the pool: one chunk per (GPU thread, consumer) pair, touched once
        std::vector<Chunk *> warm;
        for (size_t i = 0; i < (size_t)nc * (s->workers.size() + 1); i++)
            if (Chunk *c = s->pool.get()) {
                for (size_t off = 0; off < sizeof(Chunk); off += 4096) reinterpret_cast<volatile char *>(c)[off] = 0;
                warm.push_back(c);
            }
        for (Chunk *c : warm) s->pool.put(c);
    }
    s->t_start = Clock::now();
    s->started = true;
    for (Consumer *c : s->consumers) c->th = std::thread(consumer_main, s, c);
}

// one table thread sustains 2-5 M inserts per second depending on the table size (tools/dp_ingest_bench); one GPU
// emits 1.3 M points/s at its own suggested DP size, eight GPUs 85 M/s at theirs (the suggestion shrinks with the
// population, Kangaroo.cpp:980-993) -- which 16 threads absorb on the GPU box's host (profiles/r02_dp_ingest_arena.txt):
// four threads per GPU, within half of the host's hardware threads
// CPUs this process may actually use: hardware threads, cut by its affinity mask and by the cgroup's CPU quota.  (The GPU
// boxes of this project show 256 hardware threads under a quota of 16 CPUs: threads beyond the quota do not run in parallel,
// they take turns -- 32 table threads were SLOWER than 16 there, 8 s of their 27 s spent runnable but waiting for a CPU,
// profiles/r04_dp_probe3.txt.)
double effective_cpus() { return kng_effective_cpus(); }

// A table thread inserts 11-14 M points per second of CPU time (75-90 ns each, tools/dp_table_bench); a GPU thread spends
// 16-25 ns per point handing them over.  One GPU emits 1.5 M points/s at its own suggested DP size, eight GPUs 100 M/s at
// theirs (the suggestion shrinks with the population, Kangaroo.cpp:980-993): four table threads per GPU give that 3-4x
// headroom -- but never more threads than the process has CPUs, because surplus threads under a CPU quota only take turns
// (the GPU threads need about a seventh of a CPU each; on the 16-CPU box 16 table threads absorbed 174 M points/s from
// eight flat-out feeders, 14 threads 128 M, 32 threads the same 174 M with half their time spent waiting for a CPU:
// profiles/r04_dp_host_after.txt).
int default_consumers(int asked, int n_gpus) {
    if (asked > 0) return asked;
    int nc = n_gpus == 1 ? 1 : 4 * n_gpus;
    const int room = (int)(effective_cpus() + 0.5);
    if (nc > room) nc = room;
    if (nc > 64) nc = 64;
    if (nc < 2 && n_gpus > 1) nc = 2;
    return nc < 1 ? 1 : nc;
}

uint64_t draw_seed() {
    std::random_device rd; // the reference seeds from the clock (Timer::getSeed32, main.cpp:177)
    uint64_t v = ((uint64_t)rd() << 32) ^ (uint64_t)rd() ^ (uint64_t)Clock::now().time_since_epoch().count();
    return v ? v : 1;
}

} // namespace

int kngs_prepare(kngs_solver *s) {
    if (!s) return fail("null argument");
    if (s->started) return fail("already started");
    if (s->prepared) return 0;
    if (s->herd_spent)
        return fail("an earlier kngs_prepare failed after kangaroos had been read from the work file: load the file again (kngs_load) before retrying");
    const kngs_config &cfg = s->cfg;
    // a failed start leaves the solver as it was before the call: no half-built workers to trip a second attempt.  The one
    // thing that cannot be put back is the read position of the work file: a retry would silently create fresh kangaroos in
    // place of the ones already taken, so it is refused until the file has been loaded again.
    auto undo = [&](int rc) {
        for (Worker *w : s->workers) {
            if (w->eng) kng_destroy(w->eng);
            delete w;
        }
        s->workers.clear();
        s->herd_loaded = s->herd_created = 0;
        if (s->herd_file && s->herd_left != s->herd_total) s->herd_spent = true;
        s->prepared = false;
        return rc;
    };
    // warm-up launches are a benchmarking aid: they walk the herd and throw the distinguished points away.  On a herd
    // restored from a work file that would drop trails the table never sees and under-count the work done.
    if (cfg.warmup_launches && s->herd_file && s->herd_left)
        return fail("warmup_launches with a loaded work file would discard distinguished points of the restored herd");
    // seed 0 = draw one: two runs (or a resumed run) must not rebuild the same herds -- their walks would retrace
    // trails already in the table and every point would come back as a duplicate
    s->seed_used = cfg.seed ? cfg.seed : draw_seed();
    uint64_t total = 0;
    for (int g = 0; g < cfg.n_gpus; g++) {
        Worker *w = new Worker();
        w->index = g;
        w->dev = cfg.gpu_ids[g];
        w->grid_x = cfg.grid_x;
        w->grid_y = cfg.grid_y;
        s->workers.push_back(w);
        if (w->grid_x <= 0 || w->grid_y <= 0) {
            if (kng_default_grid(w->dev, &w->grid_x, &w->grid_y) != KNG_OK)
                return undo(fail("kng_default_grid(%d): %s", cfg.gpu_ids[g], kng_last_error()));
        }
        w->n = (uint64_t)w->grid_x * (uint64_t)w->grid_y * KNG_GRP_SIZE;
        total += w->n;
    }
    // Run (Kangaroo.cpp:974-993): suggested DP size for the whole population
    s->dp = cfg.dp >= 0 ? cfg.dp : kngh_suggest_dp(s->range_power, (double)total);
    if (s->dp > 64) s->dp = 64;
    s->dp_mask = kngh_dp_mask(s->dp);
    for (Worker *w : s->workers) {
        uint32_t mf = cfg.max_found;
        if (mf == 0) {
            const double expect = (double)w->n * KNG_NB_RUN / (s->dp >= 63 ? 9.2e18 : (double)(1ULL << s->dp));
            const double want = 2.0 * expect < 131072.0 ? 131072.0 : 2.0 * expect;
            mf = want > 4.0e8 ? 400000000u : (uint32_t)want;
        }
        if (mf > s->cfg.max_found) s->cfg.max_found = mf; // one drain buffer size for every worker
    }
    const std::vector<cpu_set_t> nodes = (cfg.flags & KNGS_FLAG_NO_PIN) ? std::vector<cpu_set_t>() : numa_node_cpus();
    for (Worker *w : s->workers) {
        // The engine's pinned buffers (the DP rings the kernel writes over PCIe) are placed by the allocating thread's node:
        // create the engine from the node its device hangs off, and keep the GPU's host thread there (worker_main).
        cpu_set_t before;
        bool moved = false;
        w->numa_node = kng_device_numa_node(w->dev);
        // (looked up BY NODE ID; a node none of whose CPUs this process may use -- numactl, a cpuset -- pins nothing)
        int usable_nodes = 0;
        for (size_t k = 0; k < nodes.size(); k++) usable_nodes += CPU_COUNT(&nodes[k]) > 0;
        if (usable_nodes > 1 && node_usable(nodes, w->numa_node) && sched_getaffinity(0, sizeof before, &before) == 0) {
            w->pin = true;
            w->cpus = nodes[(size_t)w->numa_node];
            moved = pin_this_thread(w->cpus, "GPU");
            if (!moved) w->pin = false;
        }
        const int crc = kng_create(w->dev, w->grid_x, w->grid_y, s->cfg.max_found, &w->eng);
        if (moved) (void)sched_setaffinity(0, sizeof before, &before);
        if (crc != KNG_OK)
            return undo(fail("kng_create(gpu %d): %s", w->dev, kng_last_error()));
        if (kng_set_params(w->eng, s->dp_mask, s->jd, s->jx, s->jy) != KNG_OK) return undo(fail("kng_set_params: %s", kng_last_error()));
        // FetchWalks (Kangaroo.cpp:646-668): take what the file still holds, create the rest
        const uint64_t from_file = s->herd_file ? (s->herd_left < w->n ? s->herd_left : w->n) : 0;
        if (from_file < w->n) {
            // herd built on the device (kng_build_herd); each GPU gets its own stream of distances
            const uint32_t windows = (uint32_t)(s->range_power + 7) / 8;
            std::vector<uint64_t> table((size_t)windows * 256 * 8);
            uint64_t bt[8], bw[8], fin[8];
            const uint64_t seed = s->seed_used + 0x51ED270B1ULL * (uint64_t)(w->index + 1);
            if (kngh_herd_params(s->range_power, s->wild_offset.v, s->skx, s->sky, seed, table.data(), bt, bw, fin) != 0)
                return undo(fail("kngh_herd_params failed"));
            if (kng_build_herd(w->eng, s->range_power, seed, table.data(), windows, bt, bw, fin) != KNG_OK)
                return undo(fail("kng_build_herd: %s", kng_last_error()));
            s->herd_created += w->n - from_file;
        }
        // restored kangaroos overwrite the head of the herd (even start index: types stay aligned with parity)
        if (from_file && upload_from_file(s, w, from_file) != 0) return undo(-1);
    }
    if (s->herd_file) {
        kngw_close(s->herd_file);
        s->herd_file = nullptr;
    }
    // benchmarks: launches that are run and thrown away before the clock starts (clocks, caches, first-touch)
    uint64_t warm_jumps = 0;
    for (uint32_t i = 0; i < cfg.warmup_launches; i++) {
        for (Worker *w : s->workers)
            if (kng_launch(w->eng) != KNG_OK) return undo(fail("kng_launch: %s", kng_last_error()));
        for (Worker *w : s->workers) {
            const kng_dp_record *rec;
            uint32_t n_items, n_lost;
            if (kng_wait(w->eng, 0) != KNG_OK || kng_drain_view(w->eng, &rec, &n_items, &n_lost) != KNG_OK)
                return undo(fail("warm-up launch: %s", kng_last_error()));
            warm_jumps += w->n * KNG_NB_RUN;
        }
    }
    // the herds did advance, so the saved total must say so -- but only once every warm-up launch has succeeded (a retried
    // kngs_prepare must not count them twice), and as a figure of its own: their distinguished points were thrown away
    s->offset_count += warm_jumps;
    s->warmup_jumps = warm_jumps;
    s->prepared = true;
    return 0;
}

int kngs_start(kngs_solver *s) {
    if (!s) return fail("null argument");
    if (s->started) return fail("already started");
    if (kngs_prepare(s) != 0) return -1;
    const kngs_config &cfg = s->cfg;
    const int nc = default_consumers(cfg.consumers, cfg.n_gpus);
    start_consumers(s, nc);
    for (Worker *w : s->workers) w->th = std::thread(worker_main, s, w);
    return 0;
}

int kngs_start_ingest(kngs_solver *s, int feeders) {
    if (!s) return fail("null argument");
    if (s->started) return fail("already started");
    if (feeders < 1 || feeders > 256) return fail("feeders must be 1..256");
    s->ingest_only = true;
    s->seed_used = s->cfg.seed ? s->cfg.seed : draw_seed();
    for (int g = 0; g < feeders; g++) {
        Worker *w = new Worker();
        w->index = g;
        w->ended = true; // no GPU thread behind it
        s->workers.push_back(w);
    }
    start_consumers(s, default_consumers(s->cfg.consumers, feeders));
    return 0;
}

int kngs_ingest(kngs_solver *s, int feeder, const kng_dp_record *records, uint32_t n) {
    if (!s || (!records && n)) return fail("null argument");
    if (!s->ingest_only || !s->started || s->joined) return fail("kngs_ingest needs a solver started with kngs_start_ingest");
    if (feeder < 0 || (size_t)feeder >= s->workers.size()) return fail("no feeder %d", feeder);
    if (s->failed) return fail("%s", s->error.c_str());
    s->dps += n;
    if (n) ingest(s, s->workers[(size_t)feeder], records, n);
    return 0;
}

int kngs_drained(kngs_solver *s, double seconds) {
    if (!s) return fail("null argument");
    const auto t0 = Clock::now();
    while (s->inflight.load() && !s->failed) {
        if (seconds_since(t0) > seconds) return 0;
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    if (s->failed) return fail("%s", s->error.c_str());
    return 1;
}

int kngs_consumer_load(const kngs_solver *s, uint64_t *handled, int cap) {
    if (!s || (!handled && cap)) return fail("null argument");
    const int n = (int)s->consumers.size();
    for (int i = 0; i < n && i < cap; i++) handled[i] = s->consumers[(size_t)i]->handled.load();
    return n;
}

int kngs_gpu_stats(const kngs_solver *s, int gpu, uint64_t *launches, double *kernel_ms_sum, uint64_t *kangaroos) {
    if (!s) return fail("null argument");
    if (gpu < 0 || (size_t)gpu >= s->workers.size()) return fail("no gpu %d", gpu);
    const Worker *w = s->workers[(size_t)gpu];
    if (launches) *launches = w->launches.load();
    if (kernel_ms_sum) *kernel_ms_sum = (double)w->kernel_us_sum.load() * 1e-3;
    if (kangaroos) *kangaroos = w->n;
    return 0;
}

int kngs_gpu_option(const kngs_solver *s, int gpu, const char *key, int64_t *value) {
    if (!s || !key || !value) return fail("null argument");
    if (gpu < 0 || (size_t)gpu >= s->workers.size()) return fail("no gpu %d", gpu);
    const Worker *w = s->workers[(size_t)gpu];
    if (!w->eng) return fail("gpu %d has no engine yet (kngs_prepare)", gpu);
    if (std::strcmp(key, "numa_node") == 0) { // where this GPU's host thread and pinned buffers were placed (-1: not placed)
        *value = w->pin ? w->numa_node : -1;
        return 0;
    }
    if (kng_get_option(w->eng, key, value) != KNG_OK) return fail("%s", kng_last_error());
    return 0;
}

int kngs_host_stats(const kngs_solver *s, kngs_host_stats_t *out) {
    if (!s || !out) return fail("null argument");
    std::memset(out, 0, sizeof *out);
    const double run_s = s->started ? (s->joined ? s->run_seconds : seconds_since(s->t_start)) : 0.0;
    out->consumers = (uint32_t)s->consumers.size();
    double busy_sum = 0;
    for (const Consumer *c : s->consumers) {
        const double b = (double)c->busy_ns.load() * 1e-9;
        busy_sum += b;
        if (run_s > 0 && b / run_s > out->consumer_busy_max) out->consumer_busy_max = b / run_s;
        out->consumer_cpu_s += (double)c->cpu_ns.load() * 1e-9;
        out->consumer_runq_s += (double)c->runq_ns.load() * 1e-9;
        out->consumer_busy_s += b;
        out->consumer_nvcsw += c->nvcsw.load();
        out->consumer_nivcsw += c->nivcsw.load();
        const uint64_t qh = c->queued_high.load();
        if (qh > out->queue_high_points) out->queue_high_points = qh;
    }
    if (run_s > 0 && out->consumers) out->consumer_busy_mean = busy_sum / run_s / out->consumers;
    for (const Worker *w : s->workers) {
        const double hm = (double)w->host_us_max.load() * 1e-3, im = (double)w->ingest_us_max.load() * 1e-3;
        if (hm > out->host_ms_max) out->host_ms_max = hm;
        if (im > out->ingest_ms_max) out->ingest_ms_max = im;
        const uint64_t l = w->launches.load();
        if (l > 1) out->host_ms_mean += (double)w->host_us_sum.load() * 1e-3 / (double)(l - 1) / (double)s->workers.size();
        out->late_launches += w->late.load();
    }
    out->run_seconds = run_s;
    out->numa_nodes = (uint32_t)s->numa_nodes;
    out->effective_cpus = s->cpus > 0 ? s->cpus : effective_cpus();
    out->pin_failures = g_pin_failures.load();
    return 0;
}

// ---- placement, inspectable without a GPU (tests run these against a made-up sysfs tree, KNG_SYSFS_ROOT) ----
int kngs_plan_placement(int n_consumers, int per_core, unsigned salt, char *out, size_t cap) {
    if (!out || cap == 0 || n_consumers < 0) return fail("bad argument");
    const std::vector<cpu_set_t> nodes = numa_node_cpus();
    const std::vector<Placement> plan = plan_consumers(nodes, n_consumers, per_core != 0, salt);
    std::string text;
    for (size_t k = 0; k < nodes.size(); k++) text += "node " + std::to_string(k) + ": " + (CPU_COUNT(&nodes[k]) ? cpuset_text(nodes[k]) : std::string("-")) + "\n";
    for (size_t c = 0; c < plan.size(); c++)
        text += "consumer " + std::to_string(c) + ": node " + std::to_string(plan[c].node) + " " + (plan[c].pin ? "cpus " + cpuset_text(plan[c].cpus) : std::string("unconfined")) + "\n";
    snprintf(out, cap, "%s", text.c_str());
    return (int)nodes.size();
}

int kngs_gpu_thread_cpus(int device_numa_node, char *out, size_t cap) {
    if (!out || cap == 0) return fail("bad argument");
    const std::vector<cpu_set_t> nodes = numa_node_cpus();
    int usable_nodes = 0;
    for (size_t k = 0; k < nodes.size(); k++) usable_nodes += CPU_COUNT(&nodes[k]) > 0;
    const bool pin = usable_nodes > 1 && node_usable(nodes, device_numa_node);
    snprintf(out, cap, "%s", pin ? cpuset_text(nodes[(size_t)device_numa_node]).c_str() : "unconfined");
    return pin ? 1 : 0;
}

int kngs_try_pin(const char *cpulist) {
    cpu_set_t all, want, before;
    CPU_ZERO(&all);
    for (int c = 0; c < CPU_SETSIZE; c++) CPU_SET(c, &all);
    (void)parse_cpulist(cpulist ? cpulist : "", all, &want);
    if (sched_getaffinity(0, sizeof before, &before) != 0) return -1;
    const bool ok = pin_this_thread(want, "test");
    (void)sched_setaffinity(0, sizeof before, &before);
    return ok ? 1 : 0;
}

uint64_t kngs_pin_failures(void) { return g_pin_failures.load(); }

const kngt_table *kngs_table(const kngs_solver *s) { return s ? s->table : nullptr; }

int kngs_wait(kngs_solver *s, double seconds) {
    if (!s) return fail("null argument");
    if (!s->started) return fail("not started");
    std::unique_lock<std::mutex> lk(s->ctl_m);
    auto all_ended = [&] {
        for (Worker *w : s->workers)
            if (!w->ended) return false;
        return true;
    };
    // a huge or infinite timeout would overflow the clock's integer tick count
    seconds = seconds > 0 ? (seconds < 1e9 ? seconds : 1e9) : 0;
    const auto deadline = Clock::now() + std::chrono::duration_cast<Clock::duration>(std::chrono::duration<double>(seconds));
    while (!s->solved && !s->failed && !all_ended()) {
        if (s->ctl_cv.wait_until(lk, deadline) == std::cv_status::timeout) break;
    }
    if (s->failed) return fail("%s", s->error.c_str());
    if (s->solved) return 1;
    if (all_ended()) {
        lk.unlock();
        // the last batches may still be on their way into the table
        while (s->inflight.load() && !s->solved && !s->failed) std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (s->failed) return fail("%s", s->error.c_str());
        return s->solved ? 1 : 2;
    }
    return 0;
}

int kngs_stop(kngs_solver *s) {
    if (!s) return fail("null argument");
    if (!s->started) return 0;
    {
        std::lock_guard<std::mutex> g(s->ctl_m);
        s->stop = true;
        s->ctl_cv.notify_all();
    }
    join_all(s);
    if (s->failed) return fail("%s", s->error.c_str());
    return 0;
}

int kngs_result(const kngs_solver *s, uint64_t priv[4]) {
    if (!s || !priv) return fail("null argument");
    if (!s->solved) return fail("not solved");
    std::memcpy(priv, s->priv, 32);
    return 0;
}

int kngs_get_stats(const kngs_solver *s, kngs_stats *st) {
    if (!s || !st) return fail("null argument");
    std::memset(st, 0, sizeof *st);
    double kms = 0;
    int running = 0;
    {
        std::lock_guard<std::mutex> g(const_cast<kngs_solver *>(s)->ctl_m);
        for (const Worker *w : s->workers)
            if (!w->ended) running++;
    }
    for (const Worker *w : s->workers) {
        const uint64_t l = w->launches.load();
        st->launches += l;
        st->jumps += l * w->n * KNG_NB_RUN;
        st->kangaroos += w->n;
        kms += (double)w->kernel_us_sum.load() * 1e-3;
    }
    st->jumps += s->offset_count;
    st->dps = s->dps;
    st->dps_lost = s->dps_lost;
    st->same_herd = s->same_herd;
    st->wrong_collisions = s->wrong;
    st->table_items = s->joined ? kngt_count(s->table) : 0; // exact only when the consumers are quiet
    st->seconds = s->offset_seconds + (s->started ? (s->joined ? s->run_seconds : seconds_since(s->t_start)) : 0.0);
    st->kernel_ms_avg = st->launches ? kms / (double)st->launches : 0.0;
    st->dp = s->dp;
    st->range_power = s->range_power;
    st->solved = s->solved ? 1 : 0;
    st->running = s->started && !s->joined ? running : 0;
    st->seed = s->seed_used;
    st->herd_loaded = s->herd_loaded;
    st->herd_created = s->herd_created;
    st->table_bytes = s->joined || s->ingest_only ? kngt_memory_bytes(s->table) : 0;
    st->warmup_jumps = s->warmup_jumps;
    st->audits = s->audits;
    st->audited_kangaroos = s->audited_kangaroos;
    st->audit_mismatches = s->audit_mismatches;
    if (s->ingest_only) st->table_items = kngt_count(s->table); // racy while points are in flight: callers drain first
    return 0;
}

int kngs_collision_key(const kngs_solver *s, const uint64_t tame_d[4], const uint64_t wild_d[4], uint64_t priv[4]) {
    if (!s || !tame_d || !wild_d || !priv) return fail("null argument");
    return resolve(s, tame_d, wild_d, priv) ? 1 : 0;
}

namespace {
// SaveWork (Backup.cpp:446-470): wait until every GPU thread blocks at a launch boundary with no kernel in flight and every
// queued point is in the table.  The engines then belong to the caller until release_workers.
int park_workers(kngs_solver *s) {
    s->pause_req = 1;
    bool parked;
    {
        // a launch lasts tens of milliseconds; a worker that has not reached its boundary after two minutes is
        // stuck in the driver (the reference's SaveWork has the same guard, wtimeout, Backup.cpp:446-470)
        std::unique_lock<std::mutex> lk(s->ctl_m);
        parked = s->ctl_cv.wait_for(lk, std::chrono::seconds(120), [&] {
            for (Worker *w : s->workers)
                if (!w->paused && !w->ended) return false;
            return true;
        });
    }
    if (!parked) {
        {
            std::lock_guard<std::mutex> g(s->ctl_m);
            s->pause_req = 0;
        }
        s->ctl_cv.notify_all();
        return fail("timed out waiting for the GPU threads to reach a launch boundary");
    }
    while (s->inflight.load() && !s->failed) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    return 0;
}
void release_workers(kngs_solver *s) {
    {
        std::lock_guard<std::mutex> g(s->ctl_m);
        s->pause_req = 0;
    }
    s->ctl_cv.notify_all();
}
} // namespace

int kngs_save(kngs_solver *s, const char *path, int with_kangaroos) {
    if (!s || !path) return fail("null argument");
    if (!s->started) return fail("not started");
    std::lock_guard<std::mutex> save_lock(s->save_m);
    if (park_workers(s) != 0) return -1;
    int rc = 0;
    kngs_stats st;
    kngs_get_stats(s, &st);
    kngw_header h;
    std::memset(&h, 0, sizeof h);
    h.magic = KNGW_HEADW;
    h.dp_size = (uint32_t)s->dp;
    std::memcpy(h.range_start, s->cfg.range_start, 32);
    std::memcpy(h.range_end, s->cfg.range_end, 32);
    std::memcpy(h.key_x, s->cfg.key_x, 32);
    std::memcpy(h.key_y, s->cfg.key_y, 32);
    h.total_count = st.jumps;
    h.total_seconds = st.seconds;
    // The GPU threads are parked at a launch boundary and every queued point is in the table.  Each engine freezes its herd as
    // work-file records in its second device buffer (one kernel, kng_snapshot); header and table are written; then the GPU
    // threads are RELEASED and the kangaroo section is streamed from the snapshots while they walk on (rounds 3-5 kept every
    // GPU parked until the last kangaroo was on disk: 0.26 s per GPU at the default herd, one GPU after the other).
    if (with_kangaroos)
        for (Worker *w : s->workers)
            if (rc == 0 && kng_snapshot(w->eng, s->wild_offset.v) != KNG_OK) rc = fail("kng_snapshot(gpu %d): %s", w->dev, kng_last_error());
    kngw_file *f = rc == 0 ? kngw_create(path, &h, s->table, with_kangaroos ? st.kangaroos : 0) : nullptr;
    if (rc == 0 && !f) rc = fail("%s", kngw_last_error());
    release_workers(s);
    if (f) {
        if (with_kangaroos)
            for (Worker *w : s->workers)
                if (rc == 0) rc = dump_herd(w, f);
        const std::string keep = g_err;
        if (kngw_close(f) != 0 && rc == 0) rc = fail("%s", kngw_last_error());
        else if (rc != 0) g_err = keep;
    }
    return rc;
}

// Whole-run audit (kng_solver.h): the device re-derives every kangaroo of every herd -- and every entry of the table -- from
// its distance.  The reference can only do the table half, on the CPU, from a saved file (-wcheck, Check.cpp:141-411).
int kngs_audit(kngs_solver *s, int with_table, kngs_audit_result *out) {
    if (!s || !out) return fail("null argument");
    std::memset(out, 0, sizeof *out);
    if (s->ingest_only || s->workers.empty()) return fail("no engines to audit with");
    if (!s->prepared) return fail("not prepared");
    std::lock_guard<std::mutex> save_lock(s->save_m);
    const auto t0 = Clock::now();
    const bool running = s->started && !s->joined;
    if (running && park_workers(s) != 0) return -1;
    int rc = 0;
    // inputs of the device audit: the 16-window table and the offset points of this key (kngh_herd_params at 128 bits)
    std::vector<uint64_t> table((size_t)KNG_AUDIT_WINDOWS * 256 * 8);
    uint64_t bt[8], bw[8], fin[8];
    if (kngh_herd_params(128, s->wild_offset.v, s->skx, s->sky, s->seed_used ^ 0xA0D17ULL, table.data(), bt, bw, fin) != 0)
        rc = fail("kngh_herd_params failed");
    for (size_t g = 0; g < s->workers.size() && rc == 0; g++) {
        Worker *w = s->workers[g];
        if (kng_outstanding(w->eng)) { // a worker that ended on an error may have left one
            if (kng_wait(w->eng, 0) != KNG_OK) rc = fail("kng_wait: %s", kng_last_error());
        }
        uint64_t bad = 0, idx[8];
        if (rc == 0 && kng_audit_setup(w->eng, table.data(), bt, bw, fin) != KNG_OK) rc = fail("kng_audit_setup: %s", kng_last_error());
        if (rc == 0 && kng_audit_herd(w->eng, &bad, idx, 8) != KNG_OK) rc = fail("kng_audit_herd: %s", kng_last_error());
        if (rc) break;
        int64_t us = 0;
        kng_get_option(w->eng, "audit_us", &us);
        out->herd_ms += (double)us * 1e-3;
        out->kangaroos += w->n;
        for (uint64_t i = 0; i < bad && i < 8 && out->n_first_bad < 8; i++) out->first_bad[out->n_first_bad++] = ((uint64_t)g << 56) | idx[i];
        out->kangaroo_mismatches += bad;
    }
    if (rc == 0 && with_table) {
        // every table entry back to an engine record: x limbs 0-1 + the bucket bits, device distance, type; compare mode 1.
        // Blocks of buckets are converted by a few threads side by side (decode + one addition mod n per entry), then
        // audited on the engines in turn.
        const uint32_t BLOCK = 8192, PARTS = 8; // buckets per block, threads per block
        size_t turn = 0;
        std::vector<std::vector<kng_dp_record>> part(PARTS);
        std::vector<uint64_t> unfit(PARTS);
        for (uint32_t b0 = 0; b0 < KNGT_BUCKETS && rc == 0; b0 += BLOCK) {
            std::vector<std::thread> th;
            for (uint32_t p = 0; p < PARTS; p++)
                th.emplace_back([&, p] {
                    std::vector<kng_dp_record> &recs = part[p];
                    recs.clear();
                    unfit[p] = 0;
                    std::vector<kngt_entry> ent;
                    const uint32_t lo = b0 + p * (BLOCK / PARTS), hi = lo + BLOCK / PARTS;
                    for (uint32_t b = lo; b < hi; b++) {
                        const uint32_t cnt = kngt_bucket_count(s->table, b);
                        if (!cnt) continue;
                        ent.resize(cnt);
                        const uint32_t got = kngt_bucket_entries(s->table, b, ent.data(), cnt);
                        for (uint32_t i = 0; i < got; i++) {
                            uint64_t d[4], dd[4];
                            uint32_t type = 0;
                            kngt_decode(ent[i].d, d, &type);
                            if (type & 1) kngh_add_order(d, s->wild_offset.v, dd); else std::memcpy(dd, d, 32);
                            if (dd[2] | dd[3]) { // not a distance an engine can have produced
                                unfit[p]++;
                                continue;
                            }
                            kng_dp_record r;
                            r.x[0] = ent[i].x[0]; r.x[1] = ent[i].x[1]; r.x[2] = b; r.x[3] = 0;
                            r.d[0] = dd[0]; r.d[1] = dd[1];
                            r.kidx = type & 1;
                            r.reserved = 1;
                            recs.push_back(r);
                        }
                    }
                });
            for (auto &t : th) t.join();
            for (uint32_t p = 0; p < PARTS && rc == 0; p++) {
                out->table_points += unfit[p];
                out->table_mismatches += unfit[p];
                if (part[p].empty()) continue;
                Worker *w = s->workers[turn++ % s->workers.size()];
                uint64_t bad = 0;
                if (kng_audit_points(w->eng, part[p].data(), part[p].size(), &bad, nullptr, 0) != KNG_OK) {
                    rc = fail("kng_audit_points: %s", kng_last_error());
                    break;
                }
                int64_t us = 0;
                kng_get_option(w->eng, "audit_us", &us);
                out->table_ms += (double)us * 1e-3;
                out->table_points += part[p].size();
                out->table_mismatches += bad;
            }
        }
    }
    if (running) release_workers(s);
    out->seconds = seconds_since(t0);
    s->audits++;
    s->audited_kangaroos += out->kangaroos;
    s->audit_mismatches += out->kangaroo_mismatches + out->table_mismatches;
    return rc;
}

} // extern "C"
