// kng_ingest.h -- the table side of the GPU threads of SolveKeyGPU_kng.cpp: between the threads that drain engines and
// `class HashTable` (kng_ht_ingest, kng_hashtable_ext.h).  Needs nothing of the reference beyond the HashTable object it is
// handed, so the logic -- routing, back-pressure, flush, hold, events, shutdown with work still queued -- is tested on the CPU
// (oracle/ingestprobe.cpp, tests/test_hashtable_class_cpu.py).
//
// Round 6: OWNER-PARTITIONED table threads, one pool per table, shared by every GPU thread.  Round 5 gave each GPU thread its
// own table threads, each inserting whatever chunk came next: sixteen threads then wrote all over one table -- every insertion
// a stripe-lock line, a bucket header, a run header and a run last touched by some OTHER core, an arena lock shared with
// fifteen others -- and took 280-590 ns per point and thread (42-57 M points/s from 16 threads, three times slower than the
// repo's own table, whose consumers own their buckets: profiles/r05_htbench_threads_gpu_host.txt,
// r06_htbench_threads_gpu_host_locked.txt).  Now table thread w of W owns buckets
// [w, w + 1) * 2^18 / W: a producer (a GPU thread) copies each 64-byte record into the chunk it is filling for the record's
// owner -- the copy it made anyway -- and a table thread only ever touches its own buckets: its bucket headers stay in its
// cache, its locks and arenas are uncontended and local.  Same interface as before: Ingest = one producer's handle
// (push / flush / hold / take_events / totals).
#ifndef KNG_INGEST_H
#define KNG_INGEST_H

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include <unistd.h>

#include "kng_hashtable_ext.h"
#include "kng_placement.h"

namespace kng_ingest {

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr uint32_t CHUNK = 8192;          // points per hand-over: 512 KB, a few milliseconds of one table thread
constexpr size_t QUEUE_LAUNCHES = 64;     // a GPU thread stalls when this many launches' points are waiting ...
constexpr size_t QUEUE_MAX_CHUNKS = 2048; // ... or 1 GiB of them, whichever is less (64 launches at -d 8 would be 8 GB per GPU)
constexpr uint32_t BUCKET_BITS = 18;      // HASH_SIZE_BIT of HashTable.h

struct Producer;

struct Chunk {
  uint32_t n = 0;
  Producer *from = nullptr;
  kng_dp_record rec[CHUNK];
};

struct Event {
  kng_dp_record rec;
  uint32_t status;
  uint64_t stored_d[2];
};

// what the pool knows about one producer (one GPU thread)
struct Producer {
  std::mutex m;
  std::condition_variable room, idle;
  size_t outstanding = 0; // chunks handed over and not yet inserted
  size_t cap = 0, hold_cap = 0, high_water = 0;
  uint64_t points = 0;
  double busy_s = 0;
  std::vector<Event> events;
  // hold: this producer's chunks wait while *fin < held_gen (a work file's table section is being written)
  std::atomic<bool> held{false};
  std::atomic<uint64_t> held_gen{0};
  std::atomic<const std::atomic<uint64_t> *> fin{nullptr};
  std::atomic<bool> dead{false}; // its Ingest is being destroyed: whatever it still has queued is dropped
  bool still_held() {
    if (held.load(std::memory_order_acquire) && (dead.load(std::memory_order_relaxed) || fin.load()->load() >= held_gen.load())) {
      held.store(false, std::memory_order_release);
      room.notify_all();
    }
    return held.load(std::memory_order_relaxed);
  }
};

// the table threads of one HashTable
class Pool {
 public:
  Pool(HashTable *table, const uint64_t wild_off[2], int threads) : ht(table), W(threads < 1 ? 1 : threads), workers((size_t)W) {
    off[0] = wild_off[0];
    off[1] = wild_off[1];
    // On a machine with several NUMA nodes (eight GPUs hang off two sockets) table thread w of W is confined to the CPUs of
    // node w * nodes / W: the memory of its buckets -- first touched by it -- stays local for the whole run, as in the repo's
    // own solver (kng_placement.h).  KNG_TABLE_PIN=core: one physical core each; KNG_TABLE_PIN=0: no confinement.  A machine
    // with one usable node is left alone.
    const char *mode = getenv("KNG_TABLE_PIN");
    if (!(mode && (!strcmp(mode, "0") || !strcmp(mode, "off"))))
      plan = kng_placement::plan_consumers(kng_placement::numa_node_cpus(), W, mode && !strcmp(mode, "core"), (unsigned)getpid());
    for (int w = 0; w < W; w++) workers[(size_t)w].th = std::thread([this, w] { run(w); });
  }
  // NUMA nodes the table threads were spread over (0: not confined)
  int nodes_used() const {
    int used = 0, last = -1;
    for (const kng_placement::Placement &p : plan)
      if (p.pin && p.node != last) {
        used++;
        last = p.node;
      }
    return used;
  }
  ~Pool() {
    for (Worker &w : workers) {
      {
        std::lock_guard<std::mutex> l(w.m);
        w.stop = true;
      }
      w.cv.notify_all();
    }
    for (Worker &w : workers) w.th.join();
    for (Worker &w : workers)
      for (Chunk *c : w.q) delete c;
    for (Chunk *c : spare) delete c;
  }
  int threads() const { return W; }
  // the table thread that owns a bucket: contiguous ranges, so that an owner's bucket headers, stripe locks and arenas are its own
  uint32_t owner_of(uint64_t x2) const { return (uint32_t)(((x2 & ((1u << BUCKET_BITS) - 1)) * (uint64_t)W) >> BUCKET_BITS); }
  Chunk *fresh() {
    {
      std::lock_guard<std::mutex> l(spare_m);
      if (!spare.empty()) {
        Chunk *c = spare.back();
        spare.pop_back();
        return c;
      }
    }
    return new Chunk();
  }
  void submit(uint32_t owner, Chunk *c) {
    Worker &w = workers[owner];
    {
      std::lock_guard<std::mutex> l(w.m);
      w.q.push_back(c);
    }
    w.cv.notify_one();
  }

  // one pool per table, created by the first producer and destroyed with the last
  static std::shared_ptr<Pool> acquire(HashTable *table, const uint64_t wild_off[2], int threads) {
    std::lock_guard<std::mutex> l(registry_lock());
    std::weak_ptr<Pool> &slot = registry()[table];
    std::shared_ptr<Pool> p = slot.lock();
    if (!p) {
      p = std::make_shared<Pool>(table, wild_off, threads);
      slot = p;
    }
    return p;
  }

 private:
  struct Worker {
    std::mutex m;
    std::condition_variable cv;
    std::deque<Chunk *> q;
    bool stop = false;
    std::thread th;
  };
  static std::mutex &registry_lock() {
    static std::mutex m;
    return m;
  }
  static std::map<HashTable *, std::weak_ptr<Pool>> &registry() {
    static std::map<HashTable *, std::weak_ptr<Pool>> r;
    return r;
  }
  void run(int id) {
    if ((size_t)id < plan.size() && plan[(size_t)id].pin) (void)kng_placement::pin_this_thread(plan[(size_t)id].cpus, "table");
    Worker &w = workers[(size_t)id];
    std::vector<kng_ht_event> ev(CHUNK);
    for (;;) {
      Chunk *c = nullptr;
      {
        std::unique_lock<std::mutex> l(w.m);
        for (;;) {
          if (w.stop) return;
          bool waiting_held = false;
          for (auto it = w.q.begin(); it != w.q.end(); ++it) { // the first chunk whose producer is not on hold (per-producer order kept)
            if ((*it)->from->still_held()) {
              waiting_held = true;
              continue;
            }
            c = *it;
            w.q.erase(it);
            break;
          }
          if (c) break;
          // held chunks end their wait by themselves (nobody has to be awake to tell us): look again in 2 ms.  (system clock:
          // pthread_cond_timedwait, which ThreadSanitizer understands)
          if (waiting_held) w.cv.wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(2));
          else w.cv.wait(l);
        }
      }
      Producer &p = *c->from;
      uint32_t ne = 0;
      double dt = 0;
      if (!p.dead.load(std::memory_order_acquire)) {
        const double t0 = now_s();
        kng_ht_ingest(ht, c->rec, c->n, off, ev.data(), CHUNK, &ne);
        dt = now_s() - t0;
      }
      const uint32_t n = c->n;
      std::vector<Event> fresh_events;
      for (uint32_t i = 0; i < ne && i < CHUNK; i++) {
        Event e;
        e.rec = c->rec[ev[i].index];
        e.status = ev[i].status;
        e.stored_d[0] = ev[i].stored_d[0];
        e.stored_d[1] = ev[i].stored_d[1];
        fresh_events.push_back(e);
      }
      { // the chunk goes back BEFORE the producer learns that it is done: its destructor may then let the pool go
        std::lock_guard<std::mutex> l(spare_m);
        if (spare.size() < 4 * (size_t)W + 64) { // (what a hold made the queues grow beyond their normal bound goes back to the OS)
          spare.push_back(c);
          c = nullptr;
        }
      }
      delete c;
      {
        std::lock_guard<std::mutex> l(p.m);
        p.events.insert(p.events.end(), fresh_events.begin(), fresh_events.end());
        p.points += n;
        p.busy_s += dt;
        p.outstanding--;
        // (notified under the lock: once `outstanding` reaches 0 the producer's Ingest may be destroyed at any moment)
        p.room.notify_one();
        p.idle.notify_all();
      }
    }
  }
  HashTable *ht;
  uint64_t off[2];
  const int W;
  std::vector<Worker> workers;
  std::vector<kng_placement::Placement> plan;
  std::mutex spare_m;
  std::vector<Chunk *> spare;
};

// one producer's handle: what a GPU thread of SolveKeyGPU_kng.cpp holds.  `threads` = size of the table's pool (every producer
// of a table names the same number: the pool is created by whoever comes first); max_chunks = how many of this producer's
// chunks may wait before push() blocks.
class Ingest {
 public:
  Ingest(HashTable *table, const uint64_t wild_off[2], int threads, size_t max_chunks) : pool(Pool::acquire(table, wild_off, threads)) {
    // a launch is cut into one chunk per owner: the bound must leave room for that
    me.cap = max_chunks < 2 * (size_t)pool->threads() ? 2 * (size_t)pool->threads() : max_chunks;
    stage.assign((size_t)pool->threads(), nullptr);
  }
  ~Ingest() {
    me.dead.store(true, std::memory_order_release); // table threads drop what is still queued (the search is over)
    for (Chunk *c : stage) delete c;
    std::unique_lock<std::mutex> l(me.m);
    me.idle.wait(l, [this] { return me.outstanding == 0; });
  }
  int threads() const { return pool->threads(); }
  int nodes_used() const { return pool->nodes_used(); }
  // Copy `n` records into the chunks of their owners and hand the chunks over; blocks while too many of this producer's chunks
  // wait.  Returns the seconds spent blocked.  `tag` goes into the `reserved` word of every copied record (the table does not
  // look at it): events carry it back, which lets the caller tell which launch a point came from.
  double push(const kng_dp_record *recs, uint32_t n, uint64_t tag = 0) {
    double blocked = 0;
    for (uint32_t i = 0; i < n; i++) {
      const uint32_t o = pool->owner_of(recs[i].x[2]);
      Chunk *&c = stage[o];
      if (!c) {
        c = pool->fresh();
        c->n = 0;
        c->from = &me;
      }
      kng_dp_record &dst = c->rec[c->n++];
      dst = recs[i];
      dst.reserved = tag;
      if (c->n == CHUNK) blocked += hand_over(o);
    }
    for (uint32_t o = 0; o < stage.size(); o++) // a launch's points do not wait for the next launch to fill their chunk
      if (stage[o] && stage[o]->n) blocked += hand_over(o);
    return blocked;
  }
  // every point pushed is in the table (not while held: nothing would move)
  void flush() {
    std::unique_lock<std::mutex> l(me.m);
    me.idle.wait(l, [this] { return me.outstanding == 0; });
  }
  // Keep this producer's points out of the table while the table is being written to a work file (Backup.cpp:401-407 runs
  // HashTable::SaveTable with every thread parked; here the GPU goes on walking and its points wait in the queues, whose bound
  // rises to `cap_while_held` chunks meanwhile).  The hold ends by itself as soon as *finished >= generation: the table
  // threads look every 2 ms, so nobody has to be awake to release them -- the GPU thread may be blocked in push().
  // Call flush() first: then the table holds exactly the points of the launches this producer has drained so far.
  void hold(uint64_t generation, const std::atomic<uint64_t> *finished, size_t cap_while_held) {
    std::lock_guard<std::mutex> l(me.m);
    me.held_gen.store(generation);
    me.fin.store(finished);
    me.hold_cap = cap_while_held > me.cap ? cap_while_held : me.cap;
    me.held.store(true, std::memory_order_release);
  }
  bool holding() { return me.still_held(); }
  void take_events(std::vector<Event> &out) {
    std::lock_guard<std::mutex> l(me.m);
    out.swap(me.events);
    me.events.clear();
  }
  struct Totals {
    size_t high_water;
    uint64_t points;
    double busy_s; // table-thread seconds inside kng_ht_ingest for this producer's points
  };
  Totals totals() {
    std::lock_guard<std::mutex> l(me.m);
    return Totals{me.high_water, me.points, me.busy_s};
  }

 private:
  double hand_over(uint32_t owner) {
    double blocked = 0;
    {
      std::unique_lock<std::mutex> l(me.m);
      auto bound = [this] { return me.held.load(std::memory_order_relaxed) ? me.hold_cap : me.cap; };
      if (me.outstanding >= bound()) {
        const double t0 = now_s();
        // (timed: the end of a hold changes the bound without anybody notifying)
        while (me.outstanding >= bound()) me.room.wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(5));
        blocked = now_s() - t0;
      }
      me.outstanding++;
      if (me.outstanding > me.high_water) me.high_water = me.outstanding;
    }
    Chunk *c = stage[owner];
    stage[owner] = nullptr;
    pool->submit(owner, c);
    return blocked;
  }
  Producer me;
  std::shared_ptr<Pool> pool;
  std::vector<Chunk *> stage;
};

} // namespace kng_ingest
#endif
