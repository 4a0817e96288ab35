// kng_ingest.h -- the table side of one GPU thread of SolveKeyGPU_kng.cpp: a bounded queue of 8192-point chunks between the
// thread that drains an engine and the threads that insert into `class HashTable` with kng_ht_ingest (kng_hashtable_ext.h).
// Needs nothing of the reference beyond the HashTable object it is handed, so the queue logic -- back-pressure, flush, events,
// shutdown with work still queued -- is tested on the CPU (oracle/ingestprobe.cpp, tests/test_hashtable_class_cpu.py).
#ifndef KNG_INGEST_H
#define KNG_INGEST_H

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "kng_hashtable_ext.h"

namespace kng_ingest {

inline double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

constexpr uint32_t CHUNK = 8192;       // points per hand-over: 512 KB, a few milliseconds of one table thread
constexpr size_t QUEUE_LAUNCHES = 64;  // the GPU thread stalls when this many launches' points are waiting ...
constexpr size_t QUEUE_MAX_CHUNKS = 2048; // ... or 1 GiB of them, whichever is less (64 launches at -d 8 would be 8 GB per GPU)

struct Chunk {
  uint32_t n = 0;
  kng_dp_record rec[CHUNK];
};

struct Event {
  kng_dp_record rec;
  uint32_t status;
  uint64_t stored_d[2];
};

// the table side of one GPU thread
class Ingest {
 public:
  Ingest(HashTable *table, const uint64_t wild_off[2], int threads, size_t max_chunks) : ht(table), cap(max_chunks) {
    off[0] = wild_off[0];
    off[1] = wild_off[1];
    for (int t = 0; t < threads; t++) workers.emplace_back([this] { run(); });
  }
  ~Ingest() {
    {
      std::lock_guard<std::mutex> l(m);
      stop = true;
    }
    work.notify_all();
    for (std::thread &t : workers) t.join();
    for (Chunk *c : queue) delete c;
    for (Chunk *c : spare) delete c;
  }
  // copy `n` records into chunks and queue them; blocks while the queue is full.  Returns the seconds spent blocked.
  // `tag` goes into the `reserved` word of every copied record (the table does not look at it): events carry it back, which
  // lets the caller tell which launch a point came from.
  double push(const kng_dp_record *recs, uint32_t n, uint64_t tag = 0) {
    double blocked = 0;
    for (uint32_t at = 0; at < n; at += CHUNK) {
      const uint32_t k = n - at < CHUNK ? n - at : CHUNK;
      Chunk *c = nullptr;
      {
        std::unique_lock<std::mutex> l(m);
        if (queue.size() + busy >= (held ? hold_cap : cap)) {
          const double t0 = now_s();
          room.wait(l, [this] { return queue.size() + busy < (held ? hold_cap : cap) || stop; });
          blocked += now_s() - t0;
        }
        if (!spare.empty()) {
          c = spare.back();
          spare.pop_back();
        }
      }
      if (!c) c = new Chunk();
      memcpy(c->rec, recs + at, (size_t)k * sizeof(kng_dp_record));
      for (uint32_t i = 0; i < k; i++) c->rec[i].reserved = tag;
      c->n = k;
      {
        std::lock_guard<std::mutex> l(m);
        queue.push_back(c);
        if (queue.size() + busy > high_water) high_water = queue.size() + busy;
      }
      work.notify_one();
    }
    return blocked;
  }
  // every queued point is in the table (not while held: nothing would move)
  void flush() {
    std::unique_lock<std::mutex> l(m);
    idle.wait(l, [this] { return (queue.empty() && busy == 0) || stop; });
  }
  // Keep the table as it is while it is being written to a work file (Backup.cpp:401-407 runs HashTable::SaveTable with
  // every thread parked; here only the table threads pause -- the GPU goes on walking and its points wait in the queue, whose
  // bound rises to `cap_while_held` chunks meanwhile).  The hold ends by itself as soon as *finished >= generation: the
  // table threads look every 2 ms, so nobody has to be awake to release them -- the GPU thread may be blocked in push().
  // Call flush() first: then the table holds exactly the points of the launches drained so far.
  void hold(uint64_t generation, const std::atomic<uint64_t> *finished, size_t cap_while_held) {
    std::lock_guard<std::mutex> l(m);
    held = true;
    held_gen = generation;
    fin = finished;
    hold_cap = cap_while_held > cap ? cap_while_held : cap;
  }
  bool holding() {
    std::lock_guard<std::mutex> l(m);
    return still_held();
  }
  void take_events(std::vector<Event> &out) {
    std::lock_guard<std::mutex> l(m);
    out.swap(events);
    events.clear();
  }
  struct Totals {
    size_t high_water;
    uint64_t points;
    double busy_s; // table-thread seconds inside kng_ht_ingest
  };
  Totals totals() {
    std::lock_guard<std::mutex> l(m);
    return Totals{high_water, points, busy_s};
  }

 private:
  void run() {
    std::vector<kng_ht_event> ev(CHUNK);
    for (;;) {
      Chunk *c;
      {
        std::unique_lock<std::mutex> l(m);
        for (;;) {
          if (stop) return;
          if (!still_held() && !queue.empty()) break;
          if (held) work.wait_until(l, std::chrono::system_clock::now() + std::chrono::milliseconds(2)); // (system clock: pthread_cond_timedwait, which ThreadSanitizer understands)
          else work.wait(l);
        }
        c = queue.front();
        queue.pop_front();
        busy++;
      }
      const double t0 = now_s();
      uint32_t ne = 0;
      kng_ht_ingest(ht, c->rec, c->n, off, ev.data(), CHUNK, &ne);
      const double dt = now_s() - t0;
      {
        std::lock_guard<std::mutex> l(m);
        for (uint32_t i = 0; i < ne && i < CHUNK; i++) {
          Event e;
          e.rec = c->rec[ev[i].index];
          e.status = ev[i].status;
          e.stored_d[0] = ev[i].stored_d[0];
          e.stored_d[1] = ev[i].stored_d[1];
          events.push_back(e);
        }
        points += c->n;
        busy_s += dt;
        if (spare.size() < cap) spare.push_back(c); // (what a hold made the queue grow beyond its normal bound goes back to the OS)
        else delete c;
        busy--;
      }
      room.notify_one();
      idle.notify_all();
    }
  }
  bool still_held() { // (m taken)
    if (held && fin->load() >= held_gen) {
      held = false;
      work.notify_all();
    }
    return held;
  }
  HashTable *ht;
  uint64_t off[2];
  size_t cap, hold_cap = 0;
  bool held = false;
  uint64_t held_gen = 0;
  const std::atomic<uint64_t> *fin = nullptr;
  size_t high_water = 0;
  uint64_t points = 0;
  double busy_s = 0;
  std::mutex m;
  std::condition_variable work, room, idle;
  std::deque<Chunk *> queue;
  std::vector<Chunk *> spare;
  std::vector<Event> events;
  size_t busy = 0;
  bool stop = false;
  std::vector<std::thread> workers;
};

} // namespace kng_ingest
#endif
