/*
 * ingestprobe.cpp -- the queue between a GPU thread and its table threads (kangaroo_amd/host/kng_ingest.h, used by
 * SolveKeyGPU_kng.cpp) exercised without a GPU.  TEST INFRASTRUCTURE ONLY.  Linked with HashTable_kng.o and the reference's
 * SECPK1 objects by oracle/Makefile (_ref/ingestprobe).
 *
 *   ingestprobe <pushes> <threads> <cap-chunks> [seed]
 * One producer (what a GPU thread is) pushes `pushes` batches of 0..40000 engine records -- among them exact repeats and
 * same-x-other-distance records -- through an Ingest into the table's pool of `threads` owner-partitioned table threads, at
 * most `cap-chunks` of its chunks waiting, flushes now and then, collects the events.  A second Ingest on the SAME table (and
 * therefore the same pool) runs alongside from another producer thread (two GPUs, one table).  At the end: entries + events = records pushed, every event is a repeat or a collision of something pushed, the queue
 * never held more than its capacity, the producer was held back when the capacity is small, and an Ingest destroyed with work
 * still queued returns.
 */
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>
#include <vector>

#include "HashTable.h"
#include "SECPK1/SECP256k1.h"
#include "kng_ingest.h"

struct rng {
  uint64_t s;
  uint64_t next() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s * 0x2545F4914F6CDD1DULL;
  }
};

static const uint64_t off2[2] = {0, 1ULL << 60};

struct Feed {
  uint64_t pushed = 0, events = 0, dup = 0, coll = 0;
  double blocked = 0;
  size_t high_water = 0;
};

static void producer(HashTable *ht, int pushes, int threads, size_t cap, uint64_t seed, Feed *out) {
  rng g{seed};
  kng_ingest::Ingest ing(ht, off2, threads, cap);
  std::vector<kng_dp_record> batch;
  std::vector<kng_ingest::Event> ev;
  std::vector<kng_dp_record> seen;
  for (int p = 0; p < pushes; p++) {
    const uint32_t n = (uint32_t)(g.next() % 40001);
    batch.resize(n);
    for (uint32_t i = 0; i < n; i++) {
      kng_dp_record &r = batch[i];
      const uint64_t mode = g.next() % 64;
      if (mode == 0 && !seen.empty()) {
        r = seen[g.next() % seen.size()]; /* the same point again */
      } else if (mode == 1 && !seen.empty()) {
        r = seen[g.next() % seen.size()]; /* same x, another distance */
        r.d[0] ^= g.next() | 1;
      } else {
        for (int k = 0; k < 4; k++) r.x[k] = g.next();
        r.d[0] = g.next();
        r.d[1] = g.next() >> 4;
        r.kidx = g.next();
        r.reserved = 0;
      }
      if (seen.size() < 4096) seen.push_back(r);
      else if (mode == 2) seen[g.next() % seen.size()] = r;
    }
    out->blocked += ing.push(batch.data(), n);
    out->pushed += n;
    if (p % 7 == 3) ing.flush();
    ing.take_events(ev);
    for (const kng_ingest::Event &e : ev) {
      out->events++;
      if (e.status == ADD_DUPLICATE) out->dup++;
      else if (e.status == ADD_COLLISION) out->coll++;
      else {
        printf("unexpected event status %u\n", e.status);
        exit(1);
      }
    }
  }
  ing.flush();
  ing.take_events(ev);
  for (const kng_ingest::Event &e : ev) {
    out->events++;
    if (e.status == ADD_DUPLICATE) out->dup++;
    else out->coll++;
  }
  const kng_ingest::Ingest::Totals t = ing.totals();
  out->high_water = t.high_water;
  if (t.points != out->pushed) {
    printf("table threads handled %" PRIu64 " of %" PRIu64 " points\n", t.points, out->pushed);
    exit(1);
  }
}

int main(int argc, char **argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s <pushes> <threads> <cap-chunks> [seed]\n", argv[0]);
    return 2;
  }
  const int pushes = atoi(argv[1]), threads = atoi(argv[2]);
  const size_t cap = (size_t)atoll(argv[3]);
  const uint64_t seed = argc > 4 ? strtoull(argv[4], NULL, 0) : 0x1465E57ULL;
  Secp256K1 *secp = new Secp256K1();
  secp->Init();
  HashTable *ht = new HashTable();
  Feed a, b;
  std::thread second(producer, ht, pushes, threads, cap, seed * 3 + 1, &b); /* another GPU thread, the same table */
  producer(ht, pushes, threads, cap, seed, &a);
  second.join();
  const uint64_t entries = ht->GetNbItem();
  /* (a push is cut into one chunk per table thread: the bound a producer asks for is raised to two chunks per thread) */
  const size_t eff = cap < 2 * (size_t)threads ? 2 * (size_t)threads : cap;
  const bool ok = entries + a.events + b.events == a.pushed + b.pushed && a.high_water <= eff && b.high_water <= eff;
  printf("pushed %" PRIu64 " + %" PRIu64 " entries %" PRIu64 " events %" PRIu64 " (dup %" PRIu64 " coll %" PRIu64 ") high water %zu / %zu of %zu blocked %.3f s %s\n",
         a.pushed, b.pushed, entries, a.events + b.events, a.dup + b.dup, a.coll + b.coll, a.high_water, b.high_water, eff, a.blocked + b.blocked,
         ok ? "CONSISTENT" : "INCONSISTENT");
  /* shutdown with work still queued: must return, whatever was queued is dropped */
  {
    kng_ingest::Ingest ing(ht, off2, 1, 64);
    rng g{seed ^ 0xABCDEF};
    std::vector<kng_dp_record> batch(60000);
    for (kng_dp_record &r : batch) {
      for (int k = 0; k < 4; k++) r.x[k] = g.next();
      r.d[0] = g.next();
      r.d[1] = 1;
      r.kidx = 2;
      r.reserved = 0;
    }
    for (int i = 0; i < 6; i++) ing.push(batch.data(), (uint32_t)batch.size());
  }
  printf("shutdown with queued work: returned\n");
  /* hold (a work file's table section is being written, Backup_kng.cpp): after flush + hold the table does not change while
   * the producer keeps pushing -- beyond the normal bound, up to the bound for holds, where it blocks -- and the table threads
   * resume BY THEMSELVES once the generation is finished, with nobody awake to tell them (the producer is blocked in push);
   * every event carries the tag of its push */
  bool hold_ok = true;
  {
    std::atomic<uint64_t> finished{0};
    // (on the heap: ThreadSanitizer does not notice that a stack slot holds a NEW mutex when an earlier Ingest lived there)
    std::unique_ptr<kng_ingest::Ingest> ing_p(new kng_ingest::Ingest(ht, off2, threads, 4));
    kng_ingest::Ingest &ing = *ing_p;
    rng g{seed ^ 0x5A5A5A};
    std::vector<kng_dp_record> batch(kng_ingest::CHUNK);
    auto fresh = [&]() {
      for (kng_dp_record &r : batch) {
        for (int k = 0; k < 4; k++) r.x[k] = g.next();
        r.d[0] = g.next();
        r.d[1] = 1;
        r.kidx = 2;
        r.reserved = 0;
      }
    };
    fresh();
    ing.push(batch.data(), (uint32_t)batch.size(), 100);
    ing.flush();
    const uint64_t before = ht->GetNbItem();
    ing.hold(7, &finished, 12);
    std::thread releaser([&] {
      std::this_thread::sleep_for(std::chrono::milliseconds(300));
      hold_ok = hold_ok && ht->GetNbItem() == before && ing.holding(); /* nothing moved in 300 ms, 12 chunks waiting */
      finished = 7;
    });
    double blocked = 0;
    const std::vector<kng_dp_record> again = batch; /* pushed a second time below: 8192 duplicates, tagged */
    for (int i = 0; i < 14; i++) { /* 14 chunks against a hold bound of 12: the last pushes block until the release */
      fresh();
      blocked += ing.push(i == 5 ? again.data() : batch.data(), (uint32_t)batch.size(), 200 + (uint64_t)i);
    }
    releaser.join();
    ing.flush();
    std::vector<kng_ingest::Event> ev;
    ing.take_events(ev);
    uint64_t dup = 0;
    for (const kng_ingest::Event &e : ev) dup += (e.status == ADD_DUPLICATE && e.rec.reserved == 205) ? 1 : 0;
    const kng_ingest::Ingest::Totals t = ing.totals();
    const size_t hb = 12 < 2 * (size_t)threads ? 2 * (size_t)threads : 12; /* the bound while held */
    hold_ok = hold_ok && !ing.holding() && blocked > 0.2 && t.high_water == hb && dup == kng_ingest::CHUNK &&
              ht->GetNbItem() == before + 13 * (uint64_t)kng_ingest::CHUNK;
    printf("hold: table frozen for 300 ms, producer blocked %.3f s at %zu chunks (bound %zu), self-released, %" PRIu64 " tagged duplicates %s\n", blocked,
           t.high_water, hb, dup, hold_ok ? "CONSISTENT" : "INCONSISTENT");
  }
  /* more tables than the 16 registry slots of round 5: 20 alive at once, each keeps its own points;
   * released and deleted, 20 new ones at whatever addresses the heap hands out start empty */
  bool many_ok = true;
  for (int pass = 0; pass < 2; pass++) {
    std::vector<HashTable *> tabs;
    for (int t = 0; t < 20; t++) {
      HashTable *h = new HashTable();
      many_ok = many_ok && h->GetNbItem() == 0;
      rng g{seed + 977u * (uint64_t)t + (uint64_t)pass};
      for (int i = 0; i < 50 + t; i++) {
        int128_t x, d;
        x.i64[0] = g.next(); x.i64[1] = g.next();
        d.i64[0] = g.next(); d.i64[1] = g.next() >> 3;
        many_ok = many_ok && h->Add(g.next() & HASH_MASK, &x, &d) == ADD_OK;
      }
      tabs.push_back(h);
    }
    for (int t = 0; t < 20; t++) many_ok = many_ok && tabs[t]->GetNbItem() == (uint64_t)(50 + t);
    for (HashTable *h : tabs) {
      kng_ht_release(h);
      delete h;
    }
  }
  printf("tables: 2 x 20 alive at once, released and deleted %s\n", many_ok ? "CONSISTENT" : "INCONSISTENT");
  return ok && hold_ok && many_ok ? 0 : 1;
}
