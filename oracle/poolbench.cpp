/*
 * poolbench.cpp -- MEASUREMENT TOOL, test infrastructure only: the whole table side of the reference program with the
 * link-time replacements -- P producers (what the GPU threads of SolveKeyGPU_kng.cpp are: they copy 64-byte records into the
 * chunks of their owners, kng_ingest.h) feeding ONE `class HashTable` (HashTable_kng.o) through its pool of W owner-partitioned
 * table threads -- without a GPU.  Points are uniform, like distinguished points of a walk.
 *
 *   poolbench <points> <report-every> <table-threads W> <producers P> [points-per-push = 262144]
 * prints: entries, points/s of the whole table side over the last interval, ns per point and table thread (table-thread seconds
 * inside kng_ht_ingest / points), resident MB.  VERDICT r5 item 3: >= 120 M points/s with 16 threads, <= 110 ns per point and
 * thread up to 3e8 entries.
 */
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "HashTable.h"
#include "SECPK1/SECP256k1.h"
#include "kng_ingest.h"

static double now() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}
static double rss_mb() {
  long pages = 0, res = 0;
  FILE *f = fopen("/proc/self/statm", "r");
  if (f) {
    if (fscanf(f, "%ld %ld", &pages, &res) != 2) res = 0;
    fclose(f);
  }
  return res * (double)sysconf(_SC_PAGESIZE) / 1048576.0;
}
struct rng {
  uint64_t s;
  uint64_t next() {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s * 0x2545F4914F6CDD1DULL;
  }
};

int main(int argc, char **argv) {
  if (argc < 5) {
    fprintf(stderr, "usage: %s <points> <report-every> <table-threads> <producers> [points-per-push]\n", argv[0]);
    return 2;
  }
  const uint64_t points = strtoull(argv[1], NULL, 0), step = strtoull(argv[2], NULL, 0);
  const int W = atoi(argv[3]), P = atoi(argv[4]);
  const uint32_t per_push = argc > 5 ? (uint32_t)atol(argv[5]) : 262144;
  Secp256K1 *secp = new Secp256K1();
  secp->Init();
  HashTable *ht = new HashTable();
  static const uint64_t off2[2] = {0, 1ULL << 60};
  std::atomic<uint64_t> pushed{0};
  std::vector<double> busy((size_t)P, 0.0), blocked((size_t)P, 0.0);
  std::vector<std::thread> th;
  printf("# HashTable_kng.o behind kng_ingest.h: %d owner-partitioned table threads, %d producers, %u points per push\n", W, P, per_push);
  {
    /* where the table threads run: each reports the CPUs it may use once it has started (kng_placement.h; KNG_TABLE_PIN) */
    kng_ingest::Ingest probe(ht, off2, W, 64);
    printf("# table threads spread over %d NUMA node(s)%s\n", probe.nodes_used(), probe.nodes_used() ? "" : " (not confined)");
  }
  printf("# %12s %14s %22s %10s\n", "entries", "points/s", "ns/point/table-thread", "rss MB");
  std::atomic<int> alive{P};
  std::vector<std::atomic<uint64_t>> busy_ns((size_t)P), done_pts((size_t)P);
  for (auto &b : busy_ns) b = 0;
  for (auto &b : done_pts) b = 0;
  const double t_start = now();
  for (int p = 0; p < P; p++)
    th.emplace_back([&, p] {
      kng_ingest::Ingest ing(ht, off2, W, 64 * (size_t)(per_push / kng_ingest::CHUNK + 1));
      rng g{0x9E3779B97F4A7C15ULL * (uint64_t)(p + 1)};
      std::vector<kng_dp_record> recs(per_push);
      std::vector<kng_ingest::Event> ev;
      const uint64_t mine = points / (uint64_t)P;
      for (uint64_t at = 0; at < mine; at += per_push) {
        const uint32_t m = (uint32_t)(mine - at < per_push ? mine - at : per_push);
        for (uint32_t i = 0; i < m; i++) {
          kng_dp_record &r = recs[i];
          for (int k = 0; k < 4; k++) r.x[k] = g.next();
          r.d[0] = g.next();
          r.d[1] = g.next() >> 4;
          r.kidx = g.next();
          r.reserved = 0;
        }
        blocked[(size_t)p] += ing.push(recs.data(), m);
        ing.take_events(ev);
        pushed += m;
        const kng_ingest::Ingest::Totals tt = ing.totals();
        busy_ns[(size_t)p] = (uint64_t)(tt.busy_s * 1e9);
        done_pts[(size_t)p] = tt.points;
      }
      ing.flush();
      const kng_ingest::Ingest::Totals tt = ing.totals();
      busy_ns[(size_t)p] = (uint64_t)(tt.busy_s * 1e9);
      done_pts[(size_t)p] = tt.points;
      alive--;
    });
  /* both columns count points the table threads have FINISHED (the producers run ahead by what the queues hold) */
  uint64_t last = 0, next_report = step, last_busy = 0;
  double tl = now();
  for (;;) {
    usleep(20000);
    uint64_t d = 0, b = 0;
    for (auto &x : done_pts) d += x.load();
    for (auto &x : busy_ns) b += x.load();
    const bool over = alive.load() == 0;
    if (d >= next_report || over) {
      const double t = now();
      if (d > last)
        printf("  %12" PRIu64 " %14.0f %22.1f %10.0f\n", ht->GetNbItem(), (d - last) / (t - tl), (double)(b - last_busy) / (double)(d - last), rss_mb());
      fflush(stdout);
      last = d;
      last_busy = b;
      tl = t;
      next_report = d + step;
      if (over) break;
    }
  }
  for (std::thread &t : th) t.join();
  double bl = 0;
  for (double v : blocked) bl += v;
  printf("# %" PRIu64 " entries in %.2f s = %.1f M points/s over the whole run; producers blocked for queue room %.3f s in total\n", ht->GetNbItem(),
         now() - t_start, (double)ht->GetNbItem() / (now() - t_start) / 1e6, bl);
  return 0;
}
