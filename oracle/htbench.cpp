/*
 * htbench.cpp -- insertion cost of `class HashTable` as the table grows.  MEASUREMENT TOOL, test infrastructure only.
 * oracle/Makefile links it twice: _ref/htbench_ref with the reference's HashTable.o (the per-point path the reference program
 * uses, Kangaroo.cpp:594-612) and _ref/htbench_kng with kangaroo_amd/host/HashTable_kng.o (the same path, plus the batch
 * interface SolveKeyGPU_kng.cpp uses).  Points are uniform, like distinguished points of a walk (x is a field element).
 *
 *   htbench add    <points> <report-every>            one thread, HashTable::Add(Int*, Int*, type) per point
 *   htbench ingest <points> <report-every> <threads>  kng_ht_ingest, 32768-point batches, <threads> threads at once
 *   htbench ingestp <points> <report-every> <threads> the same, every thread feeding its own 1/<threads> of the buckets
 * prints: entries, ns per point (per thread) over the last interval, points/s of all threads, resident MB.
 */
#include <pthread.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "HashTable.h"
#include "SECPK1/SECP256k1.h"

struct probe_rec {
  uint64_t x[4], d[2], kidx, reserved;
};
extern "C" int kng_ht_ingest(HashTable *ht, const probe_rec *recs, uint32_t n, const uint64_t wild_off[2], void *ev, uint32_t ev_cap,
                             uint32_t *n_ev) __attribute__((weak));

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
static void fill(rng &g, probe_rec *r, uint32_t n) {
  for (uint32_t i = 0; i < n; i++) {
    for (int k = 0; k < 4; k++) r[i].x[k] = g.next();
    r[i].d[0] = g.next();
    r[i].d[1] = g.next() >> 4;
    r[i].kidx = g.next();
    r[i].reserved = 0;
  }
}

static HashTable *table;
static const uint64_t off2[2] = {0, 1ULL << 60};
static std::atomic<uint64_t> done_points;

struct job {
  uint64_t points;
  uint64_t seed;
  uint32_t lo, hi; /* buckets this thread feeds: all of them (ingest), or its own range (ingestp: what owner-partitioned table threads see) */
};
static void *ingest_worker(void *p) {
  job *j = (job *)p;
  rng g{j->seed};
  const uint32_t batch = 32768;
  std::vector<probe_rec> recs(batch);
  for (uint64_t at = 0; at < j->points; at += batch) {
    const uint32_t m = (uint32_t)(j->points - at < batch ? j->points - at : batch);
    fill(g, recs.data(), m);
    if (j->hi - j->lo != HASH_SIZE)
      for (uint32_t i = 0; i < m; i++) recs[i].x[2] = (recs[i].x[2] & ~(uint64_t)HASH_MASK) | (j->lo + (recs[i].x[2] & HASH_MASK) % (j->hi - j->lo));
    uint32_t ne = 0;
    kng_ht_ingest(table, recs.data(), m, off2, NULL, 0, &ne);
    done_points += m;
  }
  return NULL;
}

int main(int argc, char **argv) {
  if (argc < 4) {
    fprintf(stderr, "usage: %s add|ingest <points> <report-every> [threads]\n", argv[0]);
    return 2;
  }
  const bool partitioned = !strcmp(argv[1], "ingestp");
  const bool ingest = !strcmp(argv[1], "ingest") || partitioned;
  const uint64_t points = strtoull(argv[2], NULL, 0), step = strtoull(argv[3], NULL, 0);
  const int threads = argc > 4 ? atoi(argv[4]) : 1;
  Secp256K1 *secp = new Secp256K1();
  secp->Init();
  table = new HashTable();
  if (ingest && !kng_ht_ingest) {
    fprintf(stderr, "this build carries the reference's HashTable.o: no batch interface\n");
    return 2;
  }
  printf("# %s, %s, %d thread(s)\n", kng_ht_ingest ? "HashTable_kng.o" : "reference HashTable.o", argv[1], ingest ? threads : 1);
  printf("# %12s %12s %14s %10s\n", "entries", "ns/point", "points/s", "rss MB");
  if (!ingest) {
    rng g{0x9E3779B97F4A7C15ULL};
    const uint32_t batch = 32768;
    std::vector<probe_rec> recs(batch);
    Int off;
    off.SetInt32(0);
    off.bits64[1] = off2[1];
    double t0 = now(), spent = 0;
    uint64_t since = 0;
    for (uint64_t at = 0; at < points; at += batch) {
      fill(g, recs.data(), batch);
      const double a = now();
      for (uint32_t i = 0; i < batch; i++) { /* what GPUEngine::Launch + AddToTable do per point */
        Int x, d;
        x.SetInt32(0);
        d.SetInt32(0);
        memcpy(x.bits64, recs[i].x, 32);
        d.bits64[0] = recs[i].d[0];
        d.bits64[1] = recs[i].d[1];
        const uint32_t type = (uint32_t)(recs[i].kidx % 2);
        if (type == 1) d.ModSubK1order(&off);
        (void)table->Add(&x, &d, type);
      }
      spent += now() - a;
      since += batch;
      if (since >= step || at + batch >= points) {
        printf("  %12" PRIu64 " %12.1f %14.0f %10.0f\n", at + batch, spent / since * 1e9, since / spent, rss_mb());
        fflush(stdout);
        spent = 0;
        since = 0;
      }
    }
    (void)t0;
    return 0;
  }
  std::vector<job> jobs(threads);
  std::vector<pthread_t> tid(threads);
  for (int t = 0; t < threads; t++) {
    jobs[t] = {points / threads, 0x9E3779B97F4A7C15ULL * (t + 1), 0, HASH_SIZE};
    if (partitioned) {
      jobs[t].lo = (uint32_t)((uint64_t)HASH_SIZE * t / threads);
      jobs[t].hi = (uint32_t)((uint64_t)HASH_SIZE * (t + 1) / threads);
    }
    pthread_create(&tid[t], NULL, ingest_worker, &jobs[t]);
  }
  uint64_t last = 0, next_report = step;
  double tl = now();
  while (last < points / threads * threads) {
    usleep(20000);
    const uint64_t d = done_points.load();
    if (d >= next_report || d >= points / threads * threads) {
      const double t = now();
      /* includes generating the points (~25 ns each): an upper bound on the table's cost */
      printf("  %12" PRIu64 " %12.1f %14.0f %10.0f\n", d, (t - tl) * threads / (double)(d - last) * 1e9, (d - last) / (t - tl), rss_mb());
      fflush(stdout);
      last = d;
      tl = t;
      next_report = d + step;
    }
  }
  for (int t = 0; t < threads; t++) pthread_join(tid[t], NULL);
  return 0;
}
