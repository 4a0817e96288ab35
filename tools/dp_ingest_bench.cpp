// dp_ingest_bench.cpp -- the host DP path at the 8-GPU rate, without GPUs and without Python in the loop
// (companion of tools/dp_ingest_stress.py).  Feeder threads stand in for the per-GPU host threads:
// each hands `points` synthetic engine records per "launch" to kngs_ingest, optionally paced at --launch-ms.
// build: g++ -O2 -std=c++17 -Iinclude -Ikangaroo_amd/host -o tools/dp_ingest_bench tools/dp_ingest_bench.cpp \
//            -Lkangaroo_amd/lib -lkangaroo_host -lkangaroo_hip -Wl,-rpath,'$ORIGIN/../kangaroo_amd/lib' -lpthread
#include <sys/resource.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "kangaroo_hip.h"
#include "kng_host.h"
#include "kng_solver.h"

static inline uint64_t xs(uint64_t &s) {
    s ^= s << 13;
    s ^= s >> 7;
    s ^= s << 17;
    return s;
}

int main(int argc, char **argv) {
    int feeders = 8, consumers = 0, launches = 40;
    uint32_t points = 262144;
    double launch_ms = 0;
    uint32_t flags = 0;
    for (int i = 1; i < argc; i++)
        if (!strcmp(argv[i], "--no-pin")) { // consumers not confined to NUMA nodes
            flags |= KNGS_FLAG_NO_PIN;
            for (int j = i; j + 1 < argc; j++) argv[j] = argv[j + 1];
            argc--;
            i--;
        }
    for (int i = 1; i + 1 < argc; i += 2) {
        if (!strcmp(argv[i], "--feeders")) feeders = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--consumers")) consumers = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--launches")) launches = atoi(argv[i + 1]);
        else if (!strcmp(argv[i], "--points-per-launch")) points = (uint32_t)atol(argv[i + 1]);
        else if (!strcmp(argv[i], "--launch-ms")) launch_ms = atof(argv[i + 1]);
    }
    kngs_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.range_end[0] = ~0ULL;
    cfg.range_end[1] = 0xFFFF; // 80-bit range from 0
    uint64_t priv[4] = {0xC0FFEE123456789AULL, 0xBCD, 0, 0};
    if (kngh_pubkey(priv, cfg.key_x, cfg.key_y) != 0) return 1;
    cfg.dp = 11;
    cfg.n_gpus = 1;
    cfg.consumers = consumers;
    cfg.seed = 1;
    cfg.flags = flags;
    kngs_solver *s = nullptr;
    if (kngs_create(&cfg, &s) != 0 || kngs_start_ingest(s, feeders) != 0) {
        fprintf(stderr, "%s\n", kngs_last_error());
        return 1;
    }
    std::vector<double> fed_s(feeders), lag_ms(feeders);
    std::vector<std::vector<double>> per_launch(feeders, std::vector<double>(launches, 0.0));
    std::vector<std::thread> th;
    const auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < feeders; f++)
        th.emplace_back([&, f] {
            std::vector<kng_dp_record> rec(points);
            uint64_t st = 0x9E3779B97F4A7C15ULL * (uint64_t)(f + 1);
            double busy = 0, lag = 0;
            for (int l = 0; l < launches; l++) {
                for (auto &r : rec) { // what the kernel leaves in the pinned buffer
                    r.x[0] = xs(st);
                    r.x[1] = xs(st);
                    r.x[2] = xs(st);
                    r.x[3] = xs(st) >> 11;
                    r.d[0] = xs(st);
                    r.d[1] = 0x3FFF + (xs(st) & 0xFF);
                    r.kidx = xs(st) & 0x7FFFFF;
                    r.reserved = 0;
                }
                if (launch_ms > 0) {
                    const auto due = t0 + std::chrono::duration_cast<std::chrono::steady_clock::duration>(std::chrono::duration<double, std::milli>((l + 1) * launch_ms));
                    const auto now = std::chrono::steady_clock::now();
                    if (now < due) std::this_thread::sleep_until(due);
                    else lag = std::max(lag, std::chrono::duration<double, std::milli>(now - due).count());
                }
                const auto a = std::chrono::steady_clock::now();
                if (kngs_ingest(s, f, rec.data(), points) != 0) {
                    fprintf(stderr, "%s\n", kngs_last_error());
                    return;
                }
                const double took = std::chrono::duration<double>(std::chrono::steady_clock::now() - a).count();
                busy += took;
                per_launch[f][l] = took;
            }
            fed_s[f] = busy;
            lag_ms[f] = lag;
        });
    for (auto &t : th) t.join();
    const double t_fed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const int ok = kngs_drained(s, 900.0);
    const double t_all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    kngs_stats stt;
    kngs_get_stats(s, &stt);
    uint64_t load[256];
    const int nc = kngs_consumer_load(s, load, 256);
    const double total = (double)feeders * points * launches;
    double busy_max = 0, lag_max = 0;
    for (int f = 0; f < feeders; f++) busy_max = std::max(busy_max, fed_s[f]), lag_max = std::max(lag_max, lag_ms[f]);
    uint64_t lmin = ~0ULL, lmax = 0;
    for (int i = 0; i < nc; i++) lmin = std::min(lmin, load[i]), lmax = std::max(lmax, load[i]);
    printf("%d feeders x %d launches x %u points = %.1f M points, %d consumer threads, %u host threads\n", feeders, launches, points, total / 1e6, nc, std::thread::hardware_concurrency());
    printf("feeder side: kngs_ingest takes %.2f ms per launch of %u points = %.1f ns/point (a GPU thread has %.0f ms per launch)\n",
           busy_max / launches * 1e3, points, busy_max / launches / points * 1e9, launch_ms > 0 ? launch_ms : 25.0);
    if (launch_ms > 0) printf("paced at one launch per %.1f ms per feeder = %.1f M points/s offered; worst feeder lag %.1f ms; fed in %.2f s\n", launch_ms, feeders * points / launch_ms / 1e3, lag_max, t_fed);
    printf("end to end (every point in the table, drained=%d): %.2f s = %.1f M points/s\n", ok, t_all, total / t_all / 1e6);
    printf("table: %llu items, %.2f GiB = %.1f B/item; same-x rejects %llu\n", (unsigned long long)stt.table_items, stt.table_bytes / 1073741824.0,
           (double)stt.table_bytes / (double)(stt.table_items ? stt.table_items : 1), (unsigned long long)stt.same_herd);
    printf("consumer load: min %llu max %llu points\n", (unsigned long long)lmin, (unsigned long long)lmax);
    { // the slowest hand-overs, and when they happened
        std::vector<std::pair<double, int>> all;
        for (int f = 0; f < feeders; f++)
            for (int l = 0; l < launches; l++) all.emplace_back(per_launch[f][l], f * 100000 + l);
        std::sort(all.begin(), all.end(), [](auto &a, auto &b) { return a.first > b.first; });
        printf("slowest hand-overs (ms @ feeder:launch):");
        for (size_t i = 0; i < 6 && i < all.size(); i++) printf(" %.2f@%d:%d", all[i].first * 1e3, all[i].second / 100000, all[i].second % 100000);
        printf("; median %.2f ms\n", all[all.size() / 2].first * 1e3);
    }
    kngs_host_stats_t hs;
    if (kngs_host_stats(s, &hs) == 0)
        printf("consumers confined to %u NUMA node(s); the process may use %.1f CPUs\n", hs.numa_nodes, hs.effective_cpus);
    if (kngs_host_stats(s, &hs) == 0)
        printf("host stats: consumers busy %.0f %% mean / %.0f %% max of the run; most points waiting in one queue %llu; slowest hand-over of a launch %.2f ms\n",
               100 * hs.consumer_busy_mean, 100 * hs.consumer_busy_max, (unsigned long long)hs.queue_high_points, hs.ingest_ms_max);
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    printf("process: %.2f s user + %.2f s system CPU over %.2f s wall, %ld minor page faults\n", ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6,
           ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6, t_all, ru.ru_minflt);
    kngs_stop(s);
    if (kngs_host_stats(s, &hs) == 0)
        printf("consumer threads, summed: %.2f s inside batches, of which %.2f s on a CPU and %.2f s runnable but waiting for one; %llu voluntary / %llu involuntary context switches\n",
               hs.consumer_busy_s, hs.consumer_cpu_s, hs.consumer_runq_s, (unsigned long long)hs.consumer_nvcsw, (unsigned long long)hs.consumer_nivcsw);
    kngs_destroy(s);
    return 0;
}
