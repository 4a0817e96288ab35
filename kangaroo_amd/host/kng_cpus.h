// kng_cpus.h -- how many CPUs this process may really use: hardware threads, cut by its affinity mask and by the cgroup's CPU
// quota.  (The GPU boxes of this project show 256 hardware threads under a quota of 16 CPUs: threads beyond the quota do not run
// in parallel, they take turns -- 32 table threads were SLOWER than 16 there, profiles/r04_dp_probe3.txt.)  Used to size the
// table threads of kng_solver.cpp and of SolveKeyGPU_kng.cpp.
#ifndef KNG_CPUS_H
#define KNG_CPUS_H

#include <sched.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>

inline double kng_effective_cpus() {
    double n = (double)std::thread::hardware_concurrency();
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0 && CPU_COUNT(&set) < n) n = CPU_COUNT(&set);
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) { // cgroup v2: "<quota|max> <period>"
        char q[32];
        double period = 0;
        if (fscanf(f, "%31s %lf", q, &period) == 2 && std::strcmp(q, "max") != 0 && period > 0) {
            const double c = atof(q) / period;
            if (c > 0 && c < n) n = c;
        }
        fclose(f);
    } else { // cgroup v1
        double quota = -1, period = 0;
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (fscanf(g, "%lf", &quota) != 1) quota = -1;
            fclose(g);
        }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (fscanf(g, "%lf", &period) != 1) period = 0;
            fclose(g);
        }
        if (quota > 0 && period > 0 && quota / period < n) n = quota / period;
    }
    return n < 1 ? 1 : n;
}
#endif
