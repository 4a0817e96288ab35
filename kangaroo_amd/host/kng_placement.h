// kng_placement.h -- where the host threads around the engines run on a machine with several NUMA nodes (eight
// GPUs hang off two sockets).  Shared by the repo's own solver (kng_solver.cpp) and, since round 6, by the table threads and GPU
// threads of the reference program's link-time replacements (kng_ingest.h, SolveKeyGPU_kng.cpp).  No libnuma: sysfs +
// sched_setaffinity; KNG_SYSFS_ROOT replaces "/sys" (tests/test_placement_cpu.py runs against made-up trees).
#ifndef KNG_PLACEMENT_H
#define KNG_PLACEMENT_H

#include <sched.h>

#include <atomic>
#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace kng_placement {

// ---- NUMA placement of the table threads.  The table grows by gigabytes per second and every insertion is a handful of
// dependent cache misses: a consumer that wanders between sockets (or whose arenas were first touched on the other one)
// pays the remote latency on each of them.  Consumer i of n is confined to the CPUs of the i*nodes/n-th node that has CPUs
// this process may use, so that the memory its arenas take from the OS (first touch) stays local for the whole run.
// No libnuma: sysfs + sched_setaffinity.  KNG_SYSFS_ROOT replaces "/sys" (tests run against a made-up tree).
inline std::string sysfs_root() {
    const char *e = getenv("KNG_SYSFS_ROOT");
    return e && *e ? std::string(e) : std::string("/sys");
}

// "0-63,128-191" -> the CPUs of the list that are also in `allowed`
inline int parse_cpulist(const char *text, const cpu_set_t &allowed, cpu_set_t *out) {
    CPU_ZERO(out);
    int n_set = 0;
    for (const char *p = text; *p;) {
        char *e;
        const long a = strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        p = e;
        if (*p == '-') {
            b = strtol(p + 1, &e, 10);
            p = e;
        }
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (c >= 0 && CPU_ISSET((int)c, &allowed) && !CPU_ISSET((int)c, out)) {
                CPU_SET((int)c, out);
                n_set++;
            }
        if (*p == ',') p++;
    }
    return n_set;
}

// CPU sets INDEXED BY THE REAL NODE ID: entry k is node k's CPUs cut by this process's affinity mask -- empty when the node
// does not exist, has no CPUs (memory-only nodes), or none of its CPUs is allowed (numactl, a cpuset).  Node ids may be
// sparse; the scan covers every id up to the highest directory present (a compacted list indexed with sysfs's
// numa_node put GPU threads and their pinned rings on the wrong node).
inline std::vector<cpu_set_t> numa_node_cpus() {
    std::vector<cpu_set_t> nodes;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof allowed, &allowed) != 0) return nodes;
    const std::string root = sysfs_root();
    int missing = 0;
    for (int node = 0; node < 1024 && missing < 64; node++) {
        char path[4200];
        snprintf(path, sizeof path, "%s/devices/system/node/node%d/cpulist", root.c_str(), node);
        FILE *f = fopen(path, "r");
        cpu_set_t set;
        CPU_ZERO(&set);
        if (!f) {
            missing++;
            continue;
        }
        missing = 0;
        char buf[4096];
        const bool ok = fgets(buf, sizeof buf, f) != nullptr;
        fclose(f);
        if (ok) (void)parse_cpulist(buf, allowed, &set);
        nodes.resize((size_t)node + 1);
        nodes[(size_t)node] = set;
    }
    // (entries the resize created for ids that have no directory are zero-initialised cpu_set_t: empty sets)
    return nodes;
}

inline bool node_usable(const std::vector<cpu_set_t> &nodes, int id) {
    return id >= 0 && (size_t)id < nodes.size() && CPU_COUNT(&nodes[(size_t)id]) > 0;
}

// one logical CPU per physical core of `set`: the lowest-numbered hardware thread of each core
inline std::vector<int> primary_cpus(const cpu_set_t &set) {
    std::vector<int> out;
    const std::string root = sysfs_root();
    for (int cpu = 0; cpu < CPU_SETSIZE; cpu++) {
        if (!CPU_ISSET(cpu, &set)) continue;
        char path[4200];
        snprintf(path, sizeof path, "%s/devices/system/cpu/cpu%d/topology/thread_siblings_list", root.c_str(), cpu);
        int first = cpu;
        if (FILE *f = fopen(path, "r")) {
            if (fscanf(f, "%d", &first) != 1) first = cpu;
            fclose(f);
        }
        if (first == cpu || !CPU_ISSET(first, &set)) out.push_back(cpu);
    }
    return out;
}

// Where each of `nc` table threads may run.  Default ("node"): the whole CPU set of its node -- the scheduler spreads the
// threads, they can move away from a CPU a GPU thread or another solver's table thread sits on, memory stays local.
// KNGS_PIN=core: ONE physical core per thread, spread evenly over the node's cores (hence over its L3 slices); on the 16-CPU
// GPU box that took eight flat-out feeders from 159 to 174 M points/s (profiles/r04_dp_host_*.txt) -- and it is opt-in because
// the choice is blind to everything else on the machine: a pinned thread cannot leave a CPU that something else keeps busy
// (every solver instance of a host, or two Solver objects in one process, used to pick the SAME cores; `salt`
// -- instance count + process id -- now rotates the choice, which helps between instances, not against strangers).
// A machine with one usable node is left alone in "node" mode (confining threads to "all CPUs" says nothing).
struct Placement {
    bool pin = false;
    int node = -1;
    cpu_set_t cpus;
};
inline std::vector<Placement> plan_consumers(const std::vector<cpu_set_t> &nodes, int nc, bool per_core, unsigned salt) {
    std::vector<Placement> plan((size_t)(nc > 0 ? nc : 0));
    for (Placement &p : plan) CPU_ZERO(&p.cpus);
    std::vector<int> usable;
    for (size_t k = 0; k < nodes.size(); k++)
        if (CPU_COUNT(&nodes[k]) > 0) usable.push_back((int)k);
    if (usable.empty() || nc <= 0) return plan;
    for (size_t u = 0; u < usable.size(); u++) {
        const cpu_set_t &set = nodes[(size_t)usable[u]];
        std::vector<int> mine; // consumers of this node
        for (int c = 0; c < nc; c++)
            if ((size_t)c * usable.size() / (size_t)nc == u) mine.push_back(c);
        const std::vector<int> cores = per_core ? primary_cpus(set) : std::vector<int>();
        for (size_t j = 0; j < mine.size(); j++) {
            Placement &p = plan[(size_t)mine[j]];
            p.node = usable[u];
            // one core each only where cores abound: twice as many as threads on a single-node machine (the GPU threads and
            // everybody else need somewhere to go), as many as threads on a node of several
            if (per_core && cores.size() >= mine.size() * (usable.size() > 1 ? 1 : 2)) {
                p.pin = true;
                const size_t stride = cores.size() / mine.size();
                CPU_SET(cores[(j * stride + salt % stride) % cores.size()], &p.cpus);
            } else if (usable.size() > 1) {
                p.pin = true;
                p.cpus = set;
            }
        }
    }
    return plan;
}

inline std::atomic<unsigned> g_solver_instances{0};
inline std::atomic<uint64_t> g_pin_failures{0};

// confine the calling thread; a refusal (the mask changed under us, a cpuset without these CPUs) is counted and reported
// once -- the thread then simply runs wherever it is allowed to
inline bool pin_this_thread(const cpu_set_t &cpus, const char *who) {
    if (CPU_COUNT(&cpus) > 0 && sched_setaffinity(0, sizeof cpus, &cpus) == 0) return true;
    if (g_pin_failures.fetch_add(1) == 0)
        fprintf(stderr, "kangaroo host: could not confine a %s thread to its CPUs (%s); it runs unconfined\n", who,
                CPU_COUNT(&cpus) > 0 ? strerror(errno) : "empty CPU set");
    return false;
}

inline std::string cpuset_text(const cpu_set_t &set) {
    std::string out;
    for (int c = 0; c < CPU_SETSIZE; c++) {
        if (!CPU_ISSET(c, &set)) continue;
        int e = c;
        while (e + 1 < CPU_SETSIZE && CPU_ISSET(e + 1, &set)) e++;
        char buf[48];
        if (e > c) snprintf(buf, sizeof buf, "%s%d-%d", out.empty() ? "" : ",", c, e);
        else snprintf(buf, sizeof buf, "%s%d", out.empty() ? "" : ",", c);
        out += buf;
        c = e;
    }
    return out;
}

} // namespace kng_placement
#endif
