/*
 * kng_solver.h -- multi-GPU host pipeline around the MI355X jump engine (libkangaroo_host.so).
 *
 * SURVEY 8(f) row 1 (+ rows 3, 4): what Kangaroo::SolveKeyGPU (Kangaroo.cpp:510-644) and its helpers
 * AddToTable/CollisionCheck/CheckKey (:233-329) do around class GPUEngine, rebuilt so that the host never
 * limits the engines:
 *   - one host thread per GPU drives kng_launch / kng_wait / kng_drain and starts launch k+1 BEFORE it
 *     touches the distinguished points of launch k (the reference inserts them under one global mutex
 *     between two launches, :594-612);
 *   - DPs are converted to hash-table entries on the GPU thread and handed, in batches, to consumer
 *     threads that each own a fixed subset of the 2^18 buckets (kng_dptable.h) -- no global lock;
 *   - a kangaroo that fell into the trail of one of its own herd (DUPLICATE / same-type collision,
 *     :599-606) is replaced with kng_set_kangaroo, stream-ordered, without stalling the GPU;
 *   - a tame/wild collision is resolved with the reference's four sign combinations (:233-268);
 *   - work files (kng_workfile.h) are saved at a launch boundary and restored, herd included, in the
 *     reference's format.
 * Same mathematics as the reference: jump table (seed 0x600DCAFE), DP size suggestion, wild offset
 * rangeWidth/2, keyToSearch = key - rangeStart*G (Kangaroo.cpp:835-905).  Not rebuilt: the CLI, the
 * client/server mode, CPU worker threads, merging tools.
 *
 * There is no CPU fallback: kngs_start fails when an engine cannot be created.
 */
#ifndef KNG_SOLVER_H
#define KNG_SOLVER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KNGS_MAX_GPUS 16

typedef struct kngs_config {
    uint64_t range_start[4], range_end[4]; /* inclusive range of the private key */
    uint64_t key_x[4], key_y[4];           /* public key to solve */
    int32_t dp;                            /* distinguished-point bits; -1 = suggested (Kangaroo.cpp:980-993) */
    int32_t n_gpus;
    int32_t gpu_ids[KNGS_MAX_GPUS];
    int32_t grid_x, grid_y; /* <=0: kng_default_grid of each device */
    uint32_t max_found;     /* DP slots per launch; 0 = max(131072, 2 x expected) */
    int32_t consumers;      /* hash-table threads; 0 = automatic */
    uint64_t seed;          /* herd seed; 0 = draw one (the reference seeds from the clock, main.cpp:177). Fix it only
                               for tests and benchmarks: equal seeds rebuild equal herds, whose walks retrace trails
                               already in a restored table */
    uint64_t max_launches;  /* per GPU, 0 = until solved or stopped */
    uint32_t warmup_launches; /* benchmarks: launches run and discarded by kngs_prepare, outside any timing */
    uint32_t flags;           /* KNGS_FLAG_* */
} kngs_config;
#define KNGS_FLAG_NO_PIN 1u /* do not confine the table threads to NUMA nodes (default: consumer i of n -> node i*nodes/n) */

typedef struct kngs_stats {
    uint64_t jumps;            /* incl. the count restored from a work file */
    uint64_t launches;         /* summed over GPUs, this run */
    uint64_t dps;              /* distinguished points received from the GPUs */
    uint64_t dps_lost;         /* dropped because max_found was exceeded */
    uint64_t same_herd;        /* kangaroos replaced: duplicate point or same-type collision */
    uint64_t wrong_collisions; /* tame/wild collisions that did not resolve (should stay 0) */
    uint64_t table_items;
    uint64_t kangaroos;        /* total over GPUs */
    double seconds;            /* wall time since kngs_start (plus restored time) */
    double kernel_ms_avg;      /* mean walk-kernel duration over GPUs and launches */
    int32_t dp, range_power, solved, running;
    uint64_t seed;             /* the herd seed in use (drawn when the configuration said 0) */
    uint64_t herd_loaded;      /* kangaroos taken from the work file at start */
    uint64_t herd_created;     /* kangaroos created at start (none left in the file for them) */
    uint64_t table_bytes;      /* memory held by the table (0 while the consumers are running) */
    uint64_t warmup_jumps;     /* part of `jumps`: the discarded warm-up launches of kngs_prepare (their DPs never reached the table) */
    uint64_t audits, audited_kangaroos, audit_mismatches; /* kngs_audit, cumulative over this run */
} kngs_stats;

typedef struct kngs_solver kngs_solver;

int kngs_create(const kngs_config *cfg, kngs_solver **out);
void kngs_destroy(kngs_solver *s);
/* restore hash table, counters and (when the file has them) herds from a HEADW work file written by this
 * library or by the reference; range and key must match the configuration.  Call before kngs_start. */
int kngs_load(kngs_solver *s, const char *path);
/* create the engines and build or upload the herds (everything that is not the search itself); optional --
 * kngs_start does it when it has not been done.  A benchmark calls it first so that its clock only sees the search. */
int kngs_prepare(kngs_solver *s);
/* (kngs_prepare, then) start the GPU and consumer threads */
int kngs_start(kngs_solver *s);
/* block until solved (returns 1), every GPU has done max_launches (2), or `seconds` elapsed (0); <0 error */
int kngs_wait(kngs_solver *s, double seconds);
/* ask the threads to stop after their current launch and join them */
int kngs_stop(kngs_solver *s);
/* the private key, valid once solved (kngs_wait returned 1); returns 0, or -1 when not solved */
int kngs_result(const kngs_solver *s, uint64_t priv[4]);
int kngs_get_stats(const kngs_solver *s, kngs_stats *st);
/* save a HEADW work file at the next launch boundary (GPUs pause, resume afterwards); with_kangaroos != 0
 * appends every herd (96 B per kangaroo, GPU order) like -ws.  Works while running and after kngs_stop. */
int kngs_save(kngs_solver *s, const char *path, int with_kangaroos);
/* Whole-run audit on the device (new; the reference's nearest tool is -wcheck on a saved file, Check.cpp:141-411, and the
 * final key check, Kangaroo.cpp:196-206): every kangaroo of every herd is re-derived from its distance as d*G resp.
 * K + d*G and compared in x and y (kng_audit_herd); with_table != 0 does the same for every entry of the table (what an
 * entry keeps of x: 128 bits + the 18 bucket bits).  A walk error is permanent for its kangaroo, so a clean audit after
 * J jumps certifies all J.  Works while running (the GPUs pause at a launch boundary, as for a save) and after the run
 * has ended or was stopped.  Returns 0 when the audit RAN (look at the mismatch counts), <0 on error. */
typedef struct kngs_audit_result {
    uint64_t kangaroos, kangaroo_mismatches; /* herd half: all GPUs */
    uint64_t table_points, table_mismatches; /* table half */
    double herd_ms, table_ms;                /* device kernel time */
    double seconds;                          /* wall time of the call, pause included */
    uint32_t n_first_bad, reserved;
    uint64_t first_bad[8];                   /* gpu << 56 | kIdx of the first mismatching kangaroos */
} kngs_audit_result;
int kngs_audit(kngs_solver *s, int with_table, kngs_audit_result *out);
/* Kangaroo::CollisionCheck + CheckKey (Kangaroo.cpp:233-329) as a pure function of the configuration: given the
 * true distances (mod n) of a tame and a wild kangaroo standing on the same point, try the reference's four sign
 * combinations and return the private key (1) or 0 when none reproduces the public key.  No GPU needed. */
int kngs_collision_key(const kngs_solver *s, const uint64_t tame_d[4], const uint64_t wild_d[4], uint64_t priv[4]);
/* per-GPU figures of this run: launches done, summed walk-kernel time (HIP events), herd size */
int kngs_gpu_stats(const kngs_solver *s, int gpu, uint64_t *launches, double *kernel_ms_sum, uint64_t *kangaroos);
/* an option of one GPU's engine as kng_get_option reads it ("group", "lanes", "share", "dsplit", "asm", ...): lets a caller
 * name the walk kernel that ran instead of assuming the defaults.  Valid after kngs_prepare / kngs_start. */
int kngs_gpu_option(const kngs_solver *s, int gpu, const char *key, int64_t *value);
/* Is the host keeping up?  (bench.py --gpus N prints these so that an N-GPU line diagnoses itself.)
 *   host_ms_*       a GPU thread's own work per launch -- next launch, drain, ingest, replacements -- i.e. everything
 *                   between two kng_wait calls; the GPU never idles while this stays below the kernel time
 *   late_launches   launches for which it did not (the kernel had finished before its host thread came back for it)
 *   ingest_ms_max   the slowest hand-over of one launch's points to the consumers
 *   queue_high_points  most points ever waiting in one consumer's queue
 *   consumer_busy_* fraction of the run a consumer thread spent inserting (1.0 = saturated) */
typedef struct kngs_host_stats_t {
    double host_ms_max, host_ms_mean, ingest_ms_max;
    uint64_t late_launches, queue_high_points;
    double consumer_busy_max, consumer_busy_mean, run_seconds;
    uint32_t consumers, numa_nodes; /* numa_nodes: nodes the consumers were confined to (0/1 = not pinned) */
    /* the kernel's account of the consumer threads, summed, available once they have ended (kngs_stop): seconds on a CPU,
     * seconds runnable but waiting for one, seconds inside batches (busy), voluntary / involuntary context switches */
    double consumer_cpu_s, consumer_runq_s, consumer_busy_s;
    uint64_t consumer_nvcsw, consumer_nivcsw;
    double effective_cpus; /* CPUs the process may use: hardware threads cut by affinity and the cgroup quota (sizes the default consumer count) */
    uint64_t pin_failures; /* threads of this process that could not be confined to the CPUs planned for them (they ran unconfined) */
} kngs_host_stats_t;
int kngs_host_stats(const kngs_solver *s, kngs_host_stats_t *out);
/* points each consumer thread has taken off its queue; returns the number of consumers */
int kngs_consumer_load(const kngs_solver *s, uint64_t *handled, int cap);

/* ---- the host path without engines: measurements and tests of the DP ingest (no GPU needed) ----
 * kngs_start_ingest starts only the consumer threads; `feeders` callers (one thread each, distinct feeder ids)
 * then hand over engine records exactly as a GPU thread would after kng_drain_view.  kngs_drained waits until
 * every queued point is in the table (1) or `seconds` passed (0).  Stop with kngs_stop. */
struct kng_dp_record;
int kngs_start_ingest(kngs_solver *s, int feeders);
int kngs_ingest(kngs_solver *s, int feeder, const struct kng_dp_record *records, uint32_t n);
int kngs_drained(kngs_solver *s, double seconds);
/* the solver's table, read-only (tests; exact only while no point is in flight) */
struct kngt_table;
const struct kngt_table *kngs_table(const kngs_solver *s);
/* ---- thread placement, inspectable without a GPU.  Node CPU sets come from <root>/devices/system/node/nodeK/cpulist cut by
 * the caller's affinity mask, indexed by the REAL node id (sparse ids and nodes without usable CPUs give empty sets);
 * KNG_SYSFS_ROOT replaces "/sys".  kngs_plan_placement writes one "node K: <cpulist or ->" line per node id and one
 * "consumer C: node K cpus <list>|unconfined" line per table thread (per_core = what KNGS_PIN=core selects; the default
 * confines a table thread to its node, and not at all on a one-node machine) and returns the number of node ids.
 * kngs_gpu_thread_cpus: where the host thread (and the pinned rings) of a GPU on `device_numa_node` would be confined
 * (returns 1 + the list, or 0 + "unconfined").  kngs_try_pin: 1 when the calling thread could be confined to `cpulist`
 * (its mask is restored), 0 when the kernel refused -- counted in kngs_pin_failures / kngs_host_stats_t.pin_failures. */
int kngs_plan_placement(int n_consumers, int per_core, unsigned salt, char *out, size_t cap);
int kngs_gpu_thread_cpus(int device_numa_node, char *out, size_t cap);
int kngs_try_pin(const char *cpulist);
uint64_t kngs_pin_failures(void);
const char *kngs_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
