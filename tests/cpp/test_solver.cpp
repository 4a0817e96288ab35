// test_solver.cpp -- the host pipeline driven from C++ through its C ABI only (kng_solver.h, kng_workfile.h,
// kng_dptable.h): solves the reference's shipped 56-bit known-answer input (in.txt, answer README.md:331-357),
// saves a work file with the herd, reads it back and checks the counters.  Built and run by
// tests/test_gpu_solver.py::test_solver_from_cpp on the GPU box.
#include <cinttypes>
#include <cstdio>
#include <cstring>

#include "kng_dptable.h"
#include "kng_host.h"
#include "kng_solver.h"
#include "kng_workfile.h"

static int fail(const char *what, const char *detail) {
    std::printf("FAIL %s: %s\n", what, detail ? detail : "");
    return 1;
}

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/tmp/kng_cpp_solver.work";
    kngs_config cfg;
    std::memset(&cfg, 0, sizeof cfg);
    cfg.range_end[0] = 0xFFFFFFFFFFFFFFULL; // in.txt: [0, 2^56 - 1]
    const uint64_t answer[4] = {0x378ABDEC51BC5DULL, 0, 0, 0};
    if (kngh_pubkey(answer, cfg.key_x, cfg.key_y) != 0) return fail("kngh_pubkey", nullptr);
    if (cfg.key_x[3] != 0xE9F43F810784FF1EULL) return fail("public key", "does not match in.txt"); // 02E9F43F810784FF1E...
    cfg.dp = 8;
    cfg.n_gpus = 1;
    cfg.grid_x = 32;
    cfg.grid_y = 128;
    cfg.seed = 2024;

    kngs_solver *s = nullptr;
    if (kngs_create(&cfg, &s) != 0) return fail("kngs_create", kngs_last_error());
    if (kngs_start(s) != 0) return fail("kngs_start", kngs_last_error());
    const int rc = kngs_wait(s, 240.0);
    if (rc != 1) return fail("kngs_wait", rc < 0 ? kngs_last_error() : "not solved in time");
    uint64_t priv[4];
    if (kngs_result(s, priv) != 0 || std::memcmp(priv, answer, 32) != 0) return fail("kngs_result", "wrong key");
    if (kngs_stop(s) != 0) return fail("kngs_stop", kngs_last_error());
    kngs_stats st;
    kngs_get_stats(s, &st);
    // whole-run audit through the C ABI: every kangaroo and every table entry re-derived from its distance on the device
    kngs_audit_result au;
    if (kngs_audit(s, 1, &au) != 0) return fail("kngs_audit", kngs_last_error());
    if (au.kangaroos != st.kangaroos || au.kangaroo_mismatches != 0 || au.table_points != st.table_items || au.table_mismatches != 0)
        return fail("kngs_audit", "mismatches, or not everything was audited");
    kngs_host_stats_t hs;
    if (kngs_host_stats(s, &hs) != 0 || hs.consumers < 1 || hs.effective_cpus < 1.0) return fail("kngs_host_stats", kngs_last_error());
    if (kngs_save(s, path, 1) != 0) return fail("kngs_save", kngs_last_error());
    kngs_destroy(s);

    kngw_header h;
    uint64_t n = 0;
    kngt_table *t = kngt_create();
    kngw_file *f = kngw_open(path, &h, t, &n);
    if (!f) return fail("kngw_open", kngw_last_error());
    kngw_close(f);
    if (h.magic != KNGW_HEADW || h.dp_size != 8 || h.total_count != st.jumps || n != st.kangaroos ||
        kngt_count(t) != st.table_items || std::memcmp(h.key_x, cfg.key_x, 32) != 0)
        return fail("work file", "header or counters differ from the solver's statistics");
    kngt_destroy(t);
    std::printf("CPP solver ok: key 0x%" PRIX64 " after %" PRIu64 " launches, %" PRIu64 " DPs, %" PRIu64 " kangaroos saved; audit: %" PRIu64
                " kangaroos + %" PRIu64 " table entries clean\n", priv[0], st.launches, st.dps, n, au.kangaroos, au.table_points);
    return 0;
}
