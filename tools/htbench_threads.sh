#!/bin/bash
# The table side of the reference program with the link-time replacements, on THIS host, as the table grows (VERDICT r5 item 3:
# >= 120 M points/s with 16 threads, <= 110 ns per point and thread to 3e8 entries):
#   poolbench      producers -> kng_ingest.h pool of owner-partitioned table threads -> HashTable_kng.o (what SolveKeyGPU_kng.cpp runs)
#   htbench ingestp  kng_ht_ingest from T threads, each feeding its own 1/T of the buckets (the table threads alone, no routing)
#   htbench ingest   kng_ht_ingest from T threads, every thread all over the table (round 5's arrangement)
# usage: tools/htbench_threads.sh [points=320000000]
PTS=${1:-320000000}
ROOT=${GRAFT_REPO_ROOT:-$PWD}
B=$ROOT/oracle/_ref
echo "# host: $(nproc) hardware threads, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), $(grep MemTotal /proc/meminfo)"
echo "## poolbench: 8 table threads + 8 producers (eight GPU threads inside a 16-CPU quota)"; $B/poolbench $PTS $((PTS / 4)) 8 8
echo "## poolbench: 12 table threads + 4 producers";                                        $B/poolbench $PTS $((PTS / 4)) 12 4
echo "## poolbench: 16 table threads + 8 producers (24 threads on the quota)";              $B/poolbench $PTS $((PTS / 4)) 16 8
echo "## htbench ingestp 16";  $B/htbench_kng ingestp $PTS $((PTS / 4)) 16
echo "## htbench ingestp 8";   $B/htbench_kng ingestp $((PTS / 2)) $((PTS / 8)) 8
echo "## htbench ingest 16";   $B/htbench_kng ingest $PTS $((PTS / 4)) 16
