#!/bin/bash
# the host DP path on the GPU box's host (no GPU work): topology and CPU quota, single-thread table cost, consumer-count sweep,
# the 8-GPU cadences.  usage: bash tools/r4_dp_host.sh [tag]   -> gpurun_out/r04_dp_host_<tag>.txt
TAG=${1:-after}
OUT=gpurun_out/r04_dp_host_$TAG.txt; mkdir -p gpurun_out
B="./tools/dp_ingest_bench --feeders 8"
F="consumer threads|feeder side|paced|end to end|host stats|process|NUMA|slowest|summed|table:"
{
echo "== host"; lscpu | grep -E "Model name|Socket|Core|Thread|NUMA|^CPU\(s\)"; echo "cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; cat /sys/kernel/mm/transparent_hugepage/enabled
echo "== one thread, table only"; ./tools/dp_table_bench 40000000 | cut -c1-120
for C in 8 12 14 16 32; do echo "== unpaced, $C consumers"; $B --launches 60 --consumers $C | grep -E "$F"; done
for MS in 21 17 12; do echo "== paced $MS ms ($(python3 -c "print(round(8*262144/$MS/1e3,1))") M points/s offered), default consumers"; $B --launches 80 --launch-ms $MS | grep -E "$F"; done
echo "== 400 launches per feeder, unpaced, default consumers (839 M points)"; $B --launches 400 | grep -E "$F"
} > $OUT 2>&1
cat $OUT
