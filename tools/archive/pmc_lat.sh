#!/bin/bash
# memory-latency counters of the walk kernel (counters only): average VMEM / LDS latency = LEVEL / INSTS
# usage: bash tools/pmc_lat.sh <tag>
TAG=$1
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
for SET in "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VMEM SQ_WAVE_CYCLES" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU"; do
  (cd /tmp && rocprofv3 --pmc $SET --output-format csv -d $OUT/${TAG}_lat -o lat -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > /dev/null 2> $OUT/${TAG}_lat.err)
  f=$(find $OUT/${TAG}_lat -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "walk" in row.get("Kernel_Name", ""):
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in acc.items():
    print(f"{c:28s} mean {sum(v)/len(v):.6g}  (n={len(v)})")
PY
  else tail -3 $OUT/${TAG}_lat.err; fi
  rm -rf $OUT/${TAG}_lat
done
