#!/bin/bash
# instruction-cache, LDS-conflict and scalar-cache counters of the walk kernel (counters only)
TAG=${1:-r03}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
for SET in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
  (cd /tmp && rocprofv3 --pmc $SET --output-format csv -d $OUT/${TAG}_misc -o m -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline --no-secondary > /dev/null 2> $OUT/${TAG}_misc.err)
  f=$(find $OUT/${TAG}_misc -name "*counter_collection.csv" | head -1)
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
  else echo "set '$SET' failed: $(tail -2 $OUT/${TAG}_misc.err | tr '\n' ' ')"; fi
  rm -rf $OUT/${TAG}_misc
done
