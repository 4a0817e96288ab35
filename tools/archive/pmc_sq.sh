#!/bin/bash
# SQ counter pass for the walk kernel (counters only, no tracing domains)
# usage: bash tools/pmc_sq.sh <tag> [group]
TAG=$1; G=${2:-64}
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cat > /tmp/run_walk.py <<PY
import sys; sys.path.insert(0, "$PWD")
import numpy as np, kangaroo_amd as k
gx, gy = 512, 128; n = gx*gy*128
rng = np.random.default_rng(1)
x = rng.integers(0, 1<<64, size=(n,4), dtype=np.uint64); y = rng.integers(0, 1<<64, size=(n,4), dtype=np.uint64)
x[:,3] >>= np.uint64(1); y[:,3] >>= np.uint64(1)
d = rng.integers(0, 1<<62, size=(n,2), dtype=np.uint64)
jd = rng.integers(0, 1<<40, size=(32,2), dtype=np.uint64); jd[:,1] = 0
jx = rng.integers(0, 1<<63, size=(32,4), dtype=np.uint64); jy = rng.integers(0, 1<<63, size=(32,4), dtype=np.uint64)
eng = k.GPUEngine(gx, gy, 0, 1<<17, group=$G)
eng.SetParams(0xFFFC000000000000, jd, jx, jy); eng.SetKangaroos(x, y, d)
for _ in range(2):
    eng.callKernel(); eng.wait(); eng.drain(raw=True)
print("kernel ms", eng.last_kernel_ms())
PY
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  (cd /tmp && rocprofv3 --pmc $SET --output-format csv -d $OUT/${TAG}_sq -o sq -- python /tmp/run_walk.py > /dev/null 2> $OUT/${TAG}_sq.err)
  f=$(find $OUT/${TAG}_sq -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "walk" in row.get("Kernel_Name", ""):
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
for c, v in acc.items():
    print(f"{c:24s} mean {sum(v)/len(v):.6g}  (n={len(v)})")
PY
  fi
  rm -rf $OUT/${TAG}_sq
done
