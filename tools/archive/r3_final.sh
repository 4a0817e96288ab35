#!/bin/bash
# round 3, closing pass on the GPU box: full gpu_round (tests, smoke, reference -check, bench, rocprof, PMC), SQ counters of the
# final kernel, the bench key solved end to end through the host pipeline (sustained rate), and the solver soak with periodic saves
TAG=${1:-r03_final}
OUT=$PWD/gpurun_out; mkdir -p $OUT
bash tools/gpu_round.sh $TAG > $OUT/${TAG}_round.log 2>&1
grep -E "passed|failed|smoke ok|CPU/GPU|walk kernel kng" $OUT/${TAG}_round.log | cut -c1-300
grep -A2 '"Name","Calls"' $OUT/${TAG}_round.log | head -3
if [ "$2" != "short" ]; then
  echo "== SQ counters"; bash tools/pmc_sq.sh $TAG 2>&1 | grep -E "mean|kernel ms" | tee $OUT/${TAG}_pmc_sq.txt
  echo "== 80-bit key end to end"; timeout 400 python tools/solve_demo.py --bits 80 --max-seconds 300 2>&1 | tail -8 | tee $OUT/${TAG}_solve_80bit.txt
fi
echo "== soak"; timeout 200 python tools/solver_soak.py 2>&1 | tail -2 | tee $OUT/${TAG}_solver_soak.txt
