#!/bin/bash
# GPU round trip: parity (ours + the reference's own -gpu -check), bench line, rocprof summaries.
# usage: bash tools/gpu_round.sh <tag>
TAG=${1:-r01}
OUT=$PWD/gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== pytest -m gpu"; python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/${TAG}_pytest_gpu.txt
echo "== smoke"; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee $OUT/${TAG}_smoke.txt
if [ -x oracle/_ref/kangaroo_hip ]; then
  echo "== reference program on our engine: kangaroo -gpu -check (Check.cpp:467-621)"
  # dp is fixed at 8 and maxFound at 65536 inside Check (Check.cpp:418,492): the herd must stay below
  # 2^18 kangaroos or DPs are dropped and the harness itself derails
  timeout 900 ./oracle/_ref/kangaroo_hip -gpu -g 8,128 -check > $OUT/${TAG}_ref_gpu_check.txt 2>&1; echo "rc=$?" >> $OUT/${TAG}_ref_gpu_check.txt
  tail -12 $OUT/${TAG}_ref_gpu_check.txt
fi
echo "== bench"; python bench.py 2> $OUT/${TAG}_bench.err | tee $OUT/${TAG}_bench.json; tail -3 $OUT/${TAG}_bench.err
echo "== rocprofv3 kernel trace"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o kt -- python $OLDPWD/bench.py --no-cpu-baseline --no-pipeline > $OUT/${TAG}_prof_bench.json 2> $OUT/${TAG}_prof.err)
find $OUT/${TAG}_prof -name "*kernel_stats*" | head -3
for f in $(find $OUT/${TAG}_prof -name "*kernel_stats.csv"); do cp $f $OUT/${TAG}_kernel_stats.csv; done
cat $OUT/${TAG}_kernel_stats.csv 2>/dev/null | head -8
echo "== rocprofv3 PMC passes (counters only)"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $C --output-format csv -d $OUT/${TAG}_pmc_$C -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-pipeline > /dev/null 2> $OUT/${TAG}_pmc_$C.err)
  f=$(find $OUT/${TAG}_pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f $C | tee $OUT/${TAG}_pmc_$C.txt
done
ls -R $OUT/${TAG}_prof | head -20
rm -rf $OUT/${TAG}_prof $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
