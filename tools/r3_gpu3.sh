#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity"; timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -4 | tee $OUT/r03c_pytest_parity.txt
echo "== A/B asm 0 / 1"
for i in 1 2 3; do timeout 300 python tools/sweep.py --launches 10 --groups 64 --blocks 256 --asm 0,1 | grep asm; done 2>&1 | tee $OUT/r03c_ab_asm.txt
echo "== both words"; timeout 300 python tools/sweep.py --launches 10 --groups 64 --blocks 256 --asm 0,1 --jd-bits 56 | grep asm | tee -a $OUT/r03c_ab_asm.txt
echo "== bench"; timeout 600 python bench.py --no-secondary 2> $OUT/r03c_bench.err | tee $OUT/r03c_bench.json; tail -3 $OUT/r03c_bench.err
echo "== rocprof kernel trace"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r03c_prof -o kt -- python $OLDPWD/bench.py --no-cpu-baseline --no-pipeline --no-secondary > $OUT/r03c_prof_bench.json 2> $OUT/r03c_prof.err)
for f in $(find $OUT/r03c_prof -name "*kernel_stats.csv"); do cp $f $OUT/r03c_kernel_stats.csv; done; rm -rf $OUT/r03c_prof; head -5 $OUT/r03c_kernel_stats.csv
echo "== SQ counters"; bash tools/pmc_sq.sh r03c 2>&1 | tail -25
