#!/bin/bash
# round-2 session 4: full GPU suite, host path sweeps on the box's cores, rocprof of the both-words kernel
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
echo "== pytest -m gpu"; python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee $OUT/r02b_pytest_gpu.txt
echo "== host DP path, 8 feeders paced at one 262144-point launch per 25 ms (83.9 M points/s offered)"
for C in 16 32 48 64 96; do ./tools/dp_ingest_bench --feeders 8 --consumers $C --launches 60 --launch-ms 25 | sed -n 1,6p; done 2>&1 | tee $OUT/r02b_dp_ingest_sweep.txt
echo "== rocprofv3 kernel trace: both distance words streaming (109-bit class of ranges)"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r02b_prof -o kt -- python $OLDPWD/tools/sweep.py --launches 12 --groups 64 --blocks 256 --jd-bits 55 > $OUT/r02b_prof_sweep.txt 2> $OUT/r02b_prof.err)
for f in $(find $OUT/r02b_prof -name "*kernel_stats.csv"); do cp $f $OUT/r02b_kernel_stats_bothwords.csv; done
cat $OUT/r02b_kernel_stats_bothwords.csv | head -4; tail -1 $OUT/r02b_prof_sweep.txt
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && rocprofv3 --pmc $C --output-format csv -d $OUT/r02b_pmc_$C -o pmc -- python $OLDPWD/tools/sweep.py --launches 2 --groups 64 --blocks 256 --jd-bits 55 > /dev/null 2> $OUT/r02b_pmc_$C.err)
  f=$(find $OUT/r02b_pmc_$C -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_summary.py $f $C | grep walk | tee $OUT/r02b_pmc_bothwords_$C.txt
done
rm -rf $OUT/r02b_prof $OUT/r02b_pmc_FETCH_SIZE $OUT/r02b_pmc_WRITE_SIZE
