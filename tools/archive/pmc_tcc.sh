#!/bin/bash
OUT=$PWD/gpurun_out; export TMPDIR=/tmp
for V in s_nt base; do
  (cd /tmp && KNG_LIB_PATH=$OLDPWD/build/abl/$V/libkangaroo_hip.so rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/tcc_$V -o pmc -- python $OLDPWD/tools/sweep.py --launches 2 --groups 64 --blocks 256 > /dev/null 2> $OUT/tcc_$V.err)
  f=$(find $OUT/tcc_$V -name "*counter_collection.csv" | head -1)
  echo "-- $V (s_nt = product planes with the non-temporal hint, base = plain)"; for C in TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum; do python tools/pmc_summary.py $f $C | grep walk; done
  rm -rf $OUT/tcc_$V
done
