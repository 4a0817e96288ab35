#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
{
echo "== A/B micro-variants (sweep, 2^23 kangaroos, group 64, share 2)"
for i in 1 2 3; do python tools/ablate_run.py --launches 200 base s_plain setprio lds_b64; done
echo "== LDS bank conflicts: base vs one ds_read_b64 per word"
for V in base lds_b64; do
  (cd /tmp && KNG_LIB_PATH=$OLDPWD/build/abl/$V/libkangaroo_hip.so rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d $OUT/lds_$V -o pmc -- python $OLDPWD/tools/sweep.py --launches 2 --groups 64 --blocks 256 > /dev/null 2> $OUT/lds_$V.err)
  f=$(find $OUT/lds_$V -name "*counter_collection.csv" | head -1)
  echo "-- $V"; for C in SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS; do python tools/pmc_summary.py $f $C | grep walk; done
  rm -rf $OUT/lds_$V
done
} 2>&1 | tee $OUT/r02d_micro.txt
