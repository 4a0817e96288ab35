#!/bin/bash
# round 3, first GPU trip: parity of the scheduled asm loop, then asm loop vs compiler loop in ONE session
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== parity (asm loop is the default)"; timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -6 | tee $OUT/r03a_pytest_parity.txt
echo "== A/B: compiler-scheduled loop (asm 0) vs scheduled asm loop (asm 1), 2^23 kangaroos, group 64, share 2"
for i in 1 2 3; do
  timeout 300 python tools/sweep.py --launches 8 --groups 64 --blocks 256 --asm 0,1 | grep asm
done 2>&1 | tee $OUT/r03a_ab_asm.txt
echo "== A/B both distance words"
timeout 300 python tools/sweep.py --launches 8 --groups 64 --blocks 256 --asm 0,1 --jd-bits 56 | grep asm | tee -a $OUT/r03a_ab_asm.txt
if [ -f kangaroo_amd/lib/libkangaroo_hip_nomulasm.so ]; then
  echo "== compiler loop with the per-column multiplier of rounds 1-2 (KNG_USE_MULASM=0)"
  for i in 1 2; do KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_nomulasm.so timeout 300 python tools/sweep.py --launches 8 --groups 64 --blocks 256 --asm 0 | grep asm; done | tee $OUT/r03a_ab_nomulasm.txt
fi
echo "== bench"; timeout 600 python bench.py --no-pipeline --no-secondary 2> $OUT/r03a_bench.err | tee $OUT/r03a_bench.json; tail -3 $OUT/r03a_bench.err
