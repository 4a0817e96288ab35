#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
echo "== full gpu suite"; timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/r03f_pytest_gpu.txt
echo "== A/B: asm0 | asm1 unroll 1 | asm1 unroll 2 (default)"
for i in 1 2 3; do
  timeout 300 python tools/sweep.py --launches 10 --groups 64 --blocks 256 --asm 0,1 | grep "^asm" | sed 's/^asm 1/asm1-u2/'
  KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_unroll1.so timeout 300 python tools/sweep.py --launches 10 --groups 64 --blocks 256 --asm 1 | grep "^asm" | sed 's/^asm 1/asm1-u1/'
done 2>&1 | tee $OUT/r03f_ab_unroll.txt
echo "== bench"; timeout 600 python bench.py 2> $OUT/r03f_bench.err | tee $OUT/r03f_bench.json; tail -2 $OUT/r03f_bench.err
