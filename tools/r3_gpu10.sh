#!/bin/bash
# prefetch at the top of the iteration, cache hints re-checked for the scheduled loop (all on top of the early prefetch)
OUT=$PWD/gpurun_out; mkdir -p $OUT
for i in 1 2 3; do for v in base early early_l256 nt_S nt_s nt_d nt_X nt_x; do
  echo -n "$v: "; KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$v.so timeout 300 python tools/sweep.py --launches 16 --groups 64 --blocks 256 | grep "^asm" | grep -oE "kernel +[0-9.]+ ms +[0-9.]+ MK/s"
done; done 2>&1 | tee $OUT/r03_ab_prefetch_hints.txt
