#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== parity incl. share 8"; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "every_walk_kernel or shared_inversion" 2>&1 | tail -3
echo "== A/B share 2 vs share 8 (one inversion per CU)"
for i in 1 2 3; do timeout 300 python tools/sweep.py --launches 10 --groups 64 --blocks 256 --asm 1 --shares 2,8 | grep "^asm"; done 2>&1 | tee $OUT/r03g_ab_share8.txt
timeout 300 python tools/sweep.py --launches 10 --groups 64 --blocks 256 --asm 1 --shares 2,8 --jd-bits 56 | grep "^asm" | tee -a $OUT/r03g_ab_share8.txt
timeout 300 python tools/sweep.py --launches 10 --groups 32 --blocks 256 --asm 1 --shares 2,8 | grep "^asm" | tee -a $OUT/r03g_ab_share8.txt
