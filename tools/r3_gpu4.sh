#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== parity of both variants (quick subset)"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "every_walk_kernel or bench_config_total or shared_inversion or low_word or fold_rare" 2>&1 | tail -3
KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_elide.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "every_walk_kernel or bench_config_total or low_word" 2>&1 | tail -3
echo "== A/B: asm0 | asm1 base | asm1 elide"
for i in 1 2 3; do
  timeout 300 python tools/sweep.py --launches 10 --groups 64 --blocks 256 --asm 0,1 | grep asm
  KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_elide.so timeout 300 python tools/sweep.py --launches 10 --groups 64 --blocks 256 --asm 1 | grep asm | sed 's/^asm 1/elide/'
done 2>&1 | tee $OUT/r03d_ab.txt
