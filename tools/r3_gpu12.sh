#!/bin/bash
# the CU-wide inversion with complements prepared while wave 0 inverts (one multiplication behind the inversion)
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== parity (default lib = new tree)"
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "every_walk_kernel or bench_config_total or shared_inversion or ragged or exact_path or reference_vectors" 2>&1 | tail -3
echo "== A/B: tree of round 3's first form (elide) vs complements (tree2)"
for i in 1 2 3; do for v in elide tree2; do
  echo -n "$v: "; KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$v.so timeout 300 python tools/sweep.py --launches 16 --groups 64 --blocks 256 | grep "^asm" | grep -oE "kernel +[0-9.]+ ms +[0-9.]+ MK/s"
done; done 2>&1 | tee $OUT/r03_ab_tree.txt
