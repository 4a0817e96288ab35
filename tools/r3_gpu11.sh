#!/bin/bash
# first-carry elision under the VGPR flag superset (a0 / b7 near 2^32): -65 slow VALU per jump
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== parity of the elision build (subset)"
KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_elide.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "every_walk_kernel or bench_config_total or low_word or exact_path or shared_inversion or reference_vectors" 2>&1 | tail -3
echo "== A/B"
for i in 1 2 3; do for v in base elide; do
  echo -n "$v: "; KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$v.so timeout 300 python tools/sweep.py --launches 16 --groups 64 --blocks 256 | grep "^asm" | grep -oE "kernel +[0-9.]+ ms +[0-9.]+ MK/s"
done; done 2>&1 | tee $OUT/r03_ab_elide.txt
echo "== power / clock"
python tools/ablate_run.py --cmd "env KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_base.so python tools/sweep.py --launches 160 --groups 64 --blocks 256" --cmd "env KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_elide.so python tools/sweep.py --launches 160 --groups 64 --blocks 256" 2>&1 | sed -E 's/^env KNG_LIB_PATH=[^ ]*libkangaroo_hip_([a-z0-9_]+)\.so[^:]*:/\1:/' | cut -c1-300 | tee -a $OUT/r03_ab_elide.txt
