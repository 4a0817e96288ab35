#!/bin/bash
# flags in VGPRs (running max / min) instead of scalar ORs; LDS latency hidden by the scheduler.  build | run
cd $(dirname $0)/..
if [ "$1" == "build" ]; then
  bash tools/build_variant.sh base
  bash tools/build_variant.sh fv KASM_FLAGS=valu
  bash tools/build_variant.sh fv_l64 KASM_FLAGS=valu KASM_LDSLAT=64
  bash tools/build_variant.sh l64 KASM_LDSLAT=64
  bash tools/build_variant.sh l128 KASM_LDSLAT=128
  bash tools/build_variant.sh fv_l128 KASM_FLAGS=valu KASM_LDSLAT=128
  exit 0
fi
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== parity of the flags-in-VGPRs build (subset)"
KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_fv_l64.so timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "every_walk_kernel or bench_config_total or low_word or fold_rare or shared_inversion or reference_vectors" 2>&1 | tail -3
echo "== A/B"
for i in 1 2 3; do for v in base fv l64 fv_l64 l128 fv_l128; do
  echo -n "$v: "; KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$v.so timeout 300 python tools/sweep.py --launches 16 --groups 64 --blocks 256 | grep "^asm" | grep -oE "kernel +[0-9.]+ ms +[0-9.]+ MK/s"
done; done 2>&1 | tee $OUT/r03_ab_flags_valu.txt
