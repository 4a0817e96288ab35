#!/bin/bash
# Sensitivity map of the round-3 kernel (scheduled asm loop, one inversion per CU): builds with ONE component removed
# (wrong results on purpose -- never product code), each run for 240 launches while package power and clock are sampled.
#   build (CPU box):  bash tools/r3_sensitivity.sh build
#   run   (GPU box):  bash tools/r3_sensitivity.sh run
cd $(dirname $0)/..
if [ "$1" == "build" ]; then
  bash tools/build_variant.sh abl_base
  bash tools/build_variant.sh abl_noinv -- -DKNG_ABL_NOINV
  bash tools/build_variant.sh abl_nos KASM_ABL=nos
  bash tools/build_variant.sh abl_nostore KASM_ABL=nostore
  bash tools/build_variant.sh abl_sub2proxy KASM_ABL=nosA,plusmulA
  bash tools/build_variant.sh abl_plusmul KASM_ABL=plusmulA
  exit 0
fi
OUT=$PWD/gpurun_out; mkdir -p $OUT
L=${2:-240}
CMDS=()
for v in base noinv nos nostore sub2proxy plusmul base; do
  CMDS+=(--cmd "env KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_abl_$v.so python tools/sweep.py --launches $L --groups 64 --blocks 256")
done
python tools/ablate_run.py "${CMDS[@]}" 2>&1 | sed -E 's/^env KNG_LIB_PATH=[^ ]*libkangaroo_hip_abl_([a-z0-9]+)\.so[^:]*:/\1:/' | cut -c1-330 | tee $OUT/r03_sensitivity.txt
