#!/bin/bash
# Where does the walk phase lose its VALU slots?  Builds without the loop's global traffic / LDS reads / flag ORs (wrong
# results on purpose) and with the scheduler hiding the LDS latency (exact).   build | run
cd $(dirname $0)/..
if [ "$1" == "build" ]; then
  bash tools/build_variant.sh abl_base
  bash tools/build_variant.sh abl_noglobal KASM_ABL=noglobal
  bash tools/build_variant.sh abl_nolds KASM_ABL=nolds
  bash tools/build_variant.sh abl_noflags KASM_ABL=noflags
  bash tools/build_variant.sh abl_valuonly KASM_ABL=noglobal,nolds,noflags
  bash tools/build_variant.sh ldslat16 KASM_LDSLAT=16
  bash tools/build_variant.sh ldslat32 KASM_LDSLAT=32
  bash tools/build_variant.sh ldslat64 KASM_LDSLAT=64
  exit 0
fi
OUT=$PWD/gpurun_out; mkdir -p $OUT
L=${2:-120}
CMDS=()
for v in abl_base abl_noglobal abl_nolds abl_noflags abl_valuonly abl_base ldslat16 ldslat32 ldslat64 abl_base ldslat16 ldslat32 ldslat64; do
  CMDS+=(--cmd "env KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$v.so python tools/sweep.py --launches $L --groups 64 --blocks 256")
done
python tools/ablate_run.py "${CMDS[@]}" 2>&1 | sed -E 's/^env KNG_LIB_PATH=[^ ]*libkangaroo_hip_([a-z0-9_]+)\.so[^:]*:/\1:/' | cut -c1-330 | tee $OUT/r03_stalls.txt
