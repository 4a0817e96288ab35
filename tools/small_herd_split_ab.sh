#!/bin/bash
# A/B in ONE GPU session: the inversion of small herds ("share" 4) on one wave (round 5: libkangaroo_hip_onewave.so, built with
# tools/build_variant.sh onewave -- -DKNG_INV_ONE_WAVE) against two waves, lead + follow (round 6, the default library).
# usage (GPU box): bash tools/small_herd_split_ab.sh [rounds=3]
R=${1:-3}
for i in $(seq $R); do
  for v in onewave default; do
    if [ $v = default ]; then unset KNG_LIB_PATH; else export KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$v.so; fi
    echo "== $v"; timeout 600 python tools/small_herd_ab.py 1 | grep "share 4"
  done
done
