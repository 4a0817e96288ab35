#!/bin/bash
# A/B builds of the engine in ONE GPU session (boxes differ by a few percent): alternate runs.
#   build the variants on the CPU box first:  bash tools/build_variant.sh <name> [KASM_*=..] [-- -D...]
#   usage (GPU box): bash tools/ab.sh <rounds> <name> [<name> ...]      e.g.  bash tools/ab.sh 3 base elide
# (round 3's A/B files under profiles/r03_ab_*.txt were produced this way; generator options: tools/gen_walk_asm.py)
R=$1; shift
for i in $(seq $R); do
  for v in "$@"; do
    echo -n "$v: "; KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$v.so timeout 300 python tools/sweep.py --launches 16 --groups 64 --blocks 256 | grep "^asm" | grep -oE "kernel +[0-9.]+ ms +[0-9.]+ MK/s"
  done
done
