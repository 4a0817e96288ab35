#!/bin/bash
# A/B two builds of the engine in ONE GPU session (boxes differ by a few percent): alternate runs
# usage: bash tools/ab.sh <libA> <libB> [rounds]
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    echo -n "$(basename $L): "; KNG_LIB_PATH=$PWD/$L python tools/sweep.py --launches 6 --groups 64 --blocks 256 | tail -1 | grep -oE "kernel +[0-9.]+ ms +[0-9.]+ MK/s"
  done
done
