#!/bin/bash
# Rate of the reference program with the link-time replacements (kangaroo_mi355x) against the DP size, one MI355X, default herd
# (2^23 kangaroos), 80-bit range, key outside the range: the table of INTEGRATION.md.  Each run: herd creation (~20 s) + SECS.
# usage: tools/ref_program_dp_table.sh [seconds=40] [dp sizes...=13 12 11 10]
SECS=${1:-40}; shift
DPS=${@:-13 12 11 10}
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
printf "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000\nB60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF\n03BB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4\n" > in80.txt
for d in $DPS; do
  f=$OUT/ref_program_dp${d}.txt
  KNG_STATS=10 timeout $((SECS+32)) stdbuf -o0 -e0 $ROOT/oracle/_ref/kangaroo_mi355x -t 0 -gpu -d $d in80.txt 2>&1 | tr "\r" "\n" > $f
  echo "== -d $d  (points per launch $((536870912 >> d)))"
  grep "SolveKeyGPU_kng" $f | tail -2
  grep "^\[" $f | tail -1
done
