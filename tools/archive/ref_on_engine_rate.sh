#!/bin/bash
# end-to-end rate of the UNMODIFIED reference program (SolveKeyGPU + HashTable) on our engine,
# BASELINE configs[2]: 80-bit range, default grid, auto DP
OUT=$PWD/gpurun_out; mkdir -p $OUT
cd /tmp
printf "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000\nB60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF\n03BB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4\n" > in80.txt
timeout ${1:-170} stdbuf -o0 $GRAFT_REPO_ROOT/oracle/_ref/kangaroo_hip -t 0 -gpu ${KNG_EXTRA} in80.txt 2>&1 | tr "\r" "\n" > $OUT/ref_on_engine_80bit.txt
grep -v "^\[" $OUT/ref_on_engine_80bit.txt | head -20
grep "^\[" $OUT/ref_on_engine_80bit.txt | tail -8
