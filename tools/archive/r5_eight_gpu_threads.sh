#!/bin/bash
# The reference program with the link-time replacements driving EIGHT engines (its own thread-per-GPU path, Kangaroo.cpp:1041-1047)
# on the one device available: -gpuId 0,0,0,0,0,0,0,0 -> 2^26 kangaroos, the program suggests DP 11 itself, eight GPU threads and
# their table threads (sized to the CPU quota) share one HashTable.  The kernels time-share the GPU, so the aggregate is one GPU's
# rate; what this shows is the host side of an 8-GPU run: nothing lost, no stall, no deadlock, one table.
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
printf "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000\nB60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF\n03BB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4\n" > in80.txt
f=$OUT/r05_ref_program_eight_gpu_threads.txt
KNG_STATS=20 timeout ${1:-75} stdbuf -o0 -e0 $ROOT/oracle/_ref/kangaroo_mi355x -t 0 -gpu -gpuId 0,0,0,0,0,0,0,0 in80.txt 2>&1 | tr "\r" "\n" > $f
grep -v "^\[" $f | grep -v "^$" | grep -v "(running)" | head -30
grep "(running)" $f | tail -8
grep "^\[" $f | tail -2
