#!/bin/bash
# The reference CLI with the link-time replacements solving BASELINE configs[2] for real: 80-bit range, key = start +
# 0xC0FFEE123456789ABCD, default grid, the program's own DP.  Expected 2^41.1 jumps = ~95 s at 25 GK/s (the time of a
# kangaroo solve is random: x0.3 .. x3).  usage: tools/ref_program_solve80.sh [timeout seconds=420]
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
printf "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000\nB60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF\n03AAE5826AD4C307F4C42A9D853E151CDB90BB676320DDA88177303F5CFE1E9F62\n" > in80key.txt
f=$OUT/ref_program_solve80.txt
KNG_STATS=1 timeout ${1:-420} stdbuf -o0 -e0 $ROOT/oracle/_ref/kangaroo_mi355x -t 0 -gpu in80key.txt 2>&1 | tr "\r" "\n" > $f
grep -v "^\[" $f | grep -v "^$"
grep "^\[" $f | tail -2
grep -q "Priv: 0xB60E83280258A40F9CDF1649744D730D6E939DE92A2B0C0FFEE123456789ABCD" $f && echo "SOLVED: key correct"
