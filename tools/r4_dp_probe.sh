#!/bin/bash
OUT=gpurun_out/r04_dp_probe5.txt; mkdir -p gpurun_out
T="./tools/dp_table_bench 80000000"
B="./tools/dp_ingest_bench --feeders 8 --launches 60"
F="end to end|summed"
{
for S in 32 16 8; do echo "== split $S, one thread, 80 M inserts"; KNGT_SPLIT_AVG=$S $T 4 | cut -c1-60; done
for S in 32 16 8; do for C in 12 16; do echo "== split $S, unpaced, $C consumers"; KNGT_SPLIT_AVG=$S $B --consumers $C | grep -E "$F"; done; done
} > $OUT 2>&1
cat $OUT
