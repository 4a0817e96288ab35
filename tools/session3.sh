#!/bin/bash
OUT=$PWD/gpurun_out
{
echo "== energy per byte against working-set size (copy = read+write, 16 B/lane nt)"
for MB in 8 32 64 100 160 512 2048; do
python tools/ablate_run.py --cmd "./tools/mem_power_probe 0 0 4 $MB"
done
for MB in 16 64 128 4096; do
python tools/ablate_run.py --cmd "./tools/mem_power_probe 1 0 4 $MB"
done
} 2>&1 | tee $OUT/r02_s3_mall.txt
