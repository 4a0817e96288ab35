#!/bin/bash
# round-2 session 1: sensitivity map (ablated builds) + memory energy probe
OUT=$PWD/gpurun_out; mkdir -p $OUT
{
echo "== memory power probe (16 B/lane nt stream, 2 GiB buffers)"
python tools/ablate_run.py --cmd "./tools/mem_power_probe 0 0 5" --cmd "./tools/mem_power_probe 1 0 5" --cmd "./tools/mem_power_probe 2 0 5" --cmd "./tools/mem_power_probe 0 8 5" --cmd "./tools/mem_power_probe 0 24 5"
echo "== ablations (walk kernel, 2^23 kangaroos, group 64, share 2)"
python tools/ablate_run.py --launches 240 base no_s_traffic no_inversion no_comba_carry no_fold no_state_store no_memory s_plain base
} 2>&1 | tee $OUT/r02_s1_ablation.txt
