#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== new tests"
timeout 1500 python -m pytest tests -m gpu -x -q -k "dp_ring or two_engines or configs4 or reference_program_solves_in_txt" 2>&1 | tail -5 | tee $OUT/r03e_new_tests.txt
cat $OUT/r03_configs4_default_herd.txt
echo "== DP straight into pinned host memory (ring 1) vs device buffer + copy (ring 0): DP 14 and DP 11 (the 8-GPU rate: 262144 points per launch)"
for dp in 14 11; do for i in 1 2; do timeout 300 python tools/sweep.py --launches 12 --groups 64 --blocks 256 --asm 1 --dp $dp --dp-ring 0,1 | grep "^asm"; done; done 2>&1 | tee $OUT/r03e_ab_dp_ring.txt
echo "== power and clock: compiler loop vs asm loop (240 launches each)"
python tools/ablate_run.py --cmd "python tools/sweep.py --launches 240 --groups 64 --blocks 256 --asm 0" --cmd "python tools/sweep.py --launches 240 --groups 64 --blocks 256 --asm 1" --cmd "python tools/sweep.py --launches 240 --groups 64 --blocks 256 --asm 0" --cmd "python tools/sweep.py --launches 240 --groups 64 --blocks 256 --asm 1" 2>&1 | cut -c1-400 | tee $OUT/r03e_power_clock.txt
echo "== inversion share: herd 2^24 (group 128) for reference"
timeout 300 python tools/sweep.py --grid 1024,128 --launches 6 --groups 64,128 --blocks 256 --asm 1 | grep "^asm" | tee $OUT/r03e_herd24.txt
