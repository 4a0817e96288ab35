#!/bin/bash
# round-2 session 2: parity of the new multiplier + A/B against the round-1 kernel
OUT=$PWD/gpurun_out; mkdir -p $OUT
{
echo "== pytest -m gpu (primitives, walks)"
python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5
echo "== A/B (sweep, 2^23 kangaroos, group 64, share 2): r1 kernel / single-chain fold only / fold + first-carry elision"
for i in 1 2 3; do
python tools/ablate_run.py --launches 120 r1 full_comba base
done
} 2>&1 | tee $OUT/r02_s2_fold.txt
