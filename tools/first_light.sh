#!/bin/bash
# first GPU contact: parity tests, instruction micro-benchmarks, knob sweep
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_gpu.txt
timeout 300 ./tools/ubench 2>&1 | tee gpurun_out/ubench.txt
timeout 900 python tools/sweep.py --launches 2 2>&1 | tee gpurun_out/sweep.txt
