#!/bin/bash
# optimisation round trip: parity first, then field-op micro-benchmarks and the knob sweep
TAG=${1:-opt}
OUT=$PWD/gpurun_out
mkdir -p $OUT
python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/${TAG}_pytest_gpu.txt
timeout 300 ./tools/ubench 2>&1 | grep -A40 "256-bit field" | tee $OUT/${TAG}_ubench.txt
timeout 900 python tools/sweep.py --launches 2 --groups ${GROUPS_LIST:-16,32,64,128} --blocks ${BLOCKS_LIST:-64,256} 2>&1 | tee $OUT/${TAG}_sweep.txt
