#!/bin/bash
OUT=$PWD/gpurun_out; mkdir -p $OUT
echo "== host DP path, 8 feeders paced at one 262144-point launch per 25 ms (83.9 M points/s offered)"
for C in 16 32 48 64; do ./tools/dp_ingest_bench --feeders 8 --consumers $C --launches 60 --launch-ms 25 | sed -n 1,6p; done 2>&1 | tee $OUT/r02c_dp_ingest_sweep.txt
echo "== long run: 8 feeders, default consumers, 400 launches each (839 M points = 80 % of the points of a solved 80-bit key)"
./tools/dp_ingest_bench --feeders 8 --launches 400 --launch-ms 25 2>&1 | tee $OUT/r02c_dp_ingest_long.txt
echo "== pytest -m gpu (solver + the rest)"; python -m pytest tests/test_gpu_solver.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 | tee $OUT/r02c_pytest_gpu.txt
