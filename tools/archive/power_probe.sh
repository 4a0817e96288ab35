#!/bin/bash
# clocks and power while the walk kernel runs (evidence for "power-limited"): samples rocm-smi /
# amd-smi once a second during a 12 s run of the sweep tool.
OUT=$PWD/gpurun_out; mkdir -p $OUT
python tools/sweep.py --launches 400 --groups 64 --blocks 256 > $OUT/power_sweep.txt 2>&1 &
PID=$!
sleep 4
for i in 1 2 3 4 5; do
  echo "--- sample $i"
  (rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)" | head -8) || true
  (amd-smi metric -p -c 2>/dev/null | grep -E "SOCKET_POWER|CURRENT_POWER|GFX_0|CLK|POWER" | head -8) || true
  sleep 1
done
kill $PID 2>/dev/null; wait $PID 2>/dev/null
tail -1 $OUT/power_sweep.txt
(rocm-smi --showmaxpower 2>/dev/null | grep -i -E "max|cap" | head -4) || true
