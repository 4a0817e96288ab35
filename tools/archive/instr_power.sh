#!/bin/bash
# package power and GFX clock while ONE instruction runs back to back on every SIMD (2 waves/SIMD):
# rough energy per wave-instruction for planning (tools/gen_instr_probe.py "power" mode).
# indices into the TESTS table of tools/gen_instr_probe.py
for IDX in ${@:-37 0 1 2 11 21 22 23 15 17 27}; do
  ./tools/instr_probe power $IDX 4 2 > /tmp/ip.txt &
  PID=$!
  sleep 2.5
  P=$(rocm-smi --showpower 2>/dev/null | grep -oE "Power \(W\): [0-9.]+" | grep -oE "[0-9.]+$")
  C=$(rocm-smi --showclocks 2>/dev/null | grep -oE "sclk clock level: [0-9]+: \([0-9]+Mhz\)" | grep -oE "[0-9]+Mhz")
  wait $PID
  echo "$(cat /tmp/ip.txt)   power ${P} W   sclk ${C}"
done
