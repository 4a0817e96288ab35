#!/bin/bash
# The reference CLI with the link-time replacements (kangaroo_mi355x) solving BASELINE configs[2] for real ACROSS RESTARTS:
# 80-bit range, key = start + 0xC0FFEE123456789ABCD, default grid, -d 16, `-ws -wi 20`; the process is killed after SECS
# seconds, restarted with `-i` from its last file, killed again, restarted, and then left to finish.  Every save happens with
# the GPU walking on (Backup_kng.cpp + device snapshot); every restart uploads the file's records and unpacks them on the device.
# usage: tools/ref_program_solve80_resume.sh [seconds per interrupted leg=45] [timeout of the last leg=600]
SECS=${1:-45}; LAST=${2:-600}
ROOT=${GRAFT_REPO_ROOT:-$PWD}; OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
printf "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000\nB60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF\n03AAE5826AD4C307F4C42A9D853E151CDB90BB676320DDA88177303F5CFE1E9F62\n" > in80key.txt
EXE=$ROOT/oracle/_ref/kangaroo_mi355x
f=$OUT/ref_program_solve80_resume.txt; : > $f
rm -f solve80.work
for leg in 1 2 3; do
  IN=""; [ -f solve80.work ] && IN="-i solve80.work"
  echo "== leg $leg (killed after $SECS s) $IN" | tee -a $f
  KNG_STATS=1 timeout -s KILL $SECS stdbuf -o0 -e0 $EXE -t 0 -gpu -d 16 $IN -ws -w solve80.work -wi 20 in80key.txt 2>&1 | tr "\r" "\n" > leg.txt
  grep -v "^\[" leg.txt | grep -v "^$" | grep -E "LoadWork|Fectch|kangaroos \[|SaveWork|done \[|Priv|pool of" | cut -c1-200 | tee -a $f
  grep "^\[" leg.txt | tail -1 | tee -a $f
  grep -q "Priv: 0x" leg.txt && break
  $EXE -winfo solve80.work | grep -E "Count|Kangaroos|DP Count" | tee -a $f
done
if ! grep -q "Priv: 0x" leg.txt; then
  echo "== last leg: -i solve80.work, until the key is found" | tee -a $f
  KNG_STATS=1 timeout $LAST stdbuf -o0 -e0 $EXE -t 0 -gpu -d 16 -i solve80.work -ws -w solve80.work -wi 20 in80key.txt 2>&1 | tr "\r" "\n" > leg.txt
  grep -v "^\[" leg.txt | grep -v "^$" | grep -E "LoadWork|Fectch|kangaroos \[|done \[|Priv|Key#|SolveKeyGPU_kng|Done" | cut -c1-330 | tee -a $f
  grep "^\[" leg.txt | tail -1 | tee -a $f
fi
grep -q "Priv: 0xB60E83280258A40F9CDF1649744D730D6E939DE92A2B0C0FFEE123456789ABCD" leg.txt && echo "SOLVED ACROSS RESTARTS: key correct" | tee -a $f
