#!/bin/bash
# End-to-end rate of the reference PROGRAM on the engine at BASELINE configs[2] (80-bit range, default grid 2^23 kangaroos,
# the DP size the program suggests itself), three minutes each:
#   kangaroo_hip     reference host code unmodified (SolveKeyGPU + HashTable) + our class GPUEngine
#   kangaroo_mi355x  the same with HashTable.o and Kangaroo::SolveKeyGPU replaced at link time (HashTable_kng / SolveKeyGPU_kng)
# usage: tools/ref_program_rate.sh [seconds=180] [extra program options, e.g. -d 11]
# The key is not in the range: the search cannot end early.  Rates are computed from the Count column of the program's own
# status line (first line at >= 10 s to the last line) and, for kangaroo_mi355x, from its KNG_STATS line (exact).
SECS=${1:-180}; shift
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out; mkdir -p $OUT
ROOT=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp
printf "B60E83280258A40F9CDF1649744D730D6E939DE92A2B00000000000000000000\nB60E83280258A40F9CDF1649744D730D6E939DE92A2BFFFFFFFFFFFFFFFFFFFF\n03BB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4\n" > in80.txt
for prog in ${PROGS:-kangaroo_mi355x kangaroo_hip}; do   # PROGS="kangaroo_hip_ht" = only class HashTable replaced
  f=$OUT/ref_program_rate_${prog}.txt
  # -m: stop by itself after SECS seconds at 25 GK/s (so that the KNG_STATS line is printed); timeout is the backstop
  M=$(python3 -c "print('%.3f' % ($SECS*25.6e9/2**${EXPECTED_LOG2:-41.11}))")   # "Expected operations: 2^41.11" at DP 14; set EXPECTED_LOG2 for other -d
  KNG_STATS=1 timeout $((SECS+60)) stdbuf -o0 -e0 $ROOT/oracle/_ref/$prog -t 0 -gpu -m $M "$@" in80.txt 2>&1 | tr "\r" "\n" > $f
  echo "== $prog $@"
  grep -v "^\[" $f | grep -v "^$" | head -24
  python3 - $f <<'PY'
import re, sys
st = []
def secs(t):  # Kangaroo::GetTimeStr: "42s", "01:10", "01:02:03"
    if t.endswith("s"):
        return int(t[:-1])
    v = 0
    for part in t.split(":"):
        v = v * 60 + int(part)
    return v
for m in re.finditer(r"\[([0-9.]+) MK/s\]\[GPU [0-9.]+ MK/s\]\[Count 2\^([0-9.]+)\]\[Dead (\d+)\]\[([0-9:]+s?) \(Avg [^)]*\)\]\[([0-9.]+/[0-9.]+[MG]B)\]", open(sys.argv[1]).read()):
    st.append((secs(m.group(4)), 2.0 ** float(m.group(2)), m.group(5)))
if len(st) > 4:
    for lo, hi in ((10, 60), (60, 120), (120, 1e9), (10, 1e9)):
        w = [s for s in st if lo <= s[0] <= hi]
        if len(w) > 2:
            print("  Count column %3d..%3d s: %.2f GK/s   table %s" % (w[0][0], w[-1][0], (w[-1][1] - w[0][1]) / (w[-1][0] - w[0][0]) / 1e9, w[-1][2]))
PY
done
