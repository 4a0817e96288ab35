#!/bin/bash
# Regenerates tests/golden/ref_workfile_48bit.bin.gz: a HEADW work file WRITTEN BY THE REFERENCE PROGRAM
# (oracle/_ref/kangaroo_cpu, built from /root/reference by `make -C oracle ref`): 48-bit range, one CPU thread,
# dp 12, saved with kangaroos (-ws) after about one second.  Build-container only.  The herd is seeded from
# the clock, so every run gives a different (equally valid) file; the committed one is the fixture.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
python - "$T/in48.txt" <<PY
import sys
sys.path.insert(0, "$ROOT")
from kangaroo_amd import hostlib as hl
start = 0x5B3F38AF935A3640D158E871CE6E9666DB862636383386EE000000000000
key = start + 0x9D3A7C1B2F55
_, kx, ky = hl.pubkey(key)
open(sys.argv[1], "w").write(f"{start:064X}\n{start + (1 << 48) - 1:064X}\n{'02' if ky % 2 == 0 else '03'}{kx:064X}\n")
PY
(cd $T && (timeout 6 $ROOT/oracle/_ref/kangaroo_cpu -t 1 -d 12 -w ref.work -wi 1 -ws in48.txt > run.log 2>&1 || true))
test -s $T/ref.work
gzip -9 -c $T/ref.work > $ROOT/tests/golden/ref_workfile_48bit.bin.gz
ls -la $ROOT/tests/golden/ref_workfile_48bit.bin.gz
rm -rf $T
