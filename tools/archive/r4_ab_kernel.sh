#!/bin/bash
# round 4 kernel A/B in ONE session: resume (= HEAD), warm2 (L2 warm-up consumed behind the inversion), share4 / share4rot
# (two 256-thread blocks per CU, fixed / rotating root wave)
OUT=gpurun_out/r04_ab_kernel.txt; mkdir -p gpurun_out
run() { # name, extra sweep args
  echo -n "$1: "; KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$1.so timeout 300 python tools/sweep.py --launches 16 --groups 64 --blocks 256 $2 | grep "^asm" | grep -oE "share +[0-9]+ .*kernel +[0-9.]+ ms +[0-9.]+ MK/s" | sed 's/group.*kernel/kernel/'
}
{
echo "== parity of the share-4 builds (4096 kangaroos x 3 launches vs the oracle)"
for v in share4 share4rot; do KNG_LIB_PATH=$PWD/kangaroo_amd/lib/libkangaroo_hip_$v.so python - <<'PY'
import numpy as np, kangaroo_amd as k, kangaroo_amd.hostlib as hl
from oracle import load_oracle
orc = load_oracle()
rp, grid = 72, (4, 8)
n = grid[0] * grid[1] * 128
_, kx, ky = hl.pubkey(0xABCDEF)
x, y, d, woff = hl.create_herd(n, rp, (kx, ky), seed=5)
dd = hl.to_device_distances(d, woff)
jd, jx, jy, _ = hl.jump_table(rp)
mask = hl.dp_mask(5)
eng = k.GPUEngine(grid[0], grid[1], 0, 1 << 17, share=4, group=16)
eng.SetParams(mask, jd, jx, jy); eng.SetWildOffset(woff); eng.SetKangaroos(x, y, dd)
ox, oy, od = x.copy(), y.copy(), dd.copy()
ok = True
for _ in range(3):
    eng.callKernel(); eng.wait(); got = eng.drain(raw=True)
    want, total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask)
    gx, gy, gd = eng.GetKangaroos(raw=True)
    ok &= len(got) == total and np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gd, od)
print("share 4 parity:", "ok" if ok else "MISMATCH")
PY
done
for i in 1 2 3; do run resume ""; run warm2 ""; run share4 "--shares 4"; run share4rot "--shares 4"; done
} > $OUT 2>&1
cat $OUT
