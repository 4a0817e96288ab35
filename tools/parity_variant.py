import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys, numpy as np, kangaroo_amd as k, kangaroo_amd.hostlib as hl
from oracle import load_oracle
orc = load_oracle()
ok = True
for rp, grid, group in ((72, (4, 8), 16), (72, (3, 5), 0), (125, (8, 8), 0)):
    n = grid[0] * grid[1] * 128
    _, kx, ky = hl.pubkey(0xABCDEF)
    x, y, d, woff = hl.create_herd(n, rp, (kx, ky), seed=5)
    dd = hl.to_device_distances(d, woff)
    jd, jx, jy, _ = hl.jump_table(rp)
    mask = hl.dp_mask(5)
    eng = k.GPUEngine(grid[0], grid[1], 0, 1 << 17, **({"group": group} if group else {}))
    eng.SetParams(mask, jd, jx, jy); eng.SetWildOffset(woff); eng.SetKangaroos(x, y, dd)
    ox, oy, od = x.copy(), y.copy(), dd.copy()
    for _ in range(3):
        eng.callKernel(); eng.wait(); got = eng.drain(raw=True)
        want, total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask)
        gx, gy, gd = eng.GetKangaroos(raw=True)
        ok &= len(got) == total and np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gd, od)
    eng.close()
print("variant parity:", "ok" if ok else "MISMATCH")
