#!/usr/bin/env python3
"""Walk-kernel rate against the herd size (-g gridX,gridY => 128*gridX*gridY kangaroos) with the engine's
default geometry.  Measurement tool (synthetic state, jump distances < 2^40, dp 14).
usage: python tools/herd_sweep.py [grid ...]      e.g.  64,128 128,128 256,128 512,128 1024,128 2048,128"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kangaroo_amd as k  # noqa: E402

for g in sys.argv[1:] or ["64,128", "128,128", "256,128", "512,128", "1024,128", "2048,128"]:
    gx, gy = (int(v) for v in g.split(","))
    n = gx * gy * 128
    rng = np.random.default_rng(1)
    x = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    y = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    d = rng.integers(0, 1 << 62, size=(n, 2), dtype=np.uint64)
    jd = rng.integers(0, 1 << 40, size=(32, 2), dtype=np.uint64)
    jd[:, 1] = 0
    jx = rng.integers(0, 1 << 64, size=(32, 4), dtype=np.uint64)
    jy = rng.integers(0, 1 << 64, size=(32, 4), dtype=np.uint64)
    eng = k.GPUEngine(gx, gy, 0, 1 << 18)
    eng.SetParams(0xFFFC000000000000, jd, jx, jy)
    eng.SetKangaroos(x, y, d)
    del x, y, d
    eng.callKernel()
    eng.wait()
    eng.drain()
    ms = []
    for _ in range(5):
        eng.callKernel()
        eng.wait()
        ms.append(eng.last_kernel_ms())
        eng.drain(raw=True)
    m = float(np.mean(ms))
    grp, lanes, wpc = eng.get_option("group"), eng.get_option("lanes"), eng.get_option("waves_per_cu")
    print(f"grid {gx}x{gy}: 2^{np.log2(n):.0f} kangaroos, group {grp}, lanes {lanes}, waves/CU {wpc}: "
          f"kernel {m:8.2f} ms  {n * 64 / m / 1e3:9.1f} MK/s", flush=True)
    eng.close()
