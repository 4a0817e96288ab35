#!/usr/bin/env python3
"""A/B of the inversion-sharing form at herds too small to fill the chip (BASELINE configs[2] read literally: 2*CU x 128 =
65 536 kangaroos): 512-thread blocks / one inversion per CU ("share" 8) against 256-thread blocks / one per four waves
("share" 4), alternating inside one session.  usage: tools/small_herd_ab.py [rounds=3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import kangaroo_amd as k  # noqa: E402
import kangaroo_amd.hostlib as hl  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
cu = k.device_info(0)["cu_count"]
jd, jx, jy, _ = hl.jump_table(80)
for gx, gy in ((2 * cu, 1), (cu, 1), (2 * cu, 2), (2 * cu, 4), (2 * cu, 8)):
    n = gx * gy * 128
    for r in range(rounds):
        for share in (8, 4):
            with k.GPUEngine(gx, gy, 0, 1 << 17, share=share) as eng:
                eng.SetParams(hl.dp_mask(12), jd, jx, jy)
                eng.CreateHerdOnDevice(80, seed=7)
                ms = []
                for i in range(14):
                    eng.callKernel()
                    eng.wait()
                    eng.drain(raw=True)
                    ms.append(eng.last_kernel_ms())
                m = float(np.median(ms[4:]))
                print(f"herd {gx}x{gy}x128 = {n:8d}  share {share} group {eng.get_option('group'):3d} lanes {eng.get_option('lanes'):7d}: kernel {m:8.3f} ms  {n * 64 / m / 1e3:9.1f} MK/s", flush=True)
