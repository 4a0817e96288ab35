#!/usr/bin/env python3
"""Throughput sweep of the walk kernel over the engine's tuning knobs (measurement tool).

Uses random 256-bit words as herd state: the kernel's instruction stream is data-independent
(fixed-flow inversion, constant-time multiplier), so the rate equals that of real curve points;
bench.py is the run that uses valid kangaroos.
usage: python tools/sweep.py [--grid 512,128] [--groups 32,64,128] [--blocks 64,256] [--launches 3]
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kangaroo_amd as k  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="512,128")
    ap.add_argument("--groups", default="32,64,128")
    ap.add_argument("--blocks", default="64,256")
    ap.add_argument("--launches", type=int, default=3)
    ap.add_argument("--dp", type=int, default=14)
    ap.add_argument("--shares", default="8", help="waves of a block sharing one inversion: 8 (the only form left since round 4)")
    ap.add_argument("--dsplit", type=int, default=-1, help="-1 auto / 0 / 1: low-word streaming of the distances")
    ap.add_argument("--jd-bits", type=int, default=40, help="size of the synthetic jump distances: 54+ makes both distance words stream (the non-dsplit kernels)")
    ap.add_argument("--lanes", default="", help="explicit lane counts (ragged groups); overrides --groups")
    ap.add_argument("--dp-ring", default="1", help="comma list of 0/1: DP records into a device buffer + copy / straight into pinned host memory")
    ap.add_argument("--asm", default="1", help="comma list of 0/1: compiler-scheduled loop / scheduled asm loop")
    a = ap.parse_args()
    gx, gy = (int(v) for v in a.grid.split(","))
    n = gx * gy * 128
    rng = np.random.default_rng(1)
    x = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    y = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    d = rng.integers(0, 1 << 62, size=(n, 2), dtype=np.uint64)
    jd = rng.integers(0, 1 << a.jd_bits, size=(32, 2), dtype=np.uint64)
    jd[:, 1] = 0
    jx = rng.integers(0, 1 << 64, size=(32, 4), dtype=np.uint64)
    jy = rng.integers(0, 1 << 64, size=(32, 4), dtype=np.uint64)
    mask = (~((1 << (64 - a.dp)) - 1)) & ((1 << 64) - 1) if a.dp else 0
    print(f"herd {n} = 2^{np.log2(n):.2f} kangaroos, dp {a.dp}", flush=True)
    glist = [("lanes", int(v)) for v in a.lanes.split(",")] if a.lanes else [("group", int(v)) for v in a.groups.split(",")]
    for (gk, g), b, sh, use_asm, ring in ((g, b, sh, am, rg) for g in glist
                                          for b in (int(v) for v in a.blocks.split(",")) for sh in (int(v) for v in a.shares.split(","))
                                          for am in (int(v) for v in a.asm.split(",")) for rg in (int(v) for v in a.dp_ring.split(","))):
        if True:
            eng = k.GPUEngine(gx, gy, 0, max(1 << 17, 2 * ((n * 64) >> a.dp)) if a.dp else 1 << 17, block=b, share=sh, asm=use_asm, dp_ring=ring, **({"dsplit": a.dsplit} if a.dsplit >= 0 else {}), **{gk: g})
            eng.SetParams(mask, jd, jx, jy)
            eng.SetKangaroos(x, y, d)
            eng.callKernel()
            eng.wait()
            eng.drain()
            ms = []
            t0 = time.time()
            for _ in range(a.launches):
                eng.callKernel()
                eng.wait()
                ms.append(eng.last_kernel_ms())
                nd = len(eng.drain(raw=True))
            wall = time.time() - t0
            kms = float(np.mean(ms))
            rate = n * 64 / (kms * 1e-3) / 1e6
            print(f"asm {use_asm} ring {ring} share {sh} group {eng.get_option('group'):4d} block {b:4d} lanes {eng.get_option('lanes'):7d} waves/CU {eng.get_option('waves_per_cu'):3d}: "
                  f"kernel {kms:9.2f} ms  {rate:10.1f} MK/s  ({rate * 160 / 1e6:6.3f} TB/s @160B)  wall/launch {wall / a.launches * 1e3:8.2f} ms  DPs {nd}",
                  flush=True)
            eng.close()


if __name__ == "__main__":
    main()
