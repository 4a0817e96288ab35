#!/usr/bin/env python3
"""Herd size x group sweep of the final walk kernel in ONE session, with package power and GFX clock sampled per cell
(the default rule of choose_geometry, kng_engine.hip, rests on round-1 data that predates the scheduled
loop, the CU-wide inversion and resumed launches).

Random 256-bit words as herd state: the instruction stream is data-independent (tools/sweep.py).  Every cell: 2 untimed launches,
then `--launches` timed ones (HIP events on the engine's stream), resumed launches as in production.
usage (GPU box): python tools/geometry_sweep.py [--log2 21,22,23,24,25] [--groups 32,64,128,256] [--launches 12]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import kangaroo_amd as k  # noqa: E402
from kangaroo_amd.telemetry import GpuSampler  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2", default="21,22,23,24,25")
    ap.add_argument("--groups", default="32,64,128,256")
    ap.add_argument("--launches", type=int, default=12)
    ap.add_argument("--dp", type=int, default=14)
    a = ap.parse_args()
    rng = np.random.default_rng(1)
    jd = rng.integers(0, 1 << 40, size=(32, 2), dtype=np.uint64)
    jd[:, 1] = 0
    jx = rng.integers(0, 1 << 64, size=(32, 4), dtype=np.uint64)
    jy = rng.integers(0, 1 << 64, size=(32, 4), dtype=np.uint64)
    mask = (~((1 << (64 - a.dp)) - 1)) & ((1 << 64) - 1)
    print(f"# {k.device_info(0)['name']}, dp {a.dp}, {a.launches} timed launches per cell, default knobs except group", flush=True)
    print("# log2(herd) group lanes waves/CU share |  kernel ms (min..max)    MK/s   frac@160B |  W(median)  W(energy ctr)  sclk MHz | default?", flush=True)
    for lg in (int(v) for v in a.log2.split(",")):
        gx = (1 << lg) // (128 * 128)
        n = gx * 128 * 128
        x = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
        y = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
        d = rng.integers(0, 1 << 62, size=(n, 2), dtype=np.uint64)
        d[:, 1] = 0
        with k.GPUEngine(gx, 128, 0, 1 << 17) as e0:
            default_group = e0.get_option("group")
        best = None
        for g in (int(v) for v in a.groups.split(",")):
            try:
                eng = k.GPUEngine(gx, 128, 0, max(1 << 17, 2 * ((n * 64) >> a.dp)), group=g)
            except Exception as ex:  # noqa: BLE001
                print(f"{lg:4d} {g:5d}: {ex}", flush=True)
                continue
            with eng:
                eng.SetParams(mask, jd, jx, jy)
                eng.SetKangaroos(x, y, d)
                for _ in range(2):
                    eng.callKernel()
                    eng.wait()
                    eng.drain(raw=True)
                ms = []
                with GpuSampler([0], hz=100.0) as smp:
                    for _ in range(a.launches):
                        eng.callKernel()
                        eng.wait()
                        ms.append(eng.last_kernel_ms())
                        eng.drain(raw=True)
                s = smp.summary()
                dev = (s.get("devices") or [{}])[0] if s.get("available") else {}
                kms = float(np.mean(ms))
                rate = n * 64 / (kms * 1e-3) / 1e6
                pw = (dev.get("power_w") or {}).get("median")
                ck = (dev.get("sclk_mhz") or {}).get("median")
                print(f"{lg:4d} {eng.get_option('group'):5d} {eng.get_option('lanes'):8d} {eng.get_option('waves_per_cu'):4d} {eng.get_option('share'):3d} | "
                      f"{kms:9.3f} ({min(ms):.3f}..{max(ms):.3f}) {rate:9.1f} {rate * 160 / 8e6:7.4f} | {pw} {dev.get('power_w_from_energy_counter')} {ck} | "
                      f"{'default' if g == default_group else ''}", flush=True)
                if best is None or rate > best[1]:
                    best = (g, rate)
        if best:
            print(f"# 2^{lg}: best group {best[0]} at {best[1]:.1f} MK/s; default rule picks {default_group}", flush=True)


if __name__ == "__main__":
    main()
