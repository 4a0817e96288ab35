#!/usr/bin/env python3
"""End-to-end solve with the host pipeline (kangaroo_amd.solver): SURVEY 8(d) config 3 input by default
(80-bit range, key = start + 0xC0FFEE123456789ABCD), default grid, suggested DP.  Prints progress lines like
the reference's status line (Thread.cpp:254-300) and the result.
usage: python tools/solve_demo.py [--bits 80] [--gpus 0] [--max-seconds 600] [--save file.work]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd import hostlib as hl  # noqa: E402
from kangaroo_amd import solver as sv  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--bits", type=int, default=80)
ap.add_argument("--gpus", default="0")
ap.add_argument("--max-seconds", type=float, default=600)
ap.add_argument("--dp", type=int, default=-1)
ap.add_argument("--save", default="")
a = ap.parse_args()

start = int("B60E83280258A40F9CDF1649744D730D6E939DE92A2B" + "0" * 20, 16)
key = start + (0xC0FFEE123456789ABCD & ((1 << a.bits) - 1))
kxy = hl.pubkey(key)[1:]
s = sv.Solver(start, start + (1 << a.bits) - 1, kxy, gpus=tuple(int(g) for g in a.gpus.split(",")), dp=a.dp, seed=int(time.time()))
t0 = time.time()
s.start()
st = s.stats()
print(f"range 2^{st['range_power']}, {st['kangaroos']} kangaroos, dp {st['dp']}, expected ~2^{st['range_power'] / 2 + 1.05:.1f} jumps", flush=True)
rc = 0
while rc == 0 and time.time() - t0 < a.max_seconds:
    rc = s.wait(20)
    st = s.stats()
    import math

    print(f"[{time.time() - t0:6.1f} s] {st['jumps'] / max(st['seconds'], 1e-9) / 1e6:9.1f} MK/s  count 2^{math.log2(max(st['jumps'], 1)):.2f}  "
          f"DPs 2^{math.log2(max(st['dps'], 1)):.2f}  replaced {st['same_herd']}  lost {st['dps_lost']}", flush=True)
if a.save:
    s.save(a.save, True)
s.stop()
st = s.stats()
if rc == 1:
    priv = s.result()
    print(f"SOLVED in {time.time() - t0:.1f} s: 0x{priv:X}  ({'correct' if priv == key else 'WRONG'}), table {st['table_items']} DPs, "
          f"kernel {st['kernel_ms_avg']:.2f} ms/launch", flush=True)
else:
    print(f"not solved after {time.time() - t0:.1f} s (rc {rc}), table {st['table_items']} DPs", flush=True)
s.close()
