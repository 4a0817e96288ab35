#!/usr/bin/env python3
"""End-to-end solve with the host pipeline (kangaroo_amd.solver): SURVEY 8(d) config 3 input by default
(80-bit range, key = start + 0xC0FFEE123456789ABCD), default grid, suggested DP.  Prints progress lines like
the reference's status line (Thread.cpp:254-300) and the result.
usage: python tools/solve_demo.py [--bits 80 | --input in.txt] [--gpus 0,1,..] [--max-seconds 600] [--save f.work] [--load f.work]
--input takes the reference's configuration file (README.md:96-109): range start, range end, one public key
(compressed or uncompressed hex), one per line."""
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
ap.add_argument("--load", default="", help="resume from a work file (ours or the reference's)")
ap.add_argument("--input", default="", help="reference-style input file: start, end, public key")
a = ap.parse_args()

P = (1 << 256) - (1 << 32) - 977
if a.input:
    lines = [ln.strip() for ln in open(a.input) if ln.strip()]
    start, end = int(lines[0], 16), int(lines[1], 16)
    pub = lines[2]
    if pub[:2] in ("02", "03"):
        kx = int(pub[2:], 16)
        ky = pow((kx * kx * kx + 7) % P, (P + 1) // 4, P)
        if (ky & 1) != (int(pub[:2], 16) & 1):
            ky = P - ky
    else:
        kx, ky = int(pub[2:66], 16), int(pub[66:130], 16)
    kxy, key = (kx, ky), None
else:
    start = int("B60E83280258A40F9CDF1649744D730D6E939DE92A2B" + "0" * 20, 16)
    end = start + (1 << a.bits) - 1
    key = start + (0xC0FFEE123456789ABCD & ((1 << a.bits) - 1))
    kxy = hl.pubkey(key)[1:]
s = sv.Solver(start, end, kxy, gpus=tuple(int(g) for g in a.gpus.split(",")), dp=a.dp, seed=int(time.time()))
if a.load:
    s.load(a.load)
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
    verdict = "correct" if (key is None and hl.pubkey(priv)[1:] == kxy) or priv == key else "WRONG"
    print(f"SOLVED in {time.time() - t0:.1f} s: 0x{priv:X}  ({verdict}), table {st['table_items']} DPs, "
          f"kernel {st['kernel_ms_avg']:.2f} ms/launch", flush=True)
else:
    print(f"not solved after {time.time() - t0:.1f} s (rc {rc}), table {st['table_items']} DPs", flush=True)
s.close()
