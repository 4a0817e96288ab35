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
ap.add_argument("--audit-every", type=float, default=0, help="seconds between whole-herd audits while running (0 = only at the end)")
ap.add_argument("--no-audit", action="store_true", help="skip the closing whole-run audit (herd + every table entry)")
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
import math

next_audit = a.audit_every or float("inf")
while rc == 0 and time.time() - t0 < a.max_seconds:
    rc = s.wait(min(20, max(0.5, next_audit - (time.time() - t0))))
    st = s.stats()
    if rc == 0 and time.time() - t0 >= next_audit:
        # the GPUs pause at a launch boundary (as for a save), every kangaroo is re-derived from its distance, they resume
        au = s.audit(False)
        print(f"[{time.time() - t0:6.1f} s] audit while running: {au['kangaroos']} kangaroos re-derived from their distances, "
              f"{au['kangaroo_mismatches']} mismatches ({au['herd_ms']:.1f} ms on the device, {au['seconds'] * 1e3:.0f} ms pause)", flush=True)
        next_audit += a.audit_every

    print(f"[{time.time() - t0:6.1f} s] {st['jumps'] / max(st['seconds'], 1e-9) / 1e6:9.1f} MK/s  count 2^{math.log2(max(st['jumps'], 1)):.2f}  "
          f"DPs 2^{math.log2(max(st['dps'], 1)):.2f}  replaced {st['same_herd']}  lost {st['dps_lost']}", flush=True)
if a.save:
    s.save(a.save, True)
if not a.no_audit:
    # whole-run audit (kngs_audit): a walk error is permanent for its kangaroo, so a clean herd after J jumps certifies all J;
    # the table half re-derives every stored distinguished point the way the reference's -wcheck does (Check.cpp:141-411)
    au = s.audit(True)
    st = s.stats()
    print(f"AUDIT after 2^{math.log2(max(st['jumps'], 1)):.2f} jumps: {au['kangaroos']} kangaroos (x and y) and {au['table_points']} table entries "
          f"re-derived from their distances on the device: {au['kangaroo_mismatches']} + {au['table_mismatches']} mismatches; "
          f"{st['same_herd']} kangaroos had been replaced on the way; device {au['herd_ms']:.1f} + {au['table_ms']:.1f} ms, wall {au['seconds']:.2f} s", flush=True)
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
