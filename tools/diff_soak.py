#!/usr/bin/env python3
"""Differential soak: two engines on ONE device walk the SAME device-built herd with two independent code paths and are compared
launch by launch.

What the invariant audit (kng_audit_*) cannot see -- a distinguished point that was not emitted (ballot / compaction, re-entry
from the exact path, ring slot arithmetic), or a wrong but self-consistent jump choice -- shows up here as a difference between

  --variant asm     "asm" 1 (the scheduled gfx950 loop, kng_walk_asm.h) against "asm" 0 (the compiler-scheduled loop: another
                    instruction stream, another register allocation, the general field arithmetic; itself compared with the CPU
                    oracle in tests/test_gpu_parity.py)
  --variant dsplit  "dsplit" 1 (low-word distance streaming with L2-atomic carries) against "dsplit" 0 (full 128-bit distances)
  --variant share   "share" 4 (256-thread blocks, one-level inversion tree: what small herds get since round 5) against "share" 8
                    (512-thread blocks, two-level tree), at a herd both forms can walk

Every launch: the two DP multisets (x, device distance, kidx of every record) must be equal.  Every --state-every launches and at
the end: all (x, y, d) of both herds must be equal.  Nothing here uses the oracle; the reference's closest tool is the CPU/GPU
comparison of `-check` (Check.cpp:526-617), two launches of a small herd."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def canon(recs):
    """records ordered by (kidx, low distance word): a kangaroo may pass two distinguished points in one launch"""
    order = np.lexsort((recs["d"][:, 0], recs["kidx"]))
    r = recs[order]
    return r["x"], r["d"], r["kidx"]


def run(args):
    import kangaroo_amd as k
    import kangaroo_amd.hostlib as hl

    gx, gy = args.grid
    rp = args.range_power
    jd, jx, jy, _ = hl.jump_table(rp)
    _, kx, ky = hl.pubkey(0x1234567 + (1 << (rp - 2)))
    opts = {"asm": ({"asm": 1}, {"asm": 0}), "dsplit": ({"dsplit": 1}, {"dsplit": 0}), "share": ({"share": 4}, {"share": 8})}[args.variant]
    per_launch = (gx * gy * 128 * 64) >> args.dp
    cap = max(1 << 17, 2 * per_launch + 4096)
    engines = [k.GPUEngine(gx, gy, args.device, cap, **o) for o in opts]
    for e in engines:
        e.SetParams(hl.dp_mask(args.dp), jd, jx, jy)
        e.CreateHerdOnDevice(rp, (kx, ky), seed=args.seed)
    n = engines[0].nbKangaroo
    res = {"variant": args.variant, "options": [dict(o) for o in opts], "grid": [gx, gy], "kangaroos": n, "range_power": rp, "dp": args.dp,
           "launches": 0, "jumps_per_engine": 0, "dp_records_compared": 0, "dp_differences": 0, "state_compares": 0, "state_differences": 0,
           "exact_exits": [0, 0], "dsplit_in_effect": [e.get_option("dsplit") for e in engines], "share_in_effect": [e.get_option("share") for e in engines], "lost": 0, "kernel_ms": [[], []]}
    t0 = time.time()

    def compare_state():
        a = engines[0].GetKangaroos(raw=True)
        b = engines[1].GetKangaroos(raw=True)
        res["state_compares"] += 1
        bad = sum(int(np.count_nonzero(np.any(u != v, axis=1))) for u, v in zip(a, b))
        res["state_differences"] += bad
        return bad

    if compare_state():
        raise SystemExit("the two herds differ before the first launch")
    for i in range(args.launches):
        for e in engines:
            e.callKernel()
        recs = []
        for j, e in enumerate(engines):
            e.wait()
            recs.append(e.drain_records())
            res["lost"] += e.lastLost
            res["exact_exits"][j] += e.get_option("exact_exits")
            if i % 64 == 5:
                res["kernel_ms"][j].append(round(e.last_kernel_ms(), 3))
        a, b = canon(recs[0]), canon(recs[1])
        same = len(recs[0]) == len(recs[1]) and all(np.array_equal(u, v) for u, v in zip(a, b))
        res["dp_records_compared"] += len(recs[0])
        if not same:
            res["dp_differences"] += 1
            print(f"launch {i}: DP multisets differ ({len(recs[0])} vs {len(recs[1])} records)", flush=True)
        res["launches"] = i + 1
        if (i + 1) % args.state_every == 0 and i + 1 < args.launches:
            bad = compare_state()
            print(f"launch {i + 1}: state {'equal' if not bad else f'{bad} rows differ'}, {res['dp_records_compared']} records compared, "
                  f"{time.time() - t0:.0f} s", flush=True)
        if res["dp_differences"] > 8:
            break
    compare_state()
    res["jumps_per_engine"] = res["launches"] * n * 64
    res["log2_jumps_per_engine"] = round(float(np.log2(max(res["jumps_per_engine"], 1))), 3)
    res["seconds"] = round(time.time() - t0, 1)
    res["kernel_ms"] = [round(float(np.median(v)), 3) if v else None for v in res["kernel_ms"]]
    res["clean"] = res["dp_differences"] == 0 and res["state_differences"] == 0 and res["lost"] == 0
    for e in engines:
        e.close()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--variant", choices=("asm", "dsplit", "share"), default="asm")
    ap.add_argument("--launches", type=int, default=128)
    ap.add_argument("--grid", type=lambda s: tuple(int(v) for v in s.split(",")), default=(512, 128))
    ap.add_argument("--range-power", type=int, default=80)
    ap.add_argument("--dp", type=int, default=14)
    ap.add_argument("--state-every", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0x50AC)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    r = run(a)
    line = json.dumps(r)
    print(line)
    if a.out:
        with open(a.out, "w") as f:
            f.write(line + "\n")
    sys.exit(0 if r["clean"] else 1)
