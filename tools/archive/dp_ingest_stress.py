#!/usr/bin/env python3
"""Host DP path at the 8-GPU rate, without GPUs.

At 2^26 kangaroos the reference's DP suggestion for an 80-bit range drops to 11 (Kangaroo.cpp:980-993): every GPU
then delivers 2^23 * 64 / 2^11 = 262 144 distinguished points per 25 ms launch, eight of them ~85 M points/s into
ONE host table.  This tool drives exactly the code a GPU thread runs after kng_drain_view -- kng_solver's ingest
(record -> table entry -> per-consumer batch) and its consumer threads (sharded kng_dptable) -- with synthetic
engine records from `--feeders` threads, either paced like launches (`--launch-ms`) or flat out, and reports the
sustained insert rate, the load of each consumer and the memory the table takes.

    python tools/dp_ingest_stress.py --feeders 8 --consumers 16 --points-per-launch 262144 --launches 40
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kangaroo_amd import hostlib as hl  # noqa: E402
from kangaroo_amd import solver as sv  # noqa: E402


def make_records(rng, n, woff):
    rec = np.zeros(n, sv.DP_RECORD_DTYPE)
    rec["x"] = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    rec["x"][:, 3] &= np.uint64((1 << 53) - 1)  # a distinguished point: its top bits are zero
    rec["kidx"] = rng.integers(0, 1 << 23, size=n, dtype=np.uint64)
    rec["d"][:, 0] = rng.integers(0, 1 << 64, size=n, dtype=np.uint64)
    rec["d"][:, 1] = np.uint64(woff >> 64) + rng.integers(0, 1 << 8, size=n, dtype=np.uint64)
    return rec


def run(feeders, consumers, per_launch, launches, launch_ms, quiet=False):
    rp = 80
    start = int("B60E83280258A40F9CDF1649744D730D6E939DE92A2B" + "0" * 20, 16)
    key = start + 0xC0FFEE123456789ABCD
    s = sv.Solver(start, start + (1 << rp) - 1, hl.pubkey(key)[1:], consumers=consumers, seed=1)
    woff = ((1 << rp) - 1) >> 1
    # each feeder cycles through a few pre-built launches with fresh x words (the generator must not be the bottleneck)
    pools = [[make_records(np.random.default_rng(1000 * f + i), per_launch, woff) for i in range(2)] for f in range(feeders)]
    s.start_ingest(feeders)
    behind = [0.0] * feeders

    def feed(f):
        rng = np.random.default_rng(77 + f)
        t0 = time.perf_counter()
        for i in range(launches):
            rec = pools[f][i & 1]
            # new points every launch: re-randomise the sort key and the bucket word in place (cheap, 3 columns)
            rec["x"][:, :3] = rng.integers(0, 1 << 64, size=(per_launch, 3), dtype=np.uint64)
            if launch_ms > 0:
                due = t0 + (i + 1) * launch_ms * 1e-3
                now = time.perf_counter()
                if now < due:
                    time.sleep(due - now)
                else:
                    behind[f] = max(behind[f], now - due)
            s.ingest(f, rec)

    th = [threading.Thread(target=feed, args=(f,)) for f in range(feeders)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    t_fed = time.perf_counter() - t0
    ok = s.drained(600.0)
    t_all = time.perf_counter() - t0
    st = s.stats()
    load = s.consumer_load()
    total = feeders * per_launch * launches
    out = dict(points=total, fed_seconds=t_fed, drained_seconds=t_all, rate=total / t_all, fed_rate=total / t_fed,
               table_items=st["table_items"], table_bytes=st["table_bytes"], load=load, drained=ok,
               behind_ms=max(behind) * 1e3, duplicates=st["same_herd"])
    if not quiet:
        need = feeders * per_launch / (launch_ms * 1e-3) if launch_ms > 0 else 0
        print(f"{feeders} feeders x {launches} launches x {per_launch} points = {total / 1e6:.1f} M points, {consumers or 'auto'} consumers "
              f"({len(load)} threads), host cores {os.cpu_count()}")
        if need:
            print(f"paced: one launch per {launch_ms} ms per feeder = {need / 1e6:.1f} M points/s offered; worst feeder lag {out['behind_ms']:.1f} ms")
        print(f"ingest (feeder side) {out['fed_rate'] / 1e6:.1f} M points/s, end to end (all in the table) {out['rate'] / 1e6:.1f} M points/s")
        print(f"table: {st['table_items']} items, {st['table_bytes'] / 2**30:.2f} GiB = {st['table_bytes'] / max(1, st['table_items']):.1f} B/item "
              f"(32 B/item in the work file); same-x rejects {st['same_herd']}")
        print("consumer load (points): min %d max %d  (max/mean %.3f)" % (min(load), max(load), max(load) / (sum(load) / len(load))))
    s.stop()
    s.close()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--feeders", type=int, default=8)
    ap.add_argument("--consumers", type=int, default=0)
    ap.add_argument("--points-per-launch", type=int, default=262144)
    ap.add_argument("--launches", type=int, default=40)
    ap.add_argument("--launch-ms", type=float, default=0.0, help="pace each feeder like a GPU (25 = the walk kernel); 0 = flat out")
    a = ap.parse_args()
    run(a.feeders, a.consumers, a.points_per_launch, a.launches, a.launch_ms)
