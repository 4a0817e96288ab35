#!/usr/bin/env python3
"""Run each ablated build (build/abl/<name>/, round 2) -- or any command (tools/r3_sensitivity.sh, tools/r3_stalls.sh) -- while sampling package power and GFX clock.

usage (on the GPU box): python tools/ablate_run.py [--launches 160] [name ...]
       python tools/ablate_run.py --cmd "./tools/mem_power_probe 0 0 5" --cmd "./tools/mem_power_probe 1 0 5"
Prints one line per run: walk rate (from tools/sweep.py), median power / clock of the samples taken while it ran.
"""
import argparse
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def smi_sample():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
    except Exception:
        return None
    p = re.search(r"Power \(W\): ([0-9.]+)", out)
    c = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else None, int(c.group(1)) if c else None)


def run_sampled(cmd, env=None, settle=3.0):
    """run cmd; sample power from `settle` seconds after start until it exits"""
    samples = []
    stop = threading.Event()

    def poll():
        t0 = time.time()
        while not stop.is_set():
            s = smi_sample()
            if s and time.time() - t0 > settle:
                samples.append(s)
            time.sleep(0.25)

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    res = subprocess.run(cmd, shell=isinstance(cmd, str), capture_output=True, text=True, env=env, cwd=ROOT)
    stop.set()
    th.join()
    pw = sorted(s[0] for s in samples if s[0] is not None)
    ck = sorted(s[1] for s in samples if s[1] is not None)
    # the tail of the run (teardown) pulls the low end down: report the upper-middle of the distribution
    med = lambda v: v[(2 * len(v)) // 3] if v else float("nan")  # noqa: E731
    return res, med(pw), med(ck), len(pw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--launches", type=int, default=160)
    ap.add_argument("--cmd", action="append", default=[])
    ap.add_argument("--sweep-args", default="--groups 64 --blocks 256")
    ap.add_argument("names", nargs="*")
    a = ap.parse_args()
    for c in a.cmd:
        res, pw, ck, n = run_sampled(c, settle=1.5)
        last = (res.stdout.strip().splitlines() or ["?"])[-1]
        print(f"{c:40s}: {last}   power {pw:.0f} W  sclk {ck} MHz ({n} samples)", flush=True)
    if a.cmd and not a.names:
        return
    names = a.names or sorted(os.listdir(os.path.join(ROOT, "build", "abl")))
    for name in names:
        lib = os.path.join(ROOT, "build", "abl", name, "libkangaroo_hip.so")
        if not os.path.exists(lib):
            print(f"{name}: not built")
            continue
        env = dict(os.environ, KNG_LIB_PATH=lib)
        res, pw, ck, n = run_sampled([sys.executable, "tools/sweep.py", "--launches", str(a.launches), *a.sweep_args.split()], env=env, settle=5.0)
        m = re.search(r"kernel +([0-9.]+) ms +([0-9.]+) MK/s", res.stdout)
        if not m:
            print(f"{name}: FAILED\n{res.stdout[-500:]}\n{res.stderr[-800:]}", flush=True)
            continue
        ms, rate = float(m.group(1)), float(m.group(2))
        print(f"{name:16s}: kernel {ms:7.2f} ms  {rate:8.0f} MK/s   power {pw:.0f} W  sclk {ck} MHz ({n} samples)", flush=True)


if __name__ == "__main__":
    main()
