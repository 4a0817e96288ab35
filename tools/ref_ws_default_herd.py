#!/usr/bin/env python3
"""The REFERENCE program's own -ws / -i path at its default herd: the unmodified host code
(oracle/_ref/kangaroo_hip = reference sources + our GPUEngine) creates 2^23 kangaroos with its own CreateHerd, walks them on
the engine, saves them through GPUEngine::GetKangaroos into 3 x 2^23 `Int`s (Kangaroo.cpp:556-561, Backup.cpp:525-546),
checks the file (-winfo, -wcheck), and restores it through SetKangaroos (-i).  Prints seconds for every stage.
usage (GPU box): python tools/ref_ws_default_herd.py [--dp 20] [--wi 15]"""
import argparse
import os
import re
import select
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the bench configuration: range, key)
from kangaroo_amd import hostlib as hl  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--dp", type=int, default=20, help="DP bits: large enough for -wcheck (one CPU scalar multiplication per stored point) to finish in seconds")
ap.add_argument("--wi", type=int, default=15, help="save interval of the reference program, seconds")
a = ap.parse_args()
exe = os.path.join(ROOT, "oracle", "_ref", "kangaroo_hip")
assert os.path.exists(exe), exe


def run_until(cmd, pattern, count, max_seconds):
    """run unbuffered until `pattern` has appeared `count` times; returns (output, seconds)"""
    if shutil.which("stdbuf"):
        cmd = ["stdbuf", "-o0", "-e0"] + cmd
    t0 = time.time()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    fd = proc.stdout.fileno()
    buf = b""
    while time.time() - t0 < max_seconds:
        r, _, _ = select.select([fd], [], [], 0.5)
        if r:
            chunk = os.read(fd, 65536)
            if not chunk:
                break
            buf += chunk
        if len(re.findall(pattern, buf)) >= count or proc.poll() is not None:
            break
    dt = time.time() - t0
    proc.kill()
    proc.wait()
    return buf.decode(errors="replace"), dt


_, kx, ky = hl.pubkey(bench.KEY)
pub = ("02" if ky % 2 == 0 else "03") + f"{kx:064X}"
with tempfile.TemporaryDirectory() as td:
    cfg = os.path.join(td, "in80.txt")
    open(cfg, "w").write(f"{bench.RANGE_START:064X}\n{bench.RANGE_START + (1 << bench.RANGE_POWER) - 1:064X}\n{pub}\n")
    f1, f2 = os.path.join(td, "a.work"), os.path.join(td, "b.work")
    print(f"# reference program on the engine, default grid, 80-bit bench key, -d {a.dp} -ws -wi {a.wi}", flush=True)
    out, dt = run_until([exe, "-t", "0", "-gpu", "-d", str(a.dp), "-ws", "-w", f1, "-wi", str(a.wi), cfg], rb"done \[", 1, 240)
    m = re.search(r"SaveWork: .*?done \[([^\]]*)\]", out, re.S)
    print(f"create herd (reference CreateHerd) + walk + first save: {dt:.1f} s; the save line: {m.group(0).splitlines()[-1] if m else out[-600:]!r}", flush=True)
    grid = re.search(r"Grid\((\d+)x(\d+)\)", out)
    nk = int(grid.group(1)) * int(grid.group(2)) * 128 if grid else 0
    print(f"engine banner: {re.search(r'GPU: .*', out).group(0) if re.search(r'GPU: .*', out) else '?'}; herd = {nk} kangaroos; file {os.path.getsize(f1) / 1e6:.1f} MB", flush=True)
    t0 = time.time()
    info = subprocess.run([exe, "-winfo", f1], capture_output=True, text=True, timeout=300).stdout
    print(f"-winfo ({time.time() - t0:.1f} s): " + " | ".join(l.strip() for l in info.splitlines() if re.search(r"Kangaroos|Count|DP bits|Time", l)), flush=True)
    assert re.search(r"Kangaroos\s*:\s*%d\b" % nk, info), info
    t0 = time.time()
    chk = subprocess.run([exe, "-t", "16", "-wcheck", f1], capture_output=True, text=True, timeout=900).stdout
    print(f"-wcheck ({time.time() - t0:.1f} s): {chk.strip().splitlines()[-1] if chk.strip() else '?'}", flush=True)
    assert "100.000% OK" in chk, chk[-400:]
    out2, dt2 = run_until([exe, "-t", "0", "-gpu", "-d", str(a.dp), "-i", f1, "-ws", "-w", f2, "-wi", str(a.wi)], rb"done \[", 1, 240)
    lw = re.search(r"LoadWork:.*", out2)
    fk = re.search(r"Fetch kangaroos.*|FetchKangaroos.*", out2)
    print(f"-i restore (LoadWork + SetKangaroos of {nk} kangaroos) + walk + next save: {dt2:.1f} s; {lw.group(0) if lw else ''} {fk.group(0) if fk else ''}", flush=True)
    assert out2.count("done [") >= 1, out2[-800:]
    info2 = subprocess.run([exe, "-winfo", f2], capture_output=True, text=True, timeout=300).stdout
    c1 = int(re.search(r"Count\s*:\s*(\d+)", info).group(1))
    c2 = int(re.search(r"Count\s*:\s*(\d+)", info2).group(1))
    print(f"count in the first file {c1} = 2^{__import__('math').log2(c1):.2f}, after the restored run {c2} = 2^{__import__('math').log2(c2):.2f} (continues, not restarts: {c2 > c1})", flush=True)
    assert c2 > c1
print("OK", flush=True)
