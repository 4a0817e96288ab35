#!/usr/bin/env python3
"""What a `-ws -wi N` save (Backup.cpp:449-563) and a `-i` restore (Backup.cpp:211-231, :286-364) cost the GPU of the
reference PROGRAM, per binary under oracle/_ref/ (VERDICT r5 item 1a).

For every binary: (A) a run without a work file = the rate the GPU delivers when nothing parks it, (B) the same run with
`-ws -w f -wi <wi>`: every output line is timestamped as it arrives, so the "SaveWork: ... done" bracket has millisecond
resolution (the program's own "[03s]" is rounded to seconds), and the GPU-idle seconds per save follow from the jumps the run
did NOT do:  idle = wall x (1 - rate_B / rate_A) / saves  (the Count column, exact for binaries that print KNG_STATS),
(C) `-i f`: seconds from process start to the "2^23.00 kangaroos [..s]" line of SolveKeyGPU (LoadWork + FetchWalks +
SetKangaroos), then one more save, whose Count must continue from the file's.

usage (GPU box): python tools/ref_program_save_cost.py [--exe kangaroo_mi355x,kangaroo_hip] [--bits 80,125] [--dp 16]
                        [--wi 20] [--seconds 70] [--env KNG_REF_SAVE=1]
"""
import argparse
import math
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
import bench  # noqa: E402
from kangaroo_amd import hostlib as hl  # noqa: E402


def timed_run(cmd, seconds, env=None, stop_after=None):
    """run unbuffered for `seconds` (or until `stop_after` = (regex, count) is satisfied); -> [(t, line)], wall"""
    if shutil.which("stdbuf"):
        cmd = ["stdbuf", "-o0", "-e0"] + cmd
    t0 = time.time()
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env)
    fd = proc.stdout.fileno()
    pend, lines, blob = b"", [], b""
    marks, scanned = [], 0  # arrival time of every "SaveWork:" (it shares its line with the dots and "done": line times would hide the bracket)
    while time.time() - t0 < seconds:
        r, _, _ = select.select([fd], [], [], 0.2)
        if r:
            chunk = os.read(fd, 65536)
            if not chunk:
                break
            now = time.time() - t0
            blob += chunk
            while True:
                at = blob.find(b"SaveWork:", scanned)
                if at < 0:
                    scanned = max(0, len(blob) - 9)
                    break
                marks.append(now)
                scanned = at + 9
            pend += chunk
            parts = re.split(rb"[\r\n]", pend)
            pend = parts.pop()
            lines += [(now, p.decode(errors="replace")) for p in parts if p.strip()]
        if proc.poll() is not None:
            break
        if stop_after and len(re.findall(stop_after[0], blob)) >= stop_after[1]:
            break
    wall = time.time() - t0
    if proc.poll() is None:
        proc.terminate()  # SIGTERM: the program has no handler, it just dies; the work file on disk is the last complete one
        try:
            proc.wait(5)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.wait()
    if pend.strip():
        lines.append((wall, pend.decode(errors="replace")))
    timed_run.marks = marks
    return lines, wall


def count_rate(lines, t_lo=12.0):
    """jumps/s from the Count column of the status line, first sample after t_lo to the last one"""
    pts = [(t, 2.0 ** float(m.group(1))) for t, l in lines for m in [re.search(r"\[Count 2\^([0-9.]+)\]", l)] if m]
    pts = [p for p in pts if p[0] >= t_lo]
    if len(pts) < 3:
        return None
    return (pts[-1][1] - pts[0][1]) / (pts[-1][0] - pts[0][0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--exe", default="kangaroo_mi355x,kangaroo_hip")
    ap.add_argument("--bits", default="80,125")
    ap.add_argument("--dp", type=int, default=16)
    ap.add_argument("--wi", type=int, default=20)
    ap.add_argument("--seconds", type=float, default=70)
    ap.add_argument("--env", default="", help="comma list of NAME=VALUE for the program")
    ap.add_argument("--grid", default="", help="-g value; default = the program's own (2*CU x 128)")
    a = ap.parse_args()
    env = dict(os.environ, KNG_STATS="1")
    for kv in filter(None, a.env.split(",")):
        k, v = kv.split("=", 1)
        env[k] = v
    print(f"# -d {a.dp} -wi {a.wi}, {a.seconds:.0f} s per run, env {a.env or '-'}, grid {a.grid or 'default'}", flush=True)
    for bits in (int(b) for b in a.bits.split(",")):
        start = bench.RANGE_START if bits == 80 else 1 << 200  # any interval of that width
        key = start + ((0xC0FFEE123456789ABCD * 0x9E3779B97F4A7C15F39CC0605CEDC835) % (1 << bits) | (1 << (bits - 2)))
        _, kx, ky = hl.pubkey(key)
        pub = ("02" if ky % 2 == 0 else "03") + f"{kx:064X}"
        for exe_name in a.exe.split(","):
            exe = os.path.join(ROOT, "oracle", "_ref", exe_name)
            if not os.path.exists(exe):
                print(f"{exe_name}: not built", flush=True)
                continue
            with tempfile.TemporaryDirectory() as td:
                cfg = os.path.join(td, "in.txt")
                open(cfg, "w").write(f"{start:064X}\n{start + (1 << bits) - 1:064X}\n{pub}\n")
                base = [exe, "-t", "0", "-gpu", "-d", str(a.dp)] + (["-g", a.grid] if a.grid else [])
                print(f"== {exe_name}, {bits}-bit range", flush=True)
                # (A) no work file
                la, wa = timed_run(base + [cfg], 40, env)
                ra = count_rate(la)
                print(f"  A  no work file: {ra / 1e9 if ra else float('nan'):.3f} GK/s by the Count column over {wa:.0f} s", flush=True)
                # (B) periodic saves
                f1 = os.path.join(td, "a.work")
                lb, wb = timed_run(base + ["-ws", "-w", f1, "-wi", str(a.wi), cfg], a.seconds, env)
                rb = count_rate(lb)
                starts = list(timed_run.marks)
                dones = [(t, l) for t, l in lb if re.search(r"done \[[0-9.]+ MB\]", l)]
                brackets = [d[0] - s for s, d in zip(starts, dones)]
                nsave = len(dones)
                for (t, l) in dones:
                    print(f"     t={t:7.3f} s  {l.strip()[-70:]}", flush=True)
                size = os.path.getsize(f1) / 1e6 if os.path.exists(f1) else 0
                idle = wb * (1 - rb / ra) / nsave if (ra and rb and nsave) else float("nan")
                print(f"  B  -ws -wi {a.wi}: {nsave} saves of {size:.1f} MB in {wb:.0f} s; 'SaveWork:' -> 'done' bracket "
                      f"{', '.join('%.3f' % b for b in brackets)} s; rate {rb / 1e9 if rb else float('nan'):.3f} GK/s -> "
                      f"GPU idle {idle:.3f} s per save (= wall x (1 - B/A) / saves)", flush=True)
                for t, l in lb:
                    if "SolveKeyGPU_kng GPU#" in l or "parked" in l or "SaveWork_kng" in l:
                        print("     " + l.strip()[:400], flush=True)
                if not os.path.exists(f1):
                    continue
                # (C) restore, one more save
                f2 = os.path.join(td, "b.work")
                lc, wc = timed_run(base + ["-i", f1, "-ws", "-w", f2, "-wi", str(a.wi), cfg], 180, env, stop_after=(rb"done \[[0-9.]+ MB\]", 1))
                t_walk = next((t for t, l in lc if re.search(r"SolveKeyGPU Thread GPU#\d+: 2\^", l)), None)
                for t, l in lc:
                    if re.search(r"LoadWork:|FectchKangaroos:|Fetch kangaroos|SolveKeyGPU Thread GPU#\d+: 2\^|restor", l):
                        print(f"     t={t:7.3f} s  {l.strip()[:200]}", flush=True)
                print(f"  C  -i: process start -> kangaroos walking {t_walk if t_walk is None else round(t_walk, 3)} s "
                      f"(includes ~1 s of program start and LoadTable)", flush=True)
                if os.path.exists(f2):
                    i1 = subprocess.run([exe, "-winfo", f1], capture_output=True, text=True, timeout=300).stdout
                    i2 = subprocess.run([exe, "-winfo", f2], capture_output=True, text=True, timeout=300).stdout
                    c1 = int(re.search(r"Count\s*:\s*(\d+)", i1).group(1))
                    c2 = int(re.search(r"Count\s*:\s*(\d+)", i2).group(1))
                    k2 = re.search(r"Kangaroos\s*:\s*(\d+)", i2).group(1)
                    print(f"     count 2^{math.log2(c1):.3f} -> 2^{math.log2(c2):.3f} after the restored run ({k2} kangaroos in the file): "
                          f"{'continues' if c2 > c1 else 'DOES NOT CONTINUE'}", flush=True)
    print("OK", flush=True)


if __name__ == "__main__":
    main()
