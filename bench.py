#!/usr/bin/env python3
"""bench.py -- kangaroo jumps/s of the MI355X jump engine on BASELINE.json's throughput config.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2] / SURVEY.md 8d config 3): 80-bit range, one synthetic key with a
known answer, reference-default grid 2*CU x 128 threads x 128 kangaroos = 2^23 kangaroos per GPU,
auto DP (Kangaroo.cpp:980-988 -> 14 at one GPU), reference jump table (seed 0x600DCAFE).
One "step" = one engine launch = NB_RUN = 64 jumps of every kangaroo (Kangaroo.cpp:574-575), with
the previous launch's distinguished points drained to the host while the next one runs.
Herds are independent per GPU (no collective on the data path; Kangaroo.cpp:1041-1047) -> weak scaling.

N > 1 (torchrun or plain `--gpus N`): ONE process -- rank 0 -- drives all N devices through the host pipeline
(kangaroo_amd/host/kng_solver: a thread per GPU, one shared distinguished-point table, as the reference's
SolveKeyGPU threads share its HashTable); the other ranks only take part in the barriers.  The figure is then the
wall-clock rate of the whole job including every DP insert, with the per-GPU kernel times beside it.

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events on the engine's own
stream; `cpu_baseline` (N=1 only) times the reference's SolveKeyCPU binary (oracle/_ref/kangaroo_cpu,
built from the reference sources) on this host, or the oracle port when that binary is absent.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RANGE_POWER = 80
RANGE_START = int("B60E83280258A40F9CDF1649744D730D6E939DE92A2B" + "0" * 20, 16)
KEY = RANGE_START + 0xC0FFEE123456789ABCD  # 80-bit offset, answer known
ALG_BYTES_PER_JUMP = 160  # read + write of x(32) y(32) d(16): SURVEY.md 8d
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # MI355X_MICROARCH.md: measured copy ceiling
# BASELINE configs[3]: puzzle #110 (reference puzzle32.txt:6-9), interval [2^109, 2^110 - 1] -> rangePower 109, DP 25
P110_START, P110_END = 0x2000000000000000000000000000, 0x3FFFFFFFFFFFFFFFFFFFFFFFFFFF
P110_PUB = "0309976BA5570966BF889196B7FDF5A0F9A1E9AB340556EC29F8BB60599616167D"


def log(msg):
    print(msg, file=sys.stderr, flush=True)


def effective_cpus() -> float:
    """CPUs this process may really use: hardware threads cut by the affinity mask and the cgroup CPU quota (the GPU boxes
    show 256 hardware threads under a quota of 16 CPUs: threads beyond that only take turns)."""
    n = float(os.cpu_count() or 1)
    try:
        n = min(n, float(len(os.sched_getaffinity(0))))
    except Exception:
        pass
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, float(q) / float(period))
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, q / period)
        except Exception:
            pass
    return max(1.0, n)


def cpu_baseline(seconds: float = 20.0) -> dict:
    """Reference SolveKeyCPU path on this host's cores (bounded sample)."""
    import kangaroo_amd.hostlib as hl

    hw = os.cpu_count() or 1
    cores = max(1, int(effective_cpus() + 0.5))  # one thread per CPU the process may use (not per hardware thread it can see)
    ref = os.path.join(ROOT, "oracle", "_ref", "kangaroo_cpu")
    _, kx, ky = hl.pubkey(KEY)
    pub = ("02" if ky % 2 == 0 else "03") + f"{kx:064X}"
    if os.path.exists(ref):
        import select
        import shutil

        with tempfile.TemporaryDirectory() as td:
            cfg = os.path.join(td, "in80.txt")
            with open(cfg, "w") as f:
                f.write(f"{RANGE_START:064X}\n{RANGE_START + (1 << RANGE_POWER) - 1:064X}\n{pub}\n")
            # the status line is printf("\r[%.2f MK/s]...") without fflush (Thread.cpp:306-314):
            # unbuffer its stdout with stdbuf (LD_PRELOAD), read through a pipe, stop it after `seconds`
            cmd = [ref, "-t", str(cores), cfg]
            if shutil.which("stdbuf"):
                cmd = ["stdbuf", "-o0", "-e0"] + cmd
            proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            fd = proc.stdout.fileno()
            buf = b""
            t0 = time.time()
            while time.time() - t0 < seconds:
                r, _, _ = select.select([fd], [], [], 0.5)
                if r:
                    chunk = os.read(fd, 65536)
                    if not chunk:
                        break
                    buf += chunk
                if proc.poll() is not None:
                    break
            proc.kill()
            proc.wait()
        rates = [float(m) for m in re.findall(rb"\[([0-9.]+) MK/s\]\[GPU", buf)]
        steady = rates[3:] if len(rates) > 6 else rates  # 8-sample moving average: skip the ramp
        if steady:
            steady.sort()
            return {"value": steady[len(steady) // 2], "unit": "MK/s", "cores": cores, "kind": "reference",
                    "sample": f"reference kangaroo -t {cores} on the same 80-bit input for {seconds:.0f} s, median of "
                              f"{len(steady)} status samples; {hw} hardware threads visible, CPU quota of this process {effective_cpus():.1f}"}
    # fallback: the oracle's batched walk (single thread)
    import numpy as np

    from oracle import load_oracle

    orc = load_oracle()
    n = 1024  # CPU_GRP_SIZE, Kangaroo.cpp:68
    x, y, d, woff = hl.create_herd(n, RANGE_POWER, (kx, ky), seed=99, nthreads=1)
    jd, jx, jy, _ = orc.jump_table(RANGE_POWER)
    dd = hl.to_device_distances(d, woff)
    steps = 0
    t0 = time.time()
    while time.time() - t0 < min(seconds, 10.0):
        orc.walk(x, y, dd, 64, jd, jx, jy, hl.dp_mask(14), dp_cap=0)
        steps += 64
    el = time.time() - t0
    return {"value": n * steps / el / 1e6, "unit": "MK/s", "cores": 1, "kind": "port",
            "sample": f"oracle/kng_oracle.c orc_walk, 1024 kangaroos x {steps} jumps, 1 thread"}


def _kernel_name(eng) -> str:
    """Name of the walk kernel the engine launches with its current options (as rocprofv3 prints it).  `eng`: anything
    with get_option(key) -- a GPUEngine, or one GPU of a Solver."""
    share, ds, am = eng.get_option("share"), eng.get_option("dsplit"), eng.get_option("asm")
    if am == 2:
        return "kng_walk_valu_only_kernel"
    return f"kng_walk_share_kernel<{share}, {'true' if ds else 'false'}, {'true' if am else 'false'}>"


def _roofline(kernel, kms, n, nb_run, group, note=None, sustained_ms=None, step_ms=None, measured=None):
    """`achieved` / `frac`: algorithmic bytes per launch over the time per launch.  The time is `step_ms` -- the wall-clock
    ms_per_step of the timed region, what the driver's own clock sees -- when given, with the HIP-event
    duration of the kernel alone beside it (`achieved_kernel`, `frac_kernel`: kernel_ms <= ms_per_step)."""
    alg = n * nb_run * ALG_BYTES_PER_JUMP
    t_ms = step_ms if step_ms else kms
    achieved = alg / (t_ms * 1e-3) / 1e9
    if measured and measured.get("hbm_bytes_per_launch"):
        traffic, tsrc = measured["hbm_bytes_per_launch"], measured["source"]
    else:
        traffic, tsrc = _recorded_traffic(n, group, kernel)
        if measured and measured.get("error"):
            tsrc = f"live PMC passes failed ({measured['error']}); " + (tsrc or "")
    r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
         "time_base": "ms_per_step of the timed region (wall clock, barrier to barrier)" if step_ms else "HIP-event kernel duration",
         "frac_achievable": round(achieved / HBM_ACHIEVABLE_GBS, 4), "peak_achievable": HBM_ACHIEVABLE_GBS,
         "achieved_kernel": round(alg / (kms * 1e-3) / 1e9, 1), "frac_kernel": round(alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
         "traffic": traffic, "traffic_source": tsrc, "kernel": kernel, "kernel_ms": round(kms, 3), "alg_bytes_per_launch": alg}
    if traffic:
        r["traffic_gbs"] = round(traffic / (kms * 1e-3) / 1e9, 1)
        r["traffic_over_algorithmic"] = round(traffic / alg, 3)
    if measured:
        r["traffic_passes"] = {k: v for k, v in measured.items() if k not in ("hbm_bytes_per_launch", "source")}
    if sustained_ms:
        r["frac_sustained"] = round(alg / (sustained_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        r["sustained_kernel_ms"] = round(sustained_ms, 3)
    if note:
        r["note"] = note
    return r


def _rocprofv3():
    import shutil

    for c in (shutil.which("rocprofv3"), "/opt/rocm/bin/rocprofv3"):
        if c and os.path.exists(c):
            return c
    return None


def live_traffic(kernel, n, group, grid, steps=3) -> dict:
    """HBM bytes per launch of the walk kernel MEASURED IN THIS RUN: two rocprofv3 counter passes over a
    child of this very script (`--pmc-child`: same engine, herd, grid, DP; 1 warm-up + `steps` launches, nothing else), one
    per counter -- FETCH_SIZE and WRITE_SIZE do not fit one pass (MI355X_MICROARCH.md, TCC slots) -- counters only, no
    trace domain.  Corrections as the guide prescribes: values are KiB; FETCH_SIZE reports half the bytes of wide coalesced
    reads on gfx950, doubled.  bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, mean over the kernel's dispatches."""
    import csv
    import glob

    exe = _rocprofv3()
    if not exe:
        return {"error": "rocprofv3 not found"}
    out = {"tool": exe, "launches_per_pass": steps}
    t0 = time.time()
    means = {}
    env = dict(os.environ, TMPDIR="/tmp")
    env.pop("RANK", None), env.pop("WORLD_SIZE", None), env.pop("LOCAL_RANK", None)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", td, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__),
                   "--pmc-child", "--steps", str(steps), "--warmup", "1", "--grid", f"{grid[0]},{grid[1]}"] + (["--group", str(group)] if group else [])
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
            except Exception as e:  # noqa: BLE001
                return dict(out, error=f"{counter} pass: {e}")
            files = glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return dict(out, error=f"{counter} pass: rc {p.returncode}, {len(files)} csv; {(p.stderr or '')[-300:]}")
            vals = []
            with open(files[0]) as f:
                for row in csv.DictReader(f):
                    if row.get("Counter_Name") == counter and kernel.split("<")[0] in row.get("Kernel_Name", "") and kernel.split("<")[1].rstrip(">").replace(" ", "") in row.get("Kernel_Name", "").replace(" ", ""):
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return dict(out, error=f"{counter} pass: no dispatch of {kernel} in the csv")
            means[counter] = sum(vals) / len(vals)
            out[counter.lower() + "_kib_mean"] = round(means[counter], 1)
            out[counter.lower() + "_dispatches"] = len(vals)
    b = (2.0 * means["FETCH_SIZE"] + means["WRITE_SIZE"]) * 1024.0
    out["seconds"] = round(time.time() - t0, 1)
    out["hbm_bytes_per_launch"] = int(round(b))
    out["bytes_per_jump"] = round(b / (n * 64), 1)
    out["read_bytes_per_jump"] = round(2.0 * means["FETCH_SIZE"] * 1024.0 / (n * 64), 1)
    out["write_bytes_per_jump"] = round(means["WRITE_SIZE"] * 1024.0 / (n * 64), 1)
    out["source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `bench.py --pmc-child` "
                     f"({steps} launches each, same engine / herd / grid), KiB -> bytes, FETCH doubled per the gfx950 note of MI355X_MICROARCH.md")
    return out


def alu_ceiling(k, hl, dev, gx, gy, dp, seed, steps=8) -> dict:
    """the integer-ALU ceiling next to the HBM roofline -- measured in this run with the headline kernel's own
    instruction stream: engine option "asm" 2 launches kng_walk_valu_only_kernel, the scheduled loop with the global and LDS
    accesses of its per-kangaroo loop and the flag collection left out (tools/gen_walk_asm.py VALU_ONLY; inversion tree, grid,
    occupancy unchanged).  What the kernel would do if memory cost neither cycles nor power."""
    r = _timed_engine(k, hl, dev, gx, gy, RANGE_POWER, None, dp, steps, 2, seed, asm=2)
    rate = r["value"]
    return {"value_mks": rate, "kernel_ms": r["kernel_ms"], "as_frac_of_hbm_roofline": round(rate * 1e6 * ALG_BYTES_PER_JUMP / 1e9 / HBM_PEAK_GBS, 4),
            "kernel": "kng_walk_valu_only_kernel", "launches": steps,
            "provenance": "measured in this run: the scheduled walk loop without its memory instructions (engine option \"asm\" 2; 1025 VALU "
                          "instructions per kangaroo-jump of which 410 v_mad_u64_u32, same inversion tree / grid / 2 waves per SIMD); wrong results on purpose"}


# ---- second bound: the package power cap.  Inputs measured on this chip in earlier rounds:
#   profiles/r01_instr_energy.txt   342 W with nothing but s_nop on every SIMD (static + clocks); per wave64 instruction
#                                   v_mad_u64_u32 1.38 nJ, carry / VOP3 ops 0.75 nJ, moves 0.36 nJ
#   profiles/r02_memory_energy.txt  ~100 pJ per byte moved through L2 / fabric / HBM (copy 102, read 93, write 115)
#   tools/gen_walk_asm.py           static count of the scheduled loop per kangaroo-jump: 410 MAD, 475 other slow, 140 fast VALU
P_STATIC_W = 342.0
E_VALU_NJ_PER_WAVE_JUMP = 410 * 1.38 + 475 * 0.75 + 140 * 0.36
E_MEM_NJ_PER_BYTE = 0.100
DESIGN_BYTES_PER_JUMP = 208.0


def _power_bound(power_summary, bytes_per_jump, rate_mks):
    """roofline.power_bound: the jump rate at which the kernel's dynamic energy per jump uses up (P_cap - P_static)."""
    dev = (power_summary.get("devices") or [{}])[0] if power_summary.get("available") else {}
    cap = dev.get("power_cap_w")
    e_dyn = E_VALU_NJ_PER_WAVE_JUMP / 64.0 + bytes_per_jump * E_MEM_NJ_PER_BYTE  # nJ per kangaroo-jump
    out = {"formula": "(P_cap - P_static) / E_dyn_per_jump", "p_static_w": P_STATIC_W, "e_dyn_nj_per_jump": round(e_dyn, 2),
           "e_valu_nj_per_jump": round(E_VALU_NJ_PER_WAVE_JUMP / 64.0, 2), "e_mem_nj_per_jump": round(bytes_per_jump * E_MEM_NJ_PER_BYTE, 2),
           "bytes_per_jump": bytes_per_jump, "p_cap_w": cap,
           # where each constant comes from -- none is measured in this run, each was probed once on ONE box of the pool
           "inputs": {"p_static_w": "342 W: round 1, tools/archive/instr_power.sh on one MI355X of the pool (profiles/r01_instr_energy.txt, s_nop on every SIMD)",
                      "valu_nj_per_wave_instr": "MAD 1.38 / VOP3+carry 0.75 / move 0.36 nJ: same round-1 probe, same box",
                      "mem_nj_per_byte": "0.100 nJ/B: round 2, tools/archive/mem_power_probe.hip on another box (profiles/r02_memory_energy.txt: copy 102, read 93, write 115 pJ/B)",
                      "instr_per_jump": "410 MAD + 475 slow + 140 fast VALU: static count of the loop tools/gen_walk_asm.py prints (this tree)",
                      "bytes_per_jump": "roofline.traffic / jumps when a recorded figure applies, else the design figure 208",
                      "p_cap_w": "rocm_smi power cap of the device of THIS run", "measured_power_w": "energy counter / samples of THIS run",
                      "caveat": "boxes of the pool differ by a few per cent in leakage and clocks; the bound is a model, good to ~5 %"}}
    if cap:
        out["value_mks"] = round((cap - P_STATIC_W) / (e_dyn * 1e-9) / 1e6, 0)
    # the energy accumulator over the window where the device has one (exact average), else the median of the samples
    pw = dev.get("power_w_from_energy_counter") or (dev.get("power_w") or {}).get("median")
    if pw and rate_mks:
        out["measured_power_w"] = pw
        out["measured_over"] = f"{power_summary.get('window_s')} s window"

        out["measured_nj_per_jump"] = round(pw / (rate_mks * 1e6) * 1e9, 2)             # everything, static included
        out["value_at_measured_power_mks"] = round((pw - P_STATIC_W) / (e_dyn * 1e-9) / 1e6, 0)
        out["frac_of_bound_at_measured_power"] = round(rate_mks / out["value_at_measured_power_mks"], 3)
    return out


def _decompress(pub_hex):
    P = 2**256 - 0x1000003D1
    x = int(pub_hex[2:], 16)
    y = pow((x * x * x + 7) % P, (P + 1) // 4, P)
    return x, (y if (y & 1) == (int(pub_hex[:2], 16) & 1) else P - y)


def _shifted_key(hl, key_xy, range_start):
    """keyToSearch = K - start*G (Kangaroo.cpp:892-909)"""
    if range_start == 0:
        return key_xy
    P = 2**256 - 0x1000003D1
    _, sx, sy = hl.pubkey(range_start)
    _, x, y = hl.point_add(key_xy, (sx, P - sy))
    return x, y


KERNEL_SOURCES = ("kng_engine.hip", "kng_field.h", "kng_mul32.h", "kng_mulasm.h", "kng_walk_asm.h", "kng_modinv.h")


def kernel_source_blobs() -> dict:
    """`git hash-object` of the files the walk kernel is compiled from (computed here: the GPU box has no .git)"""
    import hashlib

    out = {}
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "kangaroo_amd", "csrc", name), "rb") as f:
            data = f.read()
        out[name] = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    return out


def _recorded_traffic(n, group, kernel):
    """HBM bytes per launch from the last PMC pass (profiles/traffic.json, written by tools/gpu_round.sh).  Quoted only
    for the kernel and geometry it was measured on AND only while the kernel's sources are byte-identical to the ones it
    was measured on (git blob ids recorded with the figure); otherwise traffic is null and the reason is given."""
    tfile = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(tfile) as f:
            tj = json.load(f)
    except Exception as e:
        return None, f"no recorded figure ({e})"
    if not (tj.get("kangaroos") == n and tj.get("group") == group and tj.get("kernel") == kernel):
        return None, f"recorded for {tj.get('kernel')} at {tj.get('kangaroos')} kangaroos / group {tj.get('group')}, not for this run"
    now, then = kernel_source_blobs(), tj.get("source_blobs") or {}
    changed = sorted(k for k in now if then.get(k) != now[k])
    if changed:
        return None, f"stale: {', '.join(changed)} changed since the PMC pass ({tj.get('source')})"
    # say what it is: a figure RECORDED by an earlier PMC pass of the builder (rocprofv3 cannot wrap the driver's bench run),
    # tied to this run only through the byte-identity of the kernel sources -- not bytes counted during this run
    return tj.get("hbm_bytes_per_launch"), (f"RECORDED, not measured in this run: {tj.get('source')}; quoted because kernel, herd, group match and all "
                                            f"{len(now)} kernel source files have the git blob ids of that pass (profiles/traffic.json)")


def _timed_engine(k, hl, dev, gx, gy, range_power, key_xy, dp, steps, warmup, seed, **opts):
    """W + K launches of one engine on a device-built herd; returns kernel name, mean kernel ms, geometry."""
    import numpy as np

    n = gx * gy * k.KNG_GRP_SIZE
    jd, jx, jy, _ = hl.jump_table(range_power)
    with k.GPUEngine(gx, gy, dev, max(65536 * 2, 2 * ((n * k.KNG_NB_RUN) >> dp)), **opts) as eng:
        eng.SetParams(hl.dp_mask(dp), jd, jx, jy)
        eng.CreateHerdOnDevice(range_power, key_xy, seed=seed)
        ms = []
        for i in range(warmup + steps):
            eng.callKernel()
            eng.wait()
            eng.drain(raw=True)
            if i >= warmup:
                ms.append(eng.last_kernel_ms())
        kms = float(np.mean(ms))
        return {"kernel": _kernel_name(eng), "kernel_ms": round(kms, 3), "kangaroos": n, "group": eng.get_option("group"),
                "lanes": eng.get_option("lanes"), "value": round(n * k.KNG_NB_RUN / (kms * 1e-3) / 1e6, 1), "unit": "MK/s",
                "frac": round(n * k.KNG_NB_RUN * ALG_BYTES_PER_JUMP / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "launches": steps}


def secondary_lines(k, hl, dev, gx, gy) -> list:
    """The other kernels and herds BASELINE.json names, next to (never instead of) the headline: kernel-rate only."""
    out = []
    try:  # configs[3]: 109-bit range, DP 25: jump distances ~2^54 -- since round 3 still low-word streaming (carries handled in the loop)
        r = _timed_engine(k, hl, dev, gx, gy, 109, _shifted_key(hl, _decompress(P110_PUB), P110_START), 25, 10, 2, 0x110)
        out.append(dict(r, name="configs[3] puzzle #110: 109-bit range, DP 25, default grid", bytes_per_jump_design=208 if r["kernel"].split(",")[1].strip() == "true" else 224))
    except Exception as e:
        out.append({"name": "configs[3]", "error": str(e)})
    try:  # configs[4]'s table: 125-bit range (jump distances ~2^62): both distance words stream (the non-dsplit kernel)
        _, k125x, k125y = hl.pubkey((1 << 124) + 0xC0FFEE123456789ABCD)
        r = _timed_engine(k, hl, dev, gx, gy, 125, (k125x, k125y), hl.suggest_dp(125, gx * gy * 128), 10, 2, 0x125)
        out.append(dict(r, name="configs[4] table: 125-bit range, auto DP, default grid", bytes_per_jump_design=224))
    except Exception as e:
        out.append({"name": "configs[4] table", "error": str(e)})
    try:  # configs[2] read literally: herd = 2*CU x 128 kangaroos = grid 2*CU x 1
        _, kx, ky = hl.pubkey(KEY)
        r = _timed_engine(k, hl, dev, gx, 1, RANGE_POWER, _shifted_key(hl, (kx, ky), RANGE_START), hl.suggest_dp(RANGE_POWER, gx * 128), 10, 2, 0x65536)
        out.append(dict(r, name="configs[2] literal herd: 2*CU x 128 = %d kangaroos, 80-bit range" % (gx * 128)))
    except Exception as e:
        out.append({"name": "configs[2] literal herd", "error": str(e)})
    return out


def bench_multi(args, ranks, n_gpus):
    """N > 1: rank 0 runs the whole job in one process (a host thread per GPU, ONE distinguished-point table)."""
    import numpy as np

    from kangaroo_amd.dist import timed_on_rank0

    from kangaroo_amd.dist import RankFailure

    out = None
    job = {}
    prep_error = None
    s = None
    if ranks.rank == 0:
      try:  # a failure here must reach the other ranks (they are about to wait in a barrier): all_ok() below
        import kangaroo_amd as k
        import kangaroo_amd.hostlib as hl
        from kangaroo_amd import solver as sv

        k.load_library()
        devices = tuple(int(v) for v in args.devices.split(",")) if args.devices else tuple(range(n_gpus))
        if len(devices) != n_gpus or max(devices) >= k.device_count():
            raise RuntimeError(f"--gpus {n_gpus}: rank 0 drives devices {devices} but sees only {k.device_count()} HIP device(s) "
                               f"(ROCR_/HIP_VISIBLE_DEVICES per rank?); the job is one process over all GPUs and one shared DP table")
        info = k.device_info(devices[0])
        if args.grid:
            gx, gy = (int(v) for v in args.grid.split(","))
        else:
            gx, gy = k.default_grid(devices[0])
        n = gx * gy * k.KNG_GRP_SIZE
        dp = hl.suggest_dp(RANGE_POWER, n * n_gpus)  # totalRW of Kangaroo.cpp:946-993
        _, kx, ky = hl.pubkey(KEY)
        s = sv.Solver(RANGE_START, RANGE_START + (1 << RANGE_POWER) - 1, (kx, ky), gpus=devices, grid=(gx, gy), dp=dp,
                      seed=0xBEEF, max_launches=args.steps, warmup_launches=args.warmup)
        t0 = time.time()
        s.prepare()  # engines, herds (built on the GPUs), W discarded launches per GPU
        log(f"{n_gpus} x {info['name']}: 2^{np.log2(n):.0f} kangaroos each, dp {dp}; prepared in {time.time() - t0:.1f} s")
        job["run"] = lambda: (s.start(), s.wait(600))
      except BaseException as e:  # noqa: BLE001
        prep_error = e
    if not ranks.all_ok(prep_error is None):
        if prep_error is not None:
            log(f"bench --gpus {n_gpus}: rank 0 could not prepare the job: {prep_error!r}")
        if s is not None:
            s.close()
        ranks.abort()
        raise SystemExit(1)  # EVERY rank leaves non-zero, promptly
    sampler = None
    if ranks.rank == 0:
        from kangaroo_amd.telemetry import GpuSampler

        sampler = GpuSampler(sorted(set(devices)), hz=20.0).start()
    try:
        elapsed = timed_on_rank0(ranks, job.get("run"))
    except RankFailure as e:
        log(f"bench --gpus {n_gpus}: {e}")
        ranks.abort()
        raise SystemExit(1)
    if ranks.rank == 0:
      try:
          power = sampler.stop().summary()
          st = s.stats()
          per_gpu = []
          for g in range(n_gpus):
              gs = s.gpu_stats(g)
              kms = gs["kernel_ms_sum"] / max(1, gs["launches"])
              per_gpu.append({"gpu": g, "device": devices[g], "numa_node": s.gpu_option(g, "numa_node"), "launches": gs["launches"], "kernel_ms": round(kms, 3),
                              "kernel_rate": round(gs["kangaroos"] * k.KNG_NB_RUN / (kms * 1e-3) / 1e6, 1)})
          load = s.consumer_load()
          host = s.host_stats()  # is the host keeping up?  (the threads are still alive: the kernel-side sums come after stop)

          class _Gpu0:  # the walk kernel that actually ran, from the engine of GPU 0 (every GPU gets the same options)
              get_option = staticmethod(lambda key: s.gpu_option(0, key))

          kernel, group = _kernel_name(_Gpu0), s.gpu_option(0, "group")
          aud = None
          try:  # whole-run audit on the devices, outside the clock: every kangaroo of every herd and every table entry
              aud = s.audit(True)
          except Exception as e:  # noqa: BLE001
              aud = {"error": str(e)}
          s.stop()
          after = s.host_stats()
          for key in ("consumer_cpu_s", "consumer_runq_s", "consumer_busy_s", "consumer_nvcsw", "consumer_nivcsw"):
              host[key] = after[key]
          s.close()
          short = [p for p in per_gpu if p["launches"] != args.steps]
          jumps = n_gpus * n * k.KNG_NB_RUN * args.steps
          kms = float(np.mean([p["kernel_ms"] for p in per_gpu]))
          out = {
              "metric": "kangaroo jumps/sec (MK/s)", "value": round(jumps / elapsed / 1e6, 2), "unit": "MK/s", "n_gpus": n_gpus,
              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
              "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
              "config": {
                  "workload": f"80-bit range single key, auto DP {dp} (population of {n_gpus} GPUs), herd {gx}x{gy}x128 = 2^{np.log2(n):.0f} "
                              f"kangaroos/GPU, {k.KNG_NB_RUN} jumps/launch",
                  "range_power": RANGE_POWER, "dp": dp, "grid": [gx, gy], "kangaroos_per_gpu": n, "device": info["name"], "arch": info["arch"],
                  "parallelism": f"independent herds x{n_gpus}: one process, a host thread per GPU, ONE shared host DP table, no collective",
                  "dps_per_step": round(st["dps"] / args.steps, 1), "dps_lost": st["dps_lost"], "table_consumers": len(load),
                  "what_is_timed": "kngs_start .. every GPU finished its K launches AND every distinguished point is in the table",
              },
              "per_gpu": per_gpu,
              "power": {"timed_region": power},  # per device: package power, GFX clock; HIP index -> rocm_smi index by PCI address (telemetry.map_devices)
              # the N-GPU line diagnoses itself: a host that cannot keep up shows here (late_launches > 0, host_ms_max above the
              # kernel time, consumers near 100 % busy or waiting for CPUs) before it shows as value < kernel_rate_sum
              "host": dict(host, points_per_s_offered=round(st["dps"] / elapsed / 1e6, 2), points_unit="M points/s",
                           cpu_ns_per_point=round((host.get("consumer_cpu_s") or 0) / max(1, st["dps"]) * 1e9, 1)),
              "audit": aud,
              "kernel_rate_sum": round(sum(p["kernel_rate"] for p in per_gpu), 1),
              "roofline": _roofline(kernel, kms, n, k.KNG_NB_RUN, group, note="per GPU, mean over GPUs"),
          }
          if short:  # the line still appears, and says it is not a valid measurement
              out["error"] = f"{len(short)} GPU(s) did not finish their {args.steps} launches: {short}"
          out["config"]["control_plane"] = f"{ranks.backend} (CPU): no collective kernel on a measured GPU"
      except BaseException as e:  # noqa: BLE001  -- the driver expects ONE line from rank 0 whatever went wrong after the timed region
        import traceback

        log(traceback.format_exc())
        out = {"metric": "kangaroo jumps/sec (MK/s)", "value": None, "unit": "MK/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
               "higher_is_better": True, "scaling": "weak", "error": f"post-processing failed: {e!r}",
               "elapsed_s": round(elapsed, 4) if isinstance(elapsed, float) else None}
    ranks.close()
    if out is not None:
        print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", default="", help="gridX,gridY (default: reference defaults 2*CU,128)")
    ap.add_argument("--group", type=int, default=0, help="kangaroos per lane (0 = engine default)")
    ap.add_argument("--block", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the two rocprofv3 counter passes that measure roofline.traffic in this run (N=1 only)")
    ap.add_argument("--no-alu-ceiling", action="store_true", help="skip the VALU-only leg (roofline.alu_ceiling, N=1 only)")
    ap.add_argument("--pmc-child", action="store_true", help="internal: warm-up + K launches of the bench engine and nothing else (what the counter passes wrap)")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the end-to-end host-pipeline sample (N=1 only)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary kernel-rate lines (N=1 only)")
    ap.add_argument("--host-herd", action="store_true", help="build the herd on the host and upload it (default: on the GPU)")
    ap.add_argument("--devices", default="", help="N > 1: explicit device list, e.g. 0,0 to exercise the multi-GPU path on one device (default 0..N-1)")
    args = ap.parse_args()

    # torch is plumbing here: rendezvous, barriers and the cross-rank max of the timings (CPU-side, gloo)

    from kangaroo_amd.dist import Ranks, timed_steps, whole_job_rate

    ranks = Ranks()  # gloo: the control plane stays off the GPUs (kangaroo_amd/dist.py)
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world
    if world != args.gpus and world > 1:
        log(f"warning: WORLD_SIZE={world} but --gpus {args.gpus}")
    n_gpus = world if world > 1 else max(1, args.gpus)
    per_rank_note = None
    if n_gpus > 1:
        # The N-GPU job is ONE process over all devices (one shared DP table).  That needs rank 0 to see them all; a launcher
        # that pins each rank to its own device (ROCR_/HIP_VISIBLE_DEVICES per rank) leaves it one.  Decided collectively, so
        # that nobody waits in a barrier for a rank that has left: then every rank walks its own herd on the device it can see
        # and drains its own distinguished points (kernel path identical; the shared table is what that form cannot show).
        seen = -1
        if rank == 0:
            try:
                import kangaroo_amd as k0

                k0.load_library()
                seen = k0.device_count()
            except Exception as e:  # noqa: BLE001 -- reported below, on every rank
                log(f"bench --gpus {n_gpus}: rank 0 cannot load the engine: {e!r}")
        one_process = ranks.all_ok(rank != 0 or bool(args.devices) or seen >= n_gpus)
        if one_process or world == 1:
            return bench_multi(args, ranks, n_gpus)
        per_rank_note = (f"rank 0 sees {seen} of {n_gpus} devices: one process per GPU, every rank drains its own distinguished points "
                         f"(no shared host table in this form)")
        if rank == 0:
            log(f"bench --gpus {n_gpus}: {per_rank_note}")

    import numpy as np

    import kangaroo_amd as k
    import kangaroo_amd.hostlib as hl

    have = 0
    try:
        k.load_library()  # raises when the HIP engine is missing: no fallback
        have = k.device_count()
    except Exception as e:  # noqa: BLE001
        log(f"rank {rank}: {e!r}")
    if not ranks.all_ok(have > 0 and (world == 1 or per_rank_note is not None or have > local_rank)):
        log(f"rank {rank}: sees {have} HIP device(s), needs device {local_rank}")
        ranks.abort()
        raise SystemExit(1)  # every rank leaves, promptly
    dev = local_rank if local_rank < have else local_rank % have  # pinned launchers expose each rank's device as device 0
    info = k.device_info(dev)

    if args.grid:
        gx, gy = (int(v) for v in args.grid.split(","))
    else:
        gx, gy = k.default_grid(dev)  # 2*CU x 128 (GPUEngine.cu:299-303)
    n = gx * gy * k.KNG_GRP_SIZE
    total_rw = ranks.total_kangaroos(n)
    dp = hl.suggest_dp(RANGE_POWER, total_rw)
    jd, jx, jy, javg = hl.jump_table(RANGE_POWER)
    _, kx, ky = hl.pubkey(KEY)
    # keyToSearch is shifted by the range start (Kangaroo.cpp:892-909): K - start*G
    _, sx, sy = hl.pubkey(RANGE_START)
    P = 2**256 - 0x1000003D1
    _, ksx, ksy = hl.point_add((kx, ky), (sx, P - sy))

    opts = {}
    if args.group:
        opts["group"] = args.group
    if args.block:
        opts["block"] = args.block
    # maxFound: the reference hard-codes 131072 (Kangaroo.cpp:523); with many GPUs the auto DP drops and a
    # launch yields more points than that, so size the buffer for 2x the expected count instead of dropping DPs
    expected_dps = (n * k.KNG_NB_RUN) >> dp
    max_found = max(65536 * 2, 2 * expected_dps)
    eng = k.GPUEngine(gx, gy, dev, max_found, **opts)
    eng.SetParams(hl.dp_mask(dp), jd, jx, jy)
    t0 = time.time()
    if args.host_herd:
        # CreateHerd semantics on the host (batched, multi-threaded), then SetKangaroos
        x, y, d_true, woff = hl.create_herd(n, RANGE_POWER, (ksx, ksy), first_type=0, seed=ranks.herd_seed(0xBEEF))
        t_herd = time.time() - t0
        eng.SetWildOffset(woff)
        t0 = time.time()
        eng.SetKangaroos(x, y, hl.to_device_distances(d_true, woff))
        t_up = time.time() - t0
        del x, y, d_true
    else:
        # the herd is built on the GPU (kng_build_herd): valid kangaroos d*G / K + d*G, uniform distances
        eng.CreateHerdOnDevice(RANGE_POWER, (ksx, ksy), seed=ranks.herd_seed(0xBEEF))
        t_herd = time.time() - t0
        t_up = 0.0
    if rank == 0:
        log(f"{eng.deviceName}: 2^{np.log2(n):.2f} kangaroos, dp {dp}, jump avg 2^{javg:.2f}, "
            f"group {eng.get_option('group')} lanes {eng.get_option('lanes')} "
            f"({eng.GetMemory() / 1048576.0:.1f} MB); herd built {'on the host' if args.host_herd else 'on the GPU'} in {t_herd * 1e3:.0f} ms, uploaded in {t_up * 1e3:.0f} ms")

    # warmup: W full steps
    for _ in range(args.warmup):
        eng.callKernel()
        eng.wait()
        eng.drain(raw=True)

    if args.pmc_child:  # what the rocprofv3 counter passes of live_traffic() wrap: the same launches, nothing else, no output
        for _ in range(args.steps):
            eng.callKernel()
            eng.wait()
            eng.drain(raw=True)
        eng.close()
        ranks.close()
        return

    kernel_ms = []
    counts = {"dps": 0, "lost": 0}

    def step(i):
        eng.callKernel()
        # drain the previous step's DPs while this launch runs (DP buffers are double-buffered)
        if i:
            counts["dps"] += len(eng.drain(raw=True))
            counts["lost"] += eng.lastLost
        eng.wait(spin=True)  # GPUEngine::Launch's spinWait (GPUEngine.cu:621-629): no wake-up latency between launches
        kernel_ms.append(eng.last_kernel_ms())
        counts["exits"] = counts.get("exits", 0) + eng.get_option("exact_exits")

    def finish():
        counts["dps"] += len(eng.drain(raw=True))
        counts["lost"] += eng.lastLost

    from kangaroo_amd.telemetry import GpuSampler

    sampler = GpuSampler([dev], hz=50.0).start()  # package power / GFX clock across the timed region (a thread, off the data path)
    elapsed = timed_steps(ranks, step, finish, args.steps)  # barrier+sync, K steps, barrier+sync, max over ranks
    power = sampler.stop().summary()
    dps, lost = counts["dps"], counts["lost"]

    jumps_per_step = n * k.KNG_NB_RUN
    value = whole_job_rate(ranks, jumps_per_step, args.steps, elapsed) / 1e6  # MK/s, whole job
    kms = float(np.mean(kernel_ms))
    kname, kgroup = _kernel_name(eng), eng.get_option("group")
    cfg_now = {"group": kgroup, "lanes": eng.get_option("lanes"), "share": eng.get_option("share"), "asm_loop": eng.get_option("asm"), "mem_mb": eng.GetMemory() / 1048576.0}
    eng.close()  # (the counter passes and the ALU-ceiling leg below bring their own engines)
    lt = None
    if rank == 0 and n_gpus == 1 and not args.no_pmc:
        try:
            lt = live_traffic(kname, n, args.group, (gx, gy))
        except Exception as e:  # noqa: BLE001 -- the line survives a failed counter pass
            lt = {"error": repr(e)}
        log(f"live PMC passes: {lt}")
    roof = _roofline(kname, kms, n, k.KNG_NB_RUN, kgroup, step_ms=elapsed / args.steps * 1e3, measured=lt)
    if rank == 0 and n_gpus == 1 and not args.no_alu_ceiling:
        try:
            roof["alu_ceiling"] = alu_ceiling(k, hl, dev, gx, gy, dp, 0xA1C)
            roof["alu_ceiling"]["kernel_rate_over_ceiling"] = round(n * k.KNG_NB_RUN / (kms * 1e-3) / 1e6 / roof["alu_ceiling"]["value_mks"], 4)
        except Exception as e:  # noqa: BLE001
            roof["alu_ceiling"] = {"value_mks": None, "provenance": f"failed: {e}"}
    out = {
        "metric": "kangaroo jumps/sec (MK/s)",
        "value": round(value, 2),
        "unit": "MK/s",
        "n_gpus": n_gpus,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64",
        "data": "synthetic",
        "config": {
            "workload": f"80-bit range single key, auto DP {dp}, herd {gx}x{gy}x128 = 2^{np.log2(n):.0f} kangaroos/GPU, "
                        f"{k.KNG_NB_RUN} jumps/launch",
            "range_power": RANGE_POWER, "dp": dp, "grid": [gx, gy], "kangaroos_per_gpu": n,
            "group": cfg_now["group"], "lanes": cfg_now["lanes"], "share": cfg_now["share"], "asm_loop": cfg_now["asm_loop"],
            "device": info["name"], "arch": info["arch"],
            "parallelism": f"independent herds x{n_gpus}, no collective" + (f"; {per_rank_note}" if per_rank_note else ""),
            "dps_per_step": round(dps / args.steps, 1), "dps_lost": lost,
            # wave-iterations per launch that left the scheduled loop for the general arithmetic (of lanes/64 * group * 64)
            "exact_exits_per_step": round(counts.get("exits", 0) / args.steps, 1),
        },
        "roofline": roof,
        # the limiter, in the line itself: package power and GFX clock over the timed region (50 Hz samples + the device's
        # energy accumulator); `sustained` below repeats it over the pipeline leg's longer run
        "power": {"timed_region": power},
    }
    bpj = (roof["traffic"] / (n * k.KNG_NB_RUN)) if roof.get("traffic") else DESIGN_BYTES_PER_JUMP
    roof["power_bound"] = _power_bound(power, round(bpj, 1), n * k.KNG_NB_RUN / (kms * 1e-3) / 1e6)
    if rank == 0 and n_gpus == 1 and not args.no_secondary:
        out["secondary"] = secondary_lines(k, hl, dev, gx, gy)
    if rank == 0 and n_gpus == 1 and not args.no_pipeline:
        # reported next to the hot-path figure, never instead of it: the same workload through the host
        # pipeline (kangaroo_amd/host/kng_solver.cpp): every DP converted, queued and inserted into the table
        try:
            from kangaroo_amd import solver as sv

            s = sv.Solver(RANGE_START, RANGE_START + (1 << RANGE_POWER) - 1, (kx, ky), gpus=(local_rank,), grid=(gx, gy), dp=dp,
                          seed=0x5EED, max_launches=max(400, args.steps))  # ~9 s: the sustained figure (clock governor settled), 2^38.6 jumps audited
            s.prepare()
            sampler = GpuSampler([dev], hz=50.0).start()
            s.start()
            s.wait(180)
            sustained = sampler.stop().summary()
            st = s.stats()
            # whole-run audit on the device: every kangaroo and every table entry re-derived from its distance (kngs_audit)
            aud = s.audit(True)
            s.stop()
            s.close()
            out["power"]["sustained"] = sustained
            # the bound against the power of the LONG window (the 0.4 s timed region still holds the clock governor's ramp)
            if st["kernel_ms_avg"] > 0:
                roof["power_bound"] = dict(_power_bound(sustained, round(bpj, 1), jumps_per_step / (st["kernel_ms_avg"] * 1e-3) / 1e6),
                                           window="sustained: the pipeline leg")
            # the sustained figure next to the 0.5-second headline: the same kernel over the pipeline's longer run
            roof["frac_sustained"] = round(jumps_per_step * ALG_BYTES_PER_JUMP / (st["kernel_ms_avg"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            roof["sustained_kernel_ms"] = round(st["kernel_ms_avg"], 3)
            out["pipeline"] = {"value": round(st["jumps"] / st["seconds"] / 1e6, 2), "unit": "MK/s", "launches": st["launches"],
                               "kernel_ms_avg": round(st["kernel_ms_avg"], 3), "dps": st["dps"], "dps_lost": st["dps_lost"],
                               "audited_kangaroos": aud["kangaroos"], "audit_mismatches": aud["kangaroo_mismatches"] + aud["table_mismatches"],
                               "audited_table_points": aud["table_points"], "audit_ms": round(aud["herd_ms"] + aud["table_ms"], 2),
                               "audit": "every kangaroo (x, y) and every table entry re-derived from its distance on the device after the run",
                               "jumps_audited_log2": round(float(np.log2(max(1, st["jumps"]))), 2),
                               "what": "kngs_* solver: async DP drain + sharded DP table, wall clock incl. first and last launch"}
        except Exception as e:
            out["pipeline"] = {"value": None, "unit": "MK/s", "what": f"failed: {e}"}
    if rank == 0 and n_gpus == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline()
        except Exception as e:  # the baseline is reported, never required
            out["cpu_baseline"] = {"value": None, "unit": "MK/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}
    ranks.close()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
