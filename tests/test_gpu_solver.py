"""SURVEY 8(f) rows 1, 3, 4 on the GPU: the multi-GPU host pipeline (kangaroo_amd/host/kng_solver.cpp) around the
engine -- known-answer solves, the reference program reading our work files (-winfo / -wcheck / -i), save and
restore of the herds, same-herd replacement, and the pipeline keeping up with the kernel at the auto DP size
(where the reference's single-mutex host loop falls behind, DESIGN.md "Drop-in end to end").
"""
from __future__ import annotations

import os
import subprocess

import numpy as np
import pytest

from tests.helpers import N_ORDER, P, ref_binary

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_CPU = os.path.join(ROOT, "oracle", "_ref", "kangaroo_cpu")
REF_HIP = os.path.join(ROOT, "oracle", "_ref", "kangaroo_hip")

IN_TXT = (0, 0xFFFFFFFFFFFFFF, "02E9F43F810784FF1E91D8BC7C4FF06BFEE935DA71D7350734C3472FE305FEF82A", 0x378ABDEC51BC5D)
_S64 = 0x5B3F38AF935A3640D158E871CE6E9666DB862636383386EE0000000000000000
IN64 = (_S64, _S64 + (1 << 64) - 1, "03BB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4", _S64 + 0x510F18CCC3BD72EB)


def _decompress(pub_hex):
    x = int(pub_hex[2:], 16)
    y = pow((x * x * x + 7) % P, (P + 1) // 4, P)
    if (y & 1) != (int(pub_hex[:2], 16) & 1):
        y = P - y
    return x, y


@pytest.fixture(scope="module")
def sv(kng):  # kng: the engine library is present and a device exists
    from kangaroo_amd import solver

    return solver


@pytest.mark.parametrize("case,gpus,grid,dp", [(IN_TXT, (0,), (32, 128), -1), (IN64, (0,), (0, 0), 12),
                                               (IN64, (0, 0), (128, 128), 11)])
def test_solver_known_answers(sv, case, gpus, grid, dp):
    """The reference's shipped known-answer inputs (in.txt 56 bit, VC_CUDA8/in64.txt 64 bit = BASELINE configs[1]);
    the last case drives two engines from two host threads with two table consumers (multi-GPU path on one device).
    dp -1 = the suggested size (6 here: half a million DPs per 2 ms launch, the host is the bottleneck by design of
    the formula for small ranges); the 64-bit cases pin a larger dp so that the test stays short."""
    start, end, pub, answer = case
    s = sv.Solver(start, end, _decompress(pub), gpus=gpus, grid=grid, dp=dp, seed=7)
    s.start()
    rc = s.wait(240)
    st = s.stats()
    s.stop()
    assert rc == 1, st
    assert s.result() == answer
    assert st["wrong_collisions"] == 0 and st["dps_lost"] == 0
    assert st["kangaroos"] == len(gpus) * (grid[0] * grid[1] * 128 if grid[0] else st["kangaroos"] // len(gpus))
    s.close()


def test_tiny_range_replaces_same_herd_kangaroos(sv):
    """2^17 kangaroos on a 36-bit range: trails of the same herd merge all the time (Kangaroo.cpp:599-606); every such
    kangaroo is replaced through kng_set_kangaroo and the key still comes out."""
    import kangaroo_amd.hostlib as hl

    start = 0x77AA000000000000000000
    key = start + 0x9ABCD1234
    s = sv.Solver(start, start + (1 << 36) - 1, hl.pubkey(key)[1:], grid=(8, 128), dp=2, seed=3)
    s.start()
    rc = s.wait(120)
    st = s.stats()
    s.stop()
    assert rc == 1 and s.result() == key
    assert st["wrong_collisions"] == 0
    s.close()


@pytest.mark.parametrize("start,rp,key_off", [(0x3C0FFEE00000000000000000, 70, 0x2B5E6F7A8C9D0E1F23),
                                              (0, 125, 0x12345678FEDCBA9876543210DEADBEEF)])
def test_save_restore_continue_and_reference_reads_it(sv, orc, tmp_path, start, rp, key_off):
    """Run a few launches, save with kangaroos, restore into a new solver, run exactly one launch, save again:
    - the reference program accepts the first file (-winfo, -wcheck: every DP re-derived from its distance);
    - the restored herd after one launch equals the oracle walking the SAVED herd 64 jumps (bit-exact): nothing
      was lost or reordered through file -> host -> device;
    - counters carry over (Backup.cpp:163-170)."""
    import kangaroo_amd.hostlib as hl
    from tests.helpers import device_distances  # noqa: F401

    key = start + key_off  # rp 125 = BASELINE configs[4]: maximal range, start 0, wild distances wrap mod n
    kxy = hl.pubkey(key)[1:]
    grid = (8, 128)
    n = grid[0] * grid[1] * 128
    f1, f2 = str(tmp_path / "a.work"), str(tmp_path / "b.work")

    a = sv.Solver(start, start + (1 << rp) - 1, kxy, grid=grid, dp=6, seed=5, max_launches=5)
    a.start()
    assert a.wait(120) == 2
    a.save(f1, True)
    sa = a.stats()
    a.stop()
    a.close()
    assert sa["jumps"] == 5 * n * 64 and sa["launches"] == 5

    t1 = sv.DpTable()
    h1, n1, (x1, y1, d1) = sv.read_workfile(f1, t1)
    assert n1 == n and h1["dp"] == 6 and h1["count"] == sa["jumps"] and h1["key"] == kxy
    assert t1.count() == sa["dps"] - sa["same_herd"] > 0

    if ref_binary("kangaroo_cpu"):  # (fails on a box with a device when the binary is missing)
        info = subprocess.run([REF_CPU, "-winfo", f1], capture_output=True, text=True, timeout=120).stdout
        assert f"DP Count  : {t1.count()} " in info and f"Kangaroos : {n} " in info, info
        chk = subprocess.run([REF_CPU, "-t", "8", "-wcheck", f1], capture_output=True, text=True, timeout=300).stdout
        assert "[100.000% OK]" in chk, chk[-1500:]  # Check.cpp:398: every DP re-derived from (distance, type)

    b = sv.Solver(start, start + (1 << rp) - 1, kxy, grid=grid, dp=-1, seed=99, max_launches=1)
    b.load(f1)
    b.start()
    assert b.wait(120) == 2
    b.save(f2, True)
    sb = b.stats()
    b.stop()
    b.close()
    assert sb["dp"] == 6  # taken from the file
    assert sb["jumps"] == sa["jumps"] + n * 64 and sb["seconds"] >= h1["seconds"]

    t2 = sv.DpTable()
    h2, n2, (x2, y2, d2) = sv.read_workfile(f2, t2)
    assert n2 == n and t2.count() >= t1.count()
    # oracle: walk the saved herd one launch
    woff = ((1 << rp) - 1) >> 1
    jd, jx, jy, _ = orc.jump_table(rp)
    od = hl.to_device_distances(d1, woff)
    ox, oy = x1.copy(), y1.copy()
    orc.walk(ox, oy, od, 64, jd, jx, jy, orc.dp_mask(6), dp_cap=0)
    # kangaroos replaced after a same-herd collision during that launch differ by design
    same = np.all(x2 == ox, axis=1)
    assert same.sum() >= n - (sb["same_herd"] - 0) - 8
    assert np.array_equal(y2[same], oy[same])
    assert np.array_equal(hl.to_device_distances(d2, woff)[same], od[same])
    t1.close()
    t2.close()


@pytest.mark.parametrize("program", ["kangaroo_hip", "kangaroo_mi355x"])
def test_reference_program_resumes_from_our_workfile(sv, tmp_path, program):
    """`kangaroo -i ours.work` (reference host code on our engine; kangaroo_mi355x = the same with HashTable and SolveKeyGPU
    replaced at link time): loads our table and herd and finishes the solve."""
    exe = ref_binary(program)
    start, end, pub, answer = IN64
    grid = (64, 128)
    s = sv.Solver(start, end, _decompress(pub), grid=grid, seed=21, max_launches=3)
    s.start()
    rc = s.wait(120)
    path = str(tmp_path / "ours.work")
    s.save(path, True)
    s.stop()
    s.close()
    if rc == 1:
        pytest.skip("solved before the save (lucky herd)")
    out = subprocess.run([exe, "-t", "0", "-gpu", "-g", "%d,%d" % grid, "-i", path], capture_output=True, text=True, timeout=600)
    assert "Priv: 0x%X" % answer in out.stdout, out.stdout[-2500:] + out.stderr[-500:]
    assert "Fetch kangaroos" in out.stdout or "LoadWork" in out.stdout


def test_saves_while_running_are_consistent_snapshots(sv, tmp_path):
    """kngs_save parks the GPU threads at a launch boundary, waits for the DP queues to drain, writes, resumes.
    Repeated saves of a RUNNING two-engine solver must each be a consistent snapshot: the header count is a whole
    number of launches, the table holds exactly the DPs received minus the rejected ones, later snapshots extend
    earlier ones, and the reference's -wcheck accepts the last one."""
    import kangaroo_amd.hostlib as hl

    start = 0x42000000000000000000
    rp = 76
    kxy = hl.pubkey(start + 0x5A5A5A5A5A5A5A5A5A5)[1:]
    grid = (16, 128)
    n = grid[0] * grid[1] * 128
    s = sv.Solver(start, start + (1 << rp) - 1, kxy, gpus=(0, 0), grid=grid, dp=12, seed=31)
    s.start()
    prev_count, prev_items = 0, 0
    for i in range(4):
        assert s.wait(0.25) == 0
        path = str(tmp_path / f"snap{i}.work")
        s.save(path, with_kangaroos=(i == 3))
        t = sv.DpTable()
        h, nk, kang = sv.read_workfile(path, t, with_kangaroos=False)
        assert h["count"] % (n * 64) == 0 and h["count"] > prev_count, (i, h["count"])
        assert t.count() > prev_items
        assert nk == (2 * n if i == 3 else 0)
        prev_count, prev_items = h["count"], t.count()
        t.close()
    assert s.wait(0.2) == 0  # still running after the saves
    s.stop()
    st = s.stats()
    assert st["jumps"] >= prev_count and st["wrong_collisions"] == 0 and st["dps_lost"] == 0
    assert st["table_items"] == st["dps"] - st["same_herd"]
    s.close()
    if ref_binary("kangaroo_cpu"):  # (fails on a box with a device when the binary is missing)
        chk = subprocess.run([REF_CPU, "-t", "8", "-wcheck", path], capture_output=True, text=True, timeout=300).stdout
        assert "[100.000% OK]" in chk, chk[-1500:]


def test_solver_from_cpp(sv, tmp_path):
    """The C ABI of the pipeline used from C++ (the reference's language): tests/cpp/test_solver.cpp."""
    host = os.path.join(ROOT, "kangaroo_amd", "host")
    lib = os.path.join(ROOT, "kangaroo_amd", "lib")
    exe = str(tmp_path / "test_solver")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wall", "-I", host, os.path.join(ROOT, "tests", "cpp", "test_solver.cpp"),
                           "-o", exe, "-L", lib, "-lkangaroo_host", "-lkangaroo_hip", "-Wl,-rpath," + lib, "-lpthread"])
    out = subprocess.run([exe, str(tmp_path / "cpp.work")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "CPP solver ok: key 0x378ABDEC51BC5D" in out.stdout, out.stdout[-1500:] + out.stderr[-500:]


def test_pipeline_keeps_up_with_the_kernel_at_auto_dp(sv):
    """SURVEY 8(d) config 3 (80-bit range, 2^23 kangaroos, auto DP 14: about 33k DPs per 27 ms launch).  The
    reference's host loop (HashTable::Add under one mutex between launches) limits its own program to ~60 % of the
    kernel rate on this engine; this pipeline must stay within 10 % of it."""
    import kangaroo_amd.hostlib as hl

    start = int("B60E83280258A40F9CDF1649744D730D6E939DE92A2B" + "0" * 20, 16)
    key = start + 0xC0FFEE123456789ABCD
    s = sv.Solver(start, start + (1 << 80) - 1, hl.pubkey(key)[1:], seed=17, max_launches=40)
    s.start()
    assert s.wait(300) == 2
    st = s.stats()
    s.stop()
    assert st["dp"] == 14 and st["kangaroos"] == 1 << 23 and st["launches"] == 40
    kernel_rate = st["kangaroos"] * 64 / (st["kernel_ms_avg"] * 1e-3)
    wall_rate = st["jumps"] / st["seconds"]
    print(f"pipeline {wall_rate / 1e9:.2f} GK/s wall, kernel {kernel_rate / 1e9:.2f} GK/s, {st['dps']} DPs, "
          f"{st['same_herd']} replaced")
    assert st["dps_lost"] == 0 and st["wrong_collisions"] == 0
    assert wall_rate > 0.9 * kernel_rate
    s.close()


def test_four_engines_one_table_at_the_8gpu_dp_rate(sv):
    """the shared-table host path under many GPU threads.  Four engines on device 0 (2^21 kangaroos
    each), DP 9: every launch of every engine delivers 262 144 points -- the per-launch load of the 8-GPU configuration
    (2^23 kangaroos at DP 11) -- through four GPU threads into ONE table.  The device is time-shared, so the yardstick
    is the same four-engine job with a DP size that yields no points: the host path may cost at most 10 %."""
    import kangaroo_amd.hostlib as hl

    start = int("B60E83280258A40F9CDF1649744D730D6E939DE92A2B" + "0" * 20, 16)
    key = start + 0xC0FFEE123456789ABCD
    rates = {}
    for dp in (40, 9):
        s = sv.Solver(start, start + (1 << 80) - 1, hl.pubkey(key)[1:], gpus=(0, 0, 0, 0), grid=(512, 32), dp=dp, seed=23,
                      max_launches=48, warmup_launches=2)
        s.prepare()
        s.start()
        assert s.wait(300) == 2
        st = s.stats()
        load = s.consumer_load()
        s.stop()
        assert st["launches"] == 4 * 48 and st["kangaroos"] == 4 << 21 and st["dps_lost"] == 0 and st["wrong_collisions"] == 0
        rates[dp] = st["jumps"] / st["seconds"]
        if dp == 9:
            expect = (4 << 21) * 64 * 48 >> 9
            assert 0.97 * expect < st["dps"] < 1.03 * expect
            assert len(load) >= 2 and sum(load) == st["dps"] and max(load) < 1.2 * st["dps"] / len(load)
            print(f"4 engines x 262144 DPs/launch: {st['dps'] / st['seconds'] / 1e6:.1f} M DPs/s into one table ({len(load)} consumers), "
                  f"{rates[9] / 1e9:.2f} GK/s wall vs {rates[40] / 1e9:.2f} GK/s without points")
        s.close()
    assert rates[9] > 0.9 * rates[40]


def test_solver_survives_a_dp_buffer_that_is_too_small(sv):
    """max_found overflow inside the pipeline (GPUEngine.cu:641-648 semantics): with 512 slots for ~16 000 points per
    launch the surplus is dropped and COUNTED (dps_lost), nothing crashes, the stored points are real and the run ends."""
    import kangaroo_amd.hostlib as hl

    key = 0xB1E55ED5EEDF00D
    s = sv.Solver(0, (1 << 64) - 1, hl.pubkey(key)[1:], grid=(8, 128), dp=3, max_found=512, seed=5, max_launches=6, consumers=2)
    s.start()
    assert s.wait(120) in (1, 2)
    st = s.stats()
    s.stop()
    assert st["launches"] >= 1 and st["dps"] == 512 * st["launches"]
    expect = (8 * 128 * 128 * 64 >> 3) * st["launches"]
    assert 0.9 * expect < st["dps"] + st["dps_lost"] < 1.1 * expect and st["dps_lost"] > 10 * st["dps"]
    assert st["wrong_collisions"] == 0
    s.close()


def test_bench_multi_gpu_mode_on_one_device(sv):
    """`bench.py --gpus 2` = ONE process over two engines and one shared table (the form the driver's scaling run uses, there
    under torchrun with one engine per device).  Here both engines sit on device 0 (`--devices 0,0`): the JSON line must carry the
    contract's fields, per-GPU kernel figures, and a whole-job rate that is not above the sum of the kernel rates."""
    import json
    import sys

    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--devices", "0,0", "--steps", "6", "--warmup", "1",
                          "--grid", "128,128"], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["warmup"] == 1 and line["unit"] == "MK/s" and line["scaling"] == "weak"
    assert line["config"]["dp"] == 15 and line["config"]["kangaroos_per_gpu"] == 128 * 128 * 128  # DP from the population of both
    assert [p["launches"] for p in line["per_gpu"]] == [6, 6] and all(p["kernel_ms"] > 0 for p in line["per_gpu"])
    assert 0 < line["value"] <= 1.02 * line["kernel_rate_sum"]
    assert line["roofline"]["kernel"].startswith("kng_walk") and 0 < line["roofline"]["frac"] < 1
    assert line["config"]["dps_lost"] == 0


def test_bench_eight_engines_with_fewer_cpus_than_threads(sv):
    """`bench.py --gpus 8 --devices 0,0,0,0,0,0,0,0` confined to four CPUs (taskset): eight GPU threads + table threads on four
    CPUs is what a small CPU quota on an 8-GPU lease looks like.  The line must still appear, complete, with every launch
    done, no point lost, the table threads sized to the CPUs there are, each device row tied to rocm_smi by PCI address, and
    a clean whole-run audit."""
    import json
    import shutil
    import sys

    if not shutil.which("taskset"):
        pytest.skip("no taskset")
    cpus = sorted(os.sched_getaffinity(0))[:4]
    env = {k_: v for k_, v in os.environ.items() if k_ not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    out = subprocess.run(["taskset", "-c", ",".join(map(str, cpus)), sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--devices",
                          "0,0,0,0,0,0,0,0", "--steps", "4", "--warmup", "1", "--grid", "32,128"], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert "error" not in line, line.get("error")
    assert line["n_gpus"] == 8 and [p["launches"] for p in line["per_gpu"]] == [4] * 8
    assert line["config"]["dps_lost"] == 0 and line["host"]["effective_cpus"] <= 4.0
    assert 1 <= line["config"]["table_consumers"] <= 4
    assert line["host"]["pin_failures"] == 0
    assert line["audit"]["kangaroo_mismatches"] == 0 and line["audit"]["table_mismatches"] == 0
    dev = line["power"]["timed_region"]
    if dev.get("available"):
        assert dev["devices"][0]["smi_mapping"] in ("pci", "index (assumed)")


def test_restore_into_a_different_number_of_gpus(sv, tmp_path, orc):
    """FetchWalks (Kangaroo.cpp:646-668): a work file with fewer kangaroos than the GPUs hold restores what it has and the
    rest is created; one with more fills the herd and the surplus stays in the file.  The restored kangaroos are the
    saved ones: one launch later they stand where the oracle walks them."""
    import kangaroo_amd.hostlib as hl

    rp = 70
    start = 0x7A00000000000000000000
    kxy = hl.pubkey(start + 0x2B1B2C3D4E5F60718)[1:]
    grid = (8, 128)
    n = grid[0] * grid[1] * 128
    f1, f2 = str(tmp_path / "one.work"), str(tmp_path / "two.work")
    a = sv.Solver(start, start + (1 << rp) - 1, kxy, grid=grid, dp=6, seed=31, max_launches=2)
    a.start()
    assert a.wait(120) == 2
    a.save(f1, True)
    sa = a.stats()
    a.stop()
    a.close()
    assert sa["herd_created"] == n and sa["herd_loaded"] == 0 and sa["seed"] == 31
    _, n1, (x1, y1, d1) = sv.read_workfile(f1, None)
    assert n1 == n

    # two engines: the first gets the file's herd, the second a fresh one
    b = sv.Solver(start, start + (1 << rp) - 1, kxy, gpus=(0, 0), grid=grid, dp=-1, seed=32, max_launches=1)
    b.load(f1)
    b.start()
    assert b.wait(120) == 2
    b.save(f2, True)
    sb = b.stats()
    b.stop()
    b.close()
    assert sb["herd_loaded"] == n and sb["herd_created"] == n and sb["kangaroos"] == 2 * n
    assert sb["jumps"] == sa["jumps"] + 2 * n * 64 and sb["wrong_collisions"] == 0
    _, n2, (x2, y2, d2) = sv.read_workfile(f2, None)
    assert n2 == 2 * n
    woff = ((1 << rp) - 1) >> 1
    jd, jx, jy, _ = orc.jump_table(rp)
    od = hl.to_device_distances(d1, woff)
    ox, oy = x1.copy(), y1.copy()
    orc.walk(ox, oy, od, 64, jd, jx, jy, orc.dp_mask(6), dp_cap=0)
    same = np.all(x2[:n] == ox, axis=1)  # kangaroos replaced after a same-herd collision differ by design
    assert same.sum() >= n - sb["same_herd"] - 8
    assert np.array_equal(y2[:n][same], oy[same]) and np.array_equal(hl.to_device_distances(d2[:n], woff)[same], od[same])
    # the created half is a valid herd too: every sampled kangaroo sits at its distance
    for i in range(n, 2 * n, 4099):
        dt = hl.to_int(d2[i])
        _, px, py = hl.pubkey(dt % N_ORDER_) if i % 2 == 0 else (0, None, None)
        if i % 2 == 0:
            assert (px, py) == (hl.to_int(x2[i]), hl.to_int(y2[i]))

    # a smaller herd than the file holds: filled from the file, nothing created
    c = sv.Solver(start, start + (1 << rp) - 1, kxy, grid=(4, 128), dp=-1, seed=33, max_launches=1)
    c.load(f2)
    c.start()
    assert c.wait(120) == 2
    sc = c.stats()
    c.stop()
    c.close()
    assert sc["herd_loaded"] == 4 * 128 * 128 and sc["herd_created"] == 0


N_ORDER_ = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def test_configs4_save_restore_at_the_default_herd(sv, orc, tmp_path):
    """BASELINE configs[4] at its real size on one GPU: 125-bit range (the maximum), the DEFAULT grid
    = 2^23 kangaroos, whose kangaroo section is 96 B x 2^23 = 805 MB (Backup.cpp:525-546: 6.4 GB at 8 GPUs).
      * kngs_save with kangaroos: seconds and GB/s reported (gpurun_out/r03_configs4_default_herd.txt);
      * the reference's own -winfo reads the count, -wcheck accepts the file;
      * kngs_load into a NEW solver, exactly one launch, save again: ALL 2^23 (x, y, d) equal the oracle walking the
        SAVED herd 64 jumps (total parity: nothing lost or reordered through device -> file -> device)."""
    import time

    import kangaroo_amd as k
    import kangaroo_amd.hostlib as hl

    rp = 125
    key = (1 << 124) + 0x1234567890ABCDEF1234567890ABCDE
    kxy = hl.pubkey(key)[1:]
    gx, gy = k.default_grid(0)
    n = gx * gy * 128
    f1, f2 = str(tmp_path / "c4a.work"), str(tmp_path / "c4b.work")
    lines = []

    # DP 20 instead of the suggestion (33+ at this range and herd: no point in two launches): ~1000 distinguished points
    # for the reference's -wcheck to re-derive
    a = sv.Solver(0, (1 << rp) - 1, kxy, grid=(gx, gy), dp=20, seed=0xC4, max_launches=2)
    a.start()
    assert a.wait(300) == 2
    t0 = time.time()
    a.save(f1, True)
    t_save = time.time() - t0
    sa = a.stats()
    a.stop()
    a.close()
    size = os.path.getsize(f1)
    assert size >= 96 * n
    lines.append(f"herd {gx}x{gy}x128 = {n} kangaroos, range 2^{rp}, dp {sa['dp']}: file {size / 1e6:.1f} MB (kangaroo section {96 * n / 1e6:.1f} MB)")
    lines.append(f"kngs_save  with kangaroos: {t_save:.3f} s = {size / t_save / 1e9:.2f} GB/s (device -> pinned staging -> file, atomic rename)")

    if ref_binary("kangaroo_cpu"):  # (fails on a box with a device when the binary is missing)
        info = subprocess.run([REF_CPU, "-winfo", f1], capture_output=True, text=True, timeout=300).stdout
        assert f"Kangaroos : {n} " in info, info
        chk = subprocess.run([REF_CPU, "-t", "8", "-wcheck", f1], capture_output=True, text=True, timeout=600).stdout
        assert "[100.000% OK]" in chk, chk[-1500:]  # Check.cpp:398: every stored DP re-derived from (distance, type)
        lines.append(f"reference -winfo: kangaroo count {n} ok; reference -wcheck: [100.000% OK] over {sa['dps']} distinguished points")

    t0 = time.time()
    h1, n1, (x1, y1, d1) = sv.read_workfile(f1, None)
    lines.append(f"kngw read  (host, {n1} kangaroos): {time.time() - t0:.3f} s")
    assert n1 == n and h1["count"] == sa["jumps"] == 2 * n * 64

    b = sv.Solver(0, (1 << rp) - 1, kxy, grid=(gx, gy), dp=-1, seed=0xDEAD, max_launches=1)
    t0 = time.time()
    b.load(f1)
    b.prepare()
    t_load = time.time() - t0
    lines.append(f"kngs_load + prepare (file -> pinned staging -> device): {t_load:.3f} s = {size / t_load / 1e9:.2f} GB/s")
    b.start()
    assert b.wait(300) == 2
    b.save(f2, True)
    sb = b.stats()
    b.stop()
    b.close()
    assert sb["jumps"] == sa["jumps"] + n * 64 and sb["herd_loaded"] == n and sb["herd_created"] == 0
    _h2, n2, (x2, y2, d2) = sv.read_workfile(f2, None)
    assert n2 == n

    woff = ((1 << rp) - 1) >> 1
    jd, jx, jy, _ = orc.jump_table(rp)
    od = hl.to_device_distances(d1, woff)
    ox, oy = x1.copy(), y1.copy()
    t0 = time.time()
    orc.walk_parallel(ox, oy, od, 64, jd, jx, jy, hl.dp_mask(sb["dp"]))
    lines.append(f"oracle walk of the saved herd (64 jumps, thread pool): {time.time() - t0:.1f} s")
    assert sa["dps"] > 200 and sb["same_herd"] == 0  # no same-herd replacement happened: every kangaroo must match
    assert np.array_equal(x2, ox) and np.array_equal(y2, oy)
    assert np.array_equal(hl.to_device_distances(d2, woff), od)
    lines.append(f"restored herd after one launch == oracle walk of the saved herd: ALL {n} (x, y, d) bit-exact")
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r03_configs4_default_herd.txt"), "w") as f:
        f.write("# tests/test_gpu_solver.py::test_configs4_save_restore_at_the_default_herd (BASELINE configs[4] on one MI355X)\n" + "\n".join(lines) + "\n")


def test_solver_audit_herd_and_table(sv):
    """kngs_audit: while the solver RUNS the GPUs pause at a launch boundary and every kangaroo is
    re-derived from its distance; after the run the same plus every table entry (what an entry keeps of x: 128 + 18 bits).
    A bogus entry planted in the table and a corrupted kangaroo are both found."""
    import kangaroo_amd.hostlib as hl

    start = 0x5A5A000000000000000000
    key = start + 0x1F2E3D4C5B6A79881
    grid = (64, 64)
    n = grid[0] * grid[1] * 128
    s = sv.Solver(start, start + (1 << 70) - 1, hl.pubkey(key)[1:], gpus=(0, 0), grid=grid, dp=8, seed=77, max_launches=40)
    s.start()
    mid = s.audit(False)          # running: parks both GPU threads, audits both herds, resumes
    assert mid["kangaroos"] == 2 * n and mid["kangaroo_mismatches"] == 0 and mid["table_points"] == 0
    rc = s.wait(120)
    assert rc == 2, rc            # 70 bits are not solved in 40 launches
    full = s.audit(True)
    st = s.stats()
    assert full["kangaroos"] == 2 * n and full["kangaroo_mismatches"] == 0
    assert full["table_mismatches"] == 0 and full["table_points"] > 100000
    s.stop()
    st = s.stats()
    assert full["table_points"] == st["table_items"], (full, st)
    assert st["audits"] == 2 and st["audit_mismatches"] == 0 and st["audited_kangaroos"] == 4 * n
    # a planted entry: right format, wrong point
    t = s.table()
    assert t.add(0x1234567890ABCDEF1234567890ABCDEF00000000000000000000000000012345, 0x777, 0)[0] == sv.ADD_OK
    bad = s.audit(True)
    assert bad["table_points"] == full["table_points"] + 1 and bad["table_mismatches"] == 1 and bad["kangaroo_mismatches"] == 0
    s.close()
