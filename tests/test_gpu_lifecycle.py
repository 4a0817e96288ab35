"""Engine lifetime over the boundary (VERDICT r3 missing 1, weak 7 / ADVICE r3).

The reference re-creates and destroys its GPUEngine once per key (Kangaroo.cpp:1021-1075: the key loop; ctor :523,
`delete gpu` :634) and its shipped multi-key known-answer file holds 1000 keys.  Here: the unmodified reference program on
our engine over 25 of them, a 200-cycle kng_create / kng_destroy loop at the default herd that must give every byte of
device memory back, and the call-order rules the round-3 review found loose (kng_set_params against a launch in flight,
switching "dp_ring" with points waiting)."""
from __future__ import annotations

import os
import re
import subprocess
import time

import numpy as np
import pytest

from helpers import P, ref_binary

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _decompress(pub_hex):
    x = int(pub_hex[2:], 16)
    y = pow((x * x * x + 7) % P, (P + 1) // 4, P)
    if (y & 1) != (int(pub_hex[:2], 16) & 1):
        y = P - y
    return x, y


@pytest.mark.parametrize("program", ["kangaroo_hip", "kangaroo_mi355x"])
def test_reference_program_solves_25_keys_one_engine_per_key(kng, program, tmp_path):
    """`kangaroo_hip -t 0 -gpu -g 32,128 in40_25keys.txt`: the unmodified program creates, uses and deletes one GPUEngine
    per key.  Every printed private key must lie in the range and reproduce its public key; 25 keys, 25 answers.  (With the
    link-time replacements -- one Ingest, one set of table threads and one table Reset per key as well -- the first 10 keys.)"""
    import kangaroo_amd.hostlib as hl

    exe = ref_binary(program)
    cfg = os.path.join(ROOT, "tests", "golden", "in40_25keys.txt")
    lines = [l.strip() for l in open(cfg) if l.strip()]
    start, end, pubs = int(lines[0], 16), int(lines[1], 16), lines[2:]
    assert len(pubs) == 25
    if program != "kangaroo_hip":
        pubs = pubs[:10]
        cfg = str(tmp_path / "in40_10keys.txt")
        with open(cfg, "w") as f:
            f.write("\n".join(lines[:2] + pubs) + "\n")
    nkeys = len(pubs)
    t0 = time.time()
    # KNG_TRACE: kng_set_params reports its engine's buffers on stderr -- one line per GPUEngine the program creates
    out = subprocess.run([exe, "-t", "0", "-gpu", "-g", "32,128", cfg], capture_output=True, text=True, timeout=900,
                         env=dict(os.environ, KNG_TRACE="1"))
    dt = time.time() - t0
    text = out.stdout
    assert "Failed" not in text, text[-2000:]
    found = re.findall(r"Key#\s*(\d+) \[\d+.\]Pub:\s+0x([0-9A-Fa-f]+)\s*\n\s+Priv: 0x([0-9A-Fa-f]+)", text)
    assert len(found) == nkeys, (len(found), text[-2500:] + out.stderr[-500:])
    assert out.stderr.count("kng: planes") == nkeys, out.stderr[-1500:]  # one engine per key (the banner is printed for key 0 only, Kangaroo.cpp:525-526)
    for i, (idx, pub, priv) in enumerate(found):
        assert int(idx) == i and pub.upper() == pubs[i].upper()
        k = int(priv, 16)
        assert start <= k <= end
        assert hl.pubkey(k)[1:] == _decompress(pubs[i])
    print(f"\n{nkeys} keys of in40_1000.txt by the reference program on the engine: {dt:.1f} s, one GPUEngine per key")


def test_200_create_destroy_cycles_give_the_memory_back(kng):
    """kng_create (default grid: 2^23 kangaroos, 896 MiB of planes + pinned rings) / kng_set_params / herd / one launch /
    kng_destroy, 200 times: free device memory returns to its starting value, and nothing drifts over the cycles."""
    import kangaroo_amd.hostlib as hl

    gx, gy = kng.default_grid(0)
    rp = 80
    jd, jx, jy, _ = hl.jump_table(rp)
    _, kx, ky = hl.pubkey(0xDEADBEEF1234567)
    mask = hl.dp_mask(14)

    def cycle(i):
        eng = kng.GPUEngine(gx, gy, 0, 1 << 17)
        eng.SetParams(mask, jd, jx, jy)
        eng.CreateHerdOnDevice(rp, (kx, ky), seed=100 + i)
        eng.callKernel()
        if i % 2:      # destroy with the launch still in flight every other time (Kangaroo.cpp:572-634 does)
            eng.wait()
            assert len(eng.drain()) > 20000
        eng.close()

    cycle(0)  # first use loads the code object and warms the runtime's own pools
    free0, total = kng.device_free_bytes(0)
    frees, t0 = [], time.time()
    for i in range(1, 201):
        cycle(i)
        if i % 20 == 0:
            frees.append(kng.device_free_bytes(0)[0])
    dt = time.time() - t0
    # (a cycle holds ~0.95 GB: one leaked cycle would show as hundreds of MiB; the allowance is for the runtime's own pools)
    assert abs(frees[-1] - free0) <= (8 << 20), (free0, frees)
    assert max(frees) - min(frees) <= (8 << 20), frees
    print(f"\n200 create/destroy cycles at {gx}x{gy}x128 kangaroos: {1000 * dt / 200:.1f} ms per cycle, free device memory "
          f"{free0 / 2**30:.2f} GiB of {total / 2**30:.2f} GiB before and after")


def test_only_the_dp_buffers_of_the_mode_in_use_are_allocated(kng):
    """ADVICE r3 / VERDICT r3 weak 10: with dp_ring = 1 (default) no device DP buffers and no pinned landing buffer exist;
    switching to 0 allocates those and releases the rings.  Seen through GetMemory() (device bytes) and free memory."""
    max_found = 1 << 22  # 256 MiB per buffer: visible against allocation granularity
    before = kng.device_free_bytes(0)[0]
    eng = kng.GPUEngine(4, 16, 0, max_found)
    ring_mode = before - kng.device_free_bytes(0)[0]
    assert eng.get_option("dp_ring") == 1
    assert eng.GetMemory() < (64 << 20)            # the herd's planes and little else
    assert ring_mode < (128 << 20), ring_mode       # no 2 x 256 MiB of device DP buffers
    eng.set_option("dp_ring", 0)
    copy_mode = before - kng.device_free_bytes(0)[0]
    assert eng.GetMemory() >= 2 * max_found * 64
    assert copy_mode >= 2 * max_found * 64
    eng.set_option("dp_ring", 1)
    assert abs((before - kng.device_free_bytes(0)[0]) - ring_mode) <= (8 << 20)
    eng.close()
    assert abs(kng.device_free_bytes(0)[0] - before) <= (8 << 20)


def test_switching_dp_ring_with_points_waiting_is_refused(kng):
    """kng_set_option("dp_ring") used to discard the points of a launch that was waited for but not drained."""
    import kangaroo_amd.hostlib as hl
    from kangaroo_amd.engine import EngineError

    rp = 64
    jd, jx, jy, _ = hl.jump_table(rp)
    _, kx, ky = hl.pubkey(0x1234567)
    eng = kng.GPUEngine(4, 16, 0, 1 << 16)
    eng.SetParams(hl.dp_mask(4), jd, jx, jy)
    eng.CreateHerdOnDevice(rp, (kx, ky), seed=3)
    eng.callKernel()
    eng.wait()
    with pytest.raises(EngineError):
        eng.set_option("dp_ring", 0)
    n = len(eng.drain())
    assert n > 1000
    eng.set_option("dp_ring", 0)  # drained: allowed
    eng.callKernel()
    eng.wait()
    assert abs(len(eng.drain()) - n) < 0.2 * n
    eng.close()


def test_set_params_during_a_launch_takes_effect_at_the_boundary(kng, orc):
    """kng_set_params while a launch is in flight.  The launch in flight must finish with the mask and
    table it started with (the scheduled loop re-reads its constants at every entry), the next launch uses the new ones
    -- the reference's cudaMemcpyToSymbol blocks behind the kernel the same way (GPUEngine.cu:565-583).  Both launches are
    compared with the oracle, DP multisets included; repeated so that a race would have many chances."""
    import kangaroo_amd.hostlib as hl

    rp, grid = 72, (64, 64)  # 2^19 kangaroos: a launch lasts long enough for the call to land in the middle of it
    n = grid[0] * grid[1] * 128
    _, kx, ky = hl.pubkey(0xABCDEF0123)
    x, y, d_true, woff = hl.create_herd(n, rp, (kx, ky), seed=9)
    dd = hl.to_device_distances(d_true, woff)
    jd, jx, jy, _ = hl.jump_table(rp)
    jd2, jx2, jy2, _ = hl.jump_table(rp + 8)  # another table altogether
    m1, m2 = hl.dp_mask(7), hl.dp_mask(10)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 1 << 20)
    eng.SetWildOffset(woff)
    key = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
    for rep in range(3):
        eng.SetParams(m1, jd, jx, jy)
        eng.SetKangaroos(x, y, dd)
        eng.callKernel()
        eng.SetParams(m2, jd2, jx2, jy2)   # in flight: must not reach launch 1
        eng.wait()
        got1 = eng.drain(raw=True)
        eng.callKernel()
        eng.wait()
        got2 = eng.drain(raw=True)
        px, py, pd = eng.GetKangaroos(raw=True)
        ox, oy, od = x.copy(), y.copy(), dd.copy()
        want1 = orc.walk_parallel(ox, oy, od, 64, jd, jx, jy, m1)
        want2 = orc.walk_parallel(ox, oy, od, 64, jd2, jx2, jy2, m2)
        assert len(want1) == len(got1) and len(want2) == len(got2), (rep, len(want1), len(got1), len(want2), len(got2))
        assert sorted(map(key, got1)) == sorted(map(key, want1)), rep
        assert sorted(map(key, got2)) == sorted(map(key, want2)), rep
        assert np.array_equal(px, ox) and np.array_equal(py, oy) and np.array_equal(pd, od), rep
    eng.close()


def test_dp_capacity_can_be_raised_between_launches(kng, orc):
    """kng_reserve_points (what GPUEngine::SetParams calls to make the program's `maxFound` a floor): a capacity of 1024 loses
    points at dp 2; raised to 2^16 between launches the same herd loses none and delivers the oracle's multiset; the call is
    refused while a launch is outstanding or undrained, and never lowers the capacity."""
    import ctypes as C

    import numpy as np

    from helpers import device_distances, ints_to_array
    from test_gpu_parity import _seeded_herd

    lib = kng.load_library()
    lib.kng_reserve_points.argtypes = [C.c_void_p, C.c_uint32]
    grid, rp = (2, 2), 72
    n = grid[0] * grid[1] * 128
    x, y, true_d, woff = _seeded_herd(orc, n, rp, seed=4242)
    jd, jx, jy, _ = orc.jump_table(rp)
    mask = orc.dp_mask(2)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 1024)
    eng.SetParams(mask, jd, jx, jy)
    eng.SetWildOffset(woff)
    eng.SetKangaroos(x, y, ints_to_array(true_d))
    assert eng.get_option("max_found") == 1024
    eng.callKernel()
    assert lib.kng_reserve_points(eng._h, 1 << 16) == -4          # KNG_E_STATE: a launch is outstanding
    eng.wait()
    assert lib.kng_reserve_points(eng._h, 1 << 16) == -4          # ... or waited but not drained
    got = eng.drain(raw=True)
    assert len(got) == 1024 and eng.lastLost > 0                   # ~n*64/4 = 8192 points wanted 1024 slots
    assert lib.kng_reserve_points(eng._h, 1 << 16) == 0 and eng.get_option("max_found") == 1 << 16
    assert lib.kng_reserve_points(eng._h, 512) == 0 and eng.get_option("max_found") == 1 << 16    # never lowers
    eng.maxFound = 1 << 16
    eng._items = np.zeros(eng.maxFound, dtype=eng._items.dtype)
    gx, gy, gd = eng.GetKangaroos(raw=True)
    ox, oy, od = gx.copy(), gy.copy(), gd.copy()
    eng.callKernel()
    eng.wait()
    got = eng.drain(raw=True)
    want, total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 20)
    key = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
    assert eng.lastLost == 0 and total == len(got) and sorted(map(key, got)) == sorted(map(key, want))
    eng.close()


@pytest.mark.parametrize("ring", [1, 0])
def test_refused_capacity_growth_keeps_the_engine_walking_into_its_old_buffers(kng, orc, monkeypatch, ring):
    """kng_reserve_points used to free the DP buffers before it had the bigger ones; a refused allocation (a pinned
    ring against a memlock limit) then left the device-side loop arguments pointing at freed memory while GPUEngine::SetParams
    carried on ("keeping %u").  Now the new set is obtained first.  KNG_TEST_FAIL_RESERVE=1 refuses every growth: the call
    fails with KNG_E_ALLOC, the capacity and the buffers stay, and the launches after it deliver the oracle's points -- through
    the scheduled loop, whose arguments live on the device, in both landing modes."""
    import ctypes as C

    from helpers import ints_to_array
    from test_gpu_parity import _seeded_herd

    lib = kng.load_library()
    lib.kng_reserve_points.argtypes = [C.c_void_p, C.c_uint32]
    grid, rp = (2, 2), 72
    n = grid[0] * grid[1] * 128
    x, y, true_d, woff = _seeded_herd(orc, n, rp, seed=777)
    jd, jx, jy, _ = orc.jump_table(rp)
    mask = orc.dp_mask(5)
    with kng.GPUEngine(grid[0], grid[1], 0, 4096, dp_ring=ring, asm=1) as eng:
        eng.SetParams(mask, jd, jx, jy)
        eng.SetWildOffset(woff)
        eng.SetKangaroos(x, y, ints_to_array(true_d))
        eng.callKernel()
        eng.wait()
        eng.drain(raw=True)
        monkeypatch.setenv("KNG_TEST_FAIL_RESERVE", "1")
        assert lib.kng_reserve_points(eng._h, 1 << 20) == -2  # KNG_E_ALLOC
        assert b"refused" in lib.kng_last_error()
        monkeypatch.delenv("KNG_TEST_FAIL_RESERVE")
        assert eng.get_option("max_found") == 4096
        gx, gy, gd = eng.GetKangaroos(raw=True)
        ox, oy, od = gx.copy(), gy.copy(), gd.copy()
        key = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
        for _ in range(3):  # both launch slots
            eng.callKernel()
            eng.wait()
            got = eng.drain(raw=True)
            want, total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 20)
            assert eng.lastLost == 0 and total == len(got) and sorted(map(key, got)) == sorted(map(key, want))
        # ... and a growth that IS granted still works afterwards
        assert lib.kng_reserve_points(eng._h, 1 << 16) == 0 and eng.get_option("max_found") == 1 << 16
