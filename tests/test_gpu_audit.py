"""Whole-run parity audit on the device (include/kangaroo_hip.h kng_audit_*, kng_solver.h kngs_audit).

The default walk kernel's short arithmetic forms are exact only because a flag superset sends the rest to the general
arithmetic (tools/gen_walk_asm.py); an un-flagged inexact operand would silently kill a kangaroo for good.  The audit
re-derives every kangaroo -- and every distinguished point -- from its DISTANCE alone with general arithmetic, the way
the reference's -wcheck re-derives stored points on the CPU (Check.cpp:141-411), and compares with what the walk left.

These tests pin the audit itself (it finds each kind of corruption, it passes host-built and oracle-walked herds) and
then use it for depth: 500 launches = 2^38 jumps at the bench herd with zero mismatches over all 2^23 kangaroos.
"""
from __future__ import annotations

import numpy as np
import pytest

from helpers import N_ORDER, P

pytestmark = pytest.mark.gpu


def _herd(kng, rp, grid, key_scalar=0xC0FFEE1234567, seed=5, **opts):
    import kangaroo_amd.hostlib as hl

    _, kx, ky = hl.pubkey(key_scalar)
    n = grid[0] * grid[1] * 128
    x, y, d_true, woff = hl.create_herd(n, rp, (kx, ky), seed=seed)
    jd, jx, jy, _ = hl.jump_table(rp)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 1 << 17, **opts)
    eng.SetWildOffset(woff)
    eng.SetKangaroos(x, y, hl.to_device_distances(d_true, woff))
    return eng, (kx, ky), (jd, jx, jy), (x, y, d_true)


def _as_int(a):
    return sum(int(v) << (64 * i) for i, v in enumerate(a))


@pytest.mark.parametrize("rp,grid", [(80, (4, 16)), (125, (3, 5)), (40, (1, 1))])
def test_audit_finds_every_kind_of_corruption(kng, rp, grid):
    """A host-built herd (kngh_create_herd, itself pinned to the oracle) audits clean before and after walking; one flipped
    bit in x, in y or in the distance of one kangaroo -- tame or wild, first or last of a lane's batch -- is found, with
    its index.  Ragged herds (3 x 5 x 128) and the smallest one (128 kangaroos) included."""
    import kangaroo_amd.hostlib as hl

    eng, key, (jd, jx, jy), _ = _herd(kng, rp, grid)
    n = eng.nbKangaroo
    eng.SetParams(hl.dp_mask(6), jd, jx, jy)
    eng.audit_setup(key)
    assert eng.audit_herd() == (0, [])
    for _ in range(3):
        eng.callKernel()
        eng.wait()
        eng.drain()
    assert eng.audit_herd() == (0, [])
    px, py, pd = eng.GetKangaroos()  # true distances
    victims = [0, 1, n - 1, n // 2 + 1, 77 % n]
    for what in ("x", "y", "d"):
        for k in victims:
            vx, vy, vd = _as_int(px[k]), _as_int(py[k]), _as_int(pd[k])
            bx, by, bd = vx, vy, vd
            if what == "x":
                bx = vx ^ (1 << 200)
            elif what == "y":
                by = vy ^ 1
            else:
                bd = (vd + 1) % N_ORDER
            eng.SetKangaroo(k, bx, by, bd)
            bad, idx = eng.audit_herd()
            assert (bad, idx) == (1, [k]), (what, k, bad, idx)
            eng.SetKangaroo(k, vx, vy, vd)
    assert eng.audit_herd() == (0, [])
    # the walk still runs after audits (they borrow its product planes between launches) and stays clean
    eng.callKernel()
    eng.wait()
    assert eng.audit_herd() == (0, [])
    # a zero distance has no point: reported, and it does not poison the rest of its lane's batch
    eng.SetKangaroo(2, 1, 1, 0)
    assert eng.audit_herd() == (1, [2])
    eng.close()


def test_audit_needs_setup_and_a_quiet_engine(kng):
    import kangaroo_amd.hostlib as hl
    from kangaroo_amd.engine import EngineError

    eng, key, (jd, jx, jy), _ = _herd(kng, 64, (2, 2))
    eng.SetParams(hl.dp_mask(4), jd, jx, jy)
    with pytest.raises(EngineError):
        eng.audit_herd()  # no kng_audit_setup yet
    eng.audit_setup(key)
    eng.callKernel()
    with pytest.raises(EngineError):
        eng.audit_herd()  # a launch is outstanding: the audit would borrow the product planes it is using
    eng.wait()
    assert eng.audit_herd() == (0, [])
    eng.close()


def test_audit_of_dp_records_and_table_bits(kng, orc):
    """Every DP record of five launches re-derived from its distance (full x), then the same points reduced to what a
    hash-table entry keeps (x limbs 0-1 + the 18 bucket bits; compare mode 1).  Corrupted records are found by position;
    the records audit runs on its own stream WHILE the next launch is in flight."""
    import kangaroo_amd.hostlib as hl
    from kangaroo_amd.engine import RECORD_DTYPE

    eng, key, (jd, jx, jy), _ = _herd(kng, 72, (8, 32))
    eng.SetParams(hl.dp_mask(5), jd, jx, jy)
    eng.audit_setup(key)
    recs = []
    eng.callKernel()
    for _ in range(5):
        eng.wait()
        eng.callKernel()
        r = eng.drain_records()
        assert len(r) > 1000
        assert eng.audit_points(r) == (0, [])  # launch k+1 is running
        recs.append(r)
    eng.wait()
    allr = np.concatenate(recs)
    assert eng.audit_points(allr) == (0, [])
    # table form: only 146 bits of x survive
    tab = allr.copy()
    tab["x"][:, 2] &= np.uint64(0x3FFFF)
    tab["x"][:, 3] = 0
    tab["reserved"] = 1
    assert eng.audit_points(tab) == (0, [])
    tab["reserved"] = 0  # ... and in full-x mode every one of them is (rightly) a mismatch
    assert eng.audit_points(tab, cap=4)[0] == len(tab)
    tab["reserved"] = 1
    bad = tab.copy()
    pos = [0, 17, len(bad) - 1]
    bad["x"][pos[0], 0] ^= np.uint64(1)            # x limb 0
    bad["x"][pos[1], 2] ^= np.uint64(1 << 17)      # a bucket bit
    bad["d"][pos[2], 0] += np.uint64(1)            # the distance
    n_bad, idx = eng.audit_points(bad)
    assert n_bad == 3 and sorted(idx) == pos
    bad = tab.copy()
    bad["kidx"][5] ^= np.uint64(1)                 # wrong herd: tame re-derived as wild
    assert eng.audit_points(bad) == (1, [5])
    assert eng.audit_points(np.zeros(0, RECORD_DTYPE)) == (0, [])
    eng.close()


def test_audit_agrees_with_the_oracle_walk(kng, orc):
    """Independent of the engine's own walk: a herd walked by the ORACLE (64 jumps) and uploaded audits clean; the same
    herd with the oracle's distances but the start positions does not."""
    import kangaroo_amd.hostlib as hl

    rp, grid = 64, (2, 4)
    eng, key, (jd, jx, jy), (x, y, d_true) = _herd(kng, rp, grid)
    woff = eng.wildOffset
    dd = hl.to_device_distances(d_true, woff)
    ox, oy, od = x.copy(), y.copy(), dd.copy()
    orc.walk(ox, oy, od, 64, jd, jx, jy, hl.dp_mask(8))
    eng.audit_setup(key)
    eng.SetKangaroos(ox, oy, od)
    assert eng.audit_herd() == (0, [])
    eng.SetKangaroos(x, y, od)
    assert eng.audit_herd(cap=0)[0] == eng.nbKangaroo
    eng.close()


def test_500_launches_at_the_bench_herd_audit_clean(kng):
    """Depth: BASELINE configs[2] (80-bit range, 512 x 128 x 128 = 2^23 kangaroos, auto DP 14, the default kernel),
    500 launches = 2^38 jumps.  Every DP record of every launch is re-derived while the next launch runs, and at the end
    ALL 2^23 kangaroos are: zero mismatches.  A single inexact jump anywhere in those 2^38 would leave its kangaroo off
    its distance for good and be counted here (the negative tests above pin that)."""
    import kangaroo_amd.hostlib as hl

    rp = 80
    start = 0xB60E83280258A40F9CDF1649744D730D6E939DE92A2B << 80
    key = start + 0xC0FFEE0DDBA11F00D5EED
    _, kx, ky = hl.pubkey(key)
    _, sx, sy = hl.pubkey(start)
    rc, skx, sky = hl.point_add((kx, ky), (sx, P - sy))  # keyToSearch = key - rangeStart*G
    assert rc == 0
    gx, gy = kng.default_grid(0)
    n = gx * gy * 128
    dp = hl.suggest_dp(rp, n)
    jd, jx, jy, _ = hl.jump_table(rp)
    with kng.GPUEngine(gx, gy, 0, 1 << 17) as eng:
        eng.SetParams(hl.dp_mask(dp), jd, jx, jy)
        eng.CreateHerdOnDevice(rp, (skx, sky), seed=2024)
        eng.audit_setup((skx, sky))
        assert eng.audit_herd() == (0, [])
        assert eng.get_option("asm") == 1 and eng.get_option("dsplit") == 1 and eng.get_option("share") == 8
        launches, n_dp, exits = 500, 0, 0
        eng.callKernel()
        for i in range(launches):
            eng.wait()
            exits += eng.get_option("exact_exits")
            if i + 1 < launches:
                eng.callKernel()
            r = eng.drain_records()
            assert eng.lastLost == 0
            n_dp += len(r)
            assert eng.audit_points(r, cap=4) == (0, []), i
        bad, idx = eng.audit_herd()
        us = eng.get_option("audit_us")
        assert (bad, idx) == (0, [])
        expect = launches * n * 64 / (1 << dp)
        assert abs(n_dp - expect) < 6 * expect ** 0.5
        print(f"\n{launches} launches x {n} kangaroos x 64 jumps = 2^{np.log2(launches * n * 64):.2f} jumps, {n_dp} DP records and "
              f"all {n} kangaroos re-derived from their distances: 0 mismatches; {exits} exact-path exits; herd audit {us / 1000:.1f} ms")
