"""GPU parity tests proper: the HIP engine, called through the C ABI, against the CPU oracle
and against vectors produced by the reference's own objects.  Bit-exact everywhere (integer path).

These mirror `kangaroo -gpu -check` (Check.cpp:467-621): SetParams / SetWildOffset / SetKangaroos,
single-kangaroo overwrite, Launch; GetKangaroos; Launch, then compare every (x, y, d) and the DP
list -- but with an exact DP multiset comparison (the reference never checks for extra GPU DPs,
SURVEY App. D.3).
"""
import os

import numpy as np
import pytest

from helpers import (M128, N_ORDER, P, array_to_ints, device_distances, dp_multiset, host_distance, ints_to_array, ref_binary, walk_fixture)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.gpu

H = lambda s: int(s, 16)  # noqa: E731


# ------------------------------------------------------------------ primitives
def _pairs(vecs):
    a = ints_to_array([H(v[0]) for v in vecs])
    b = ints_to_array([H(v[1]) for v in vecs])
    return a, b


def test_modmul_golden(kng, golden):
    a, b = _pairs(golden["modmul"])
    r = kng.test_fieldop("modmul", a, b)
    assert array_to_ints(r) == [H(v[2]) for v in golden["modmul"]]


def test_modsqr_golden(kng, golden):
    a = ints_to_array([H(v[0]) for v in golden["modsqr"]])
    r = kng.test_fieldop("modsqr", a)
    assert array_to_ints(r) == [H(v[1]) for v in golden["modsqr"]]


def test_modsub_golden(kng, golden):
    a, b = _pairs(golden["modsub"])
    r = kng.test_fieldop("modsub", a, b)
    assert array_to_ints(r) == [H(v[2]) & ((1 << 256) - 1) for v in golden["modsub"]]


def test_modinv_golden(kng, golden):
    a = ints_to_array([H(v[0]) for v in golden["modinv"]])
    r = kng.test_fieldop("modinv", a)
    assert array_to_ints(r) == [H(v[1]) for v in golden["modinv"]]


@pytest.mark.parametrize("op", ["modmul", "modsqr", "modsub", "modinv"])
def test_primitives_random_vs_oracle(kng, orc, op):
    rng = np.random.default_rng(1234)
    n = 20000 if op != "modinv" else 4096
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    b = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    # sprinkle values just below 2^256 (non-canonical operands, carry corners of the fold)
    a[::97, 1:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    b[::89, 1:] = np.uint64(0xFFFFFFFFFFFFFFFF)
    got = kng.test_fieldop(op, a, b)
    want = np.zeros_like(a)
    fn = getattr(orc.lib, "orc_" + op)
    for i in range(n):
        if op in ("modmul", "modsub"):
            fn(want[i], a[i], b[i])
        else:
            fn(want[i], a[i])
    assert np.array_equal(got, want)


def test_fold_rare_paths(kng, orc):
    """The multiplier's rarely-taken branches (kng_field.h fe_fold32 -> fe_fold32_full): operands built to raise each
    condition, scattered among ordinary operands so that waves take the wave-uniform exit with mixed lanes."""
    from helpers import fold_rare_vectors

    rng = np.random.default_rng(5)
    vecs, found, want = fold_rare_vectors(rng)
    assert found >= want, want - found
    n = 64 * 40
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    b = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    for i, (x, y) in enumerate(vecs):
        for pos in (i * 67 + 3, n - 1 - i * 131):  # one per wave, varying lane
            a[pos], b[pos] = ints_to_array([x])[0], ints_to_array([y])[0]
    got = kng.test_fieldop("modmul", a, b)
    exp = np.zeros_like(a)
    for i in range(n):
        orc.lib.orc_modmul(exp[i], a[i], b[i])
    assert np.array_equal(got, exp)


def test_fieldop_empty(kng):
    e = np.zeros((0, 4), dtype=np.uint64)
    assert kng.test_fieldop("modmul", e, e).shape == (0, 4)


# ------------------------------------------------------------------ the walk, reference vectors
def _run_check_protocol(kng, w, jd, jx, jy, grid, launches, **opts):
    """Check.cpp:492-528 protocol generalised to `launches` kernels."""
    n = len(w["start"])
    eng = kng.GPUEngine(grid[0], grid[1], 0, 65536, **opts)
    assert eng.nbKangaroo == n
    eng.SetParams(w["dp_mask"], jd, jx, jy)
    eng.SetWildOffset(w["wild_offset"])
    eng.SetKangaroos(ints_to_array([s[0] for s in w["start"]]), ints_to_array([s[1] for s in w["start"]]),
                     ints_to_array([s[2] for s in w["start"]]))
    dps = []
    first = eng.Launch()
    assert len(first) == 0
    for _ in range(launches - 1):
        dps.append(eng.Launch())
    px, py, pd = eng.GetKangaroos()  # blocks until the in-flight kernel is done
    eng.wait()
    dps.append(eng.drain())
    eng.close()
    allp = [(int(r["kidx"]), array_to_ints([r["x"]])[0], array_to_ints([r["d"]])[0]) for part in dps for r in part]
    return list(zip(array_to_ints(px), array_to_ints(py), array_to_ints(pd))), allp


@pytest.mark.parametrize("name,grid,launches", [("walk_check64", (2, 1), 1), ("walk_80", (1, 1), 1),
                                                ("walk_125", (1, 1), 2)])
@pytest.mark.parametrize("group", [1, 4, 128])
def test_walk_matches_reference_vectors(kng, orc, golden, name, grid, launches, group):
    w = walk_fixture(golden[name])
    assert w["nsteps"] == 64 * launches
    jd, jx, jy, _ = orc.jump_table(w["range_power"])
    end, dps = _run_check_protocol(kng, w, jd, jx, jy, grid, launches, group=group)
    assert end == w["end"]
    assert dp_multiset(dps) == dp_multiset(w["dps"])


# ------------------------------------------------------------------ the walk, seeded herds vs the oracle
def _seeded_herd(orc, n, range_power, seed):
    rng = np.random.default_rng(seed)
    key = int(rng.integers(1, 1 << 62)) | (1 << (range_power - 1))
    _, kx, ky = orc.pubkey(key)
    width = (1 << range_power) - 1
    wild_offset = width >> 1
    true_d = []
    for i in range(n):
        d = int.from_bytes(rng.bytes(16), "little") & width
        if i & 1:
            d = (d - wild_offset) % N_ORDER
        true_d.append(d)
    d4 = ints_to_array(true_d)
    x, y = orc.create_herd(d4, 0, kx, ky)
    return x, y, true_d, wild_offset


@pytest.mark.parametrize("grid,group,block,launches,dp", [
    ((2, 2), 1, 64, 1, 5),
    ((2, 2), 2, 64, 2, 5),
    ((3, 5), 8, 128, 2, 4),      # lanes not a multiple of the block: ragged last workgroup
    ((4, 4), 32, 256, 3, 6),
    ((2, 4), 128, 64, 2, 0),     # dp=0: every jump is a DP -> 64*n points, exercises max_found clamp
    ((8, 4), 64, 64, 1, 7),
])
def test_walk_vs_oracle(kng, orc, grid, group, block, launches, dp):
    n = grid[0] * grid[1] * 128
    rp = 72
    x, y, true_d, wild_offset = _seeded_herd(orc, n, rp, seed=grid[0] * 100 + group)
    jd, jx, jy, _ = orc.jump_table(rp)
    mask = orc.dp_mask(dp)
    max_found = 1 << 16
    eng = kng.GPUEngine(grid[0], grid[1], 0, max_found, group=group, block=block)
    eng.SetParams(mask, jd, jx, jy)
    eng.SetWildOffset(wild_offset)
    eng.SetKangaroos(x, y, ints_to_array(true_d))

    ox, oy = x.copy(), y.copy()
    od = ints_to_array(device_distances(true_d, wild_offset), 2)
    for launch in range(launches):
        eng.callKernel()
        eng.wait()
        got = eng.drain(raw=True)
        want, total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 22)
        if total <= max_found:
            assert eng.lastLost == 0
            assert dp_multiset((r["kidx"], array_to_ints([r["x"]])[0], array_to_ints([r["d"]])[0]) for r in got) == \
                dp_multiset((r["kidx"], array_to_ints([r["x"]])[0], array_to_ints([r["d"]])[0]) for r in want)
        else:
            # GPUEngine.cu:641-648: surplus points are dropped and counted
            assert len(got) == max_found and eng.lastLost == total - max_found
            wantset = set(dp_multiset((r["kidx"], array_to_ints([r["x"]])[0], array_to_ints([r["d"]])[0]) for r in want))
            assert all((int(r["kidx"]), array_to_ints([r["x"]])[0], array_to_ints([r["d"]])[0]) in wantset for r in got[::37])
        gx, gy, gd = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gd, od), f"launch {launch}"
    eng.close()


def test_set_get_roundtrip_and_single_overwrite(kng, orc):
    """SetKangaroos/GetKangaroos are exact inverses incl. the wild offset (GPUEngine.cu:381-480);
    SetKangaroo (GPUEngine.cu:483-538) lands after an in-flight launch."""
    grid = (2, 3)
    n = grid[0] * grid[1] * 128
    rp = 125
    x, y, true_d, wild_offset = _seeded_herd(orc, n, rp, seed=7)
    jd, jx, jy, _ = orc.jump_table(rp)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 4096, group=16)
    eng.SetParams(orc.dp_mask(10), jd, jx, jy)
    eng.SetWildOffset(wild_offset)
    eng.SetKangaroos(x, y, ints_to_array(true_d))
    gx, gy, gd = eng.GetKangaroos()
    assert np.array_equal(gx, x) and np.array_equal(gy, y) and array_to_ints(gd) == true_d

    # overwrite kangaroo r (odd -> wild) while a launch is in flight: takes effect after it
    r = 555
    assert r & 1
    nd = (true_d[r] + 12345) % N_ORDER
    _, nx, ny = orc.pubkey(99)
    eng.callKernel()
    eng.SetKangaroo(r, nx, ny, nd)
    eng.wait()
    gx, gy, gd = eng.GetKangaroos()
    assert array_to_ints(gx[r:r + 1])[0] == nx and array_to_ints(gy[r:r + 1])[0] == ny
    assert array_to_ints(gd[r:r + 1])[0] == nd
    # everyone else did exactly 64 jumps
    ox, oy = x.copy(), y.copy()
    od = ints_to_array(device_distances(true_d, wild_offset), 2)
    orc.walk(ox, oy, od, 64, jd, jx, jy, 0, dp_cap=0)
    keep = np.arange(n) != r
    assert np.array_equal(gx[keep], ox[keep]) and np.array_equal(gy[keep], oy[keep])
    eng.close()


def test_call_sequence_errors(kng, orc):
    eng = kng.GPUEngine(1, 1, 0, 16)
    with pytest.raises(kng.EngineError):
        eng.callKernel()  # no params / herd yet
    jd, jx, jy, _ = orc.jump_table(64)
    eng.SetParams(0, jd, jx, jy)
    with pytest.raises(kng.EngineError):
        eng.callKernel()  # no herd
    with pytest.raises(kng.EngineError):
        eng.SetKangaroos(np.zeros((5, 4), np.uint64), np.zeros((5, 4), np.uint64), np.zeros((5, 2), np.uint64))
    with pytest.raises(kng.EngineError):
        eng.set_option("group", 3)
    eng.close()
    with pytest.raises(kng.EngineError):
        kng.GPUEngine(1, 1, 99, 16)  # bad device id


def test_default_grid_and_banner(kng):
    x, y = kng.default_grid(0)
    info = kng.device_info(0)
    assert x == 2 * info["cu_count"] and y == 128  # GPUEngine.cu:301-303
    assert kng.default_grid(0, 7, 9) == (7, 9)
    eng = kng.GPUEngine(1, 1, 0, 16)
    assert eng.deviceName.startswith("GPU #0 ") and "Grid(1x1)" in eng.deviceName
    assert eng.GetGroupSize() == 128 and eng.GetNbThread() == 1 and eng.GetMemory() > 0
    eng.close()


# ------------------------------------------------------------------ BASELINE.json full size
def test_full_size_herd_properties(kng, orc):
    """The headline herd (reference default grid 2*CU x 128 -> 2^23 kangaroos, 80-bit range, auto DP)
    for three launches, checked through size-independent properties:
      * the group invariant of the walk: after any number of jumps every kangaroo still sits at
        d*G (tame) / K + d*G (wild) -- verified with the oracle on a random sample of the herd,
      * every reported DP has its masked bits clear, is reported by an existing kangaroo, and its
        (x, d) satisfies the same invariant,
      * every kangaroo advanced: total distance gained equals the sum of 192 table jumps in range,
      * a sampled sub-herd replayed by the oracle for 192 jumps gives bit-identical (x, y, d)."""
    import kangaroo_amd.hostlib as hl

    gx, gy = kng.default_grid(0)
    n = gx * gy * 128
    rp = 80
    key = (0xB60E83280258A40F9CDF1649744D730D6E939DE92A2B << 80) + 0xC0FFEE123456789ABCD
    _, kx, ky = hl.pubkey(key)
    _, sx, sy = hl.pubkey(0xB60E83280258A40F9CDF1649744D730D6E939DE92A2B << 80)
    _, ksx, ksy = hl.point_add((kx, ky), (sx, P - sy))  # keyToSearch, Kangaroo.cpp:892-909
    x, y, d_true, woff = hl.create_herd(n, rp, (ksx, ksy), seed=2024)
    dev_d = hl.to_device_distances(d_true, woff)
    dp = hl.suggest_dp(rp, n)
    assert dp == 14
    jd, jx, jy, _ = hl.jump_table(rp)
    mask = hl.dp_mask(dp)
    launches = 3
    rng = np.random.default_rng(5)
    sample = np.sort(rng.choice(n, size=1536, replace=False))
    with kng.GPUEngine(gx, gy, 0, 65536 * 2) as eng:
        eng.SetParams(mask, jd, jx, jy)
        eng.SetWildOffset(woff)
        eng.SetKangaroos(x, y, dev_d)
        dps = []
        for _ in range(launches):
            eng.callKernel()
            eng.wait()
            dps.append(eng.drain(raw=True))
            assert eng.lastLost == 0
        gx_, gy_, gd_ = eng.GetKangaroos(raw=True)
    dps = np.concatenate(dps)
    # DP rate: 2^23 * 192 jumps / 2^14
    expect = n * 64 * launches / 2 ** dp
    assert 0.9 * expect < len(dps) < 1.1 * expect
    assert np.all((dps["x"][:, 3] & np.uint64(mask)) == 0)
    assert int(dps["kidx"].max()) < n

    def check_invariant(xs, ys, dd, kidx):
        true_d = hl.to_true_distances(np.ascontiguousarray(dd), woff, np.ascontiguousarray(kidx))
        # oracle: tame d*G, wild K + d*G
        ox = np.zeros_like(xs)
        oy = np.zeros_like(xs)
        for i in range(len(kidx)):
            if int(kidx[i]) & 1:
                orc.lib.orc_pubkey_add(ox[i], oy[i], true_d[i], ints_to_array([ksx])[0], ints_to_array([ksy])[0])
            else:
                orc.lib.orc_pubkey(ox[i], oy[i], true_d[i])
        assert np.array_equal(ox, xs)
        if ys is not None:
            assert np.array_equal(oy, ys)

    check_invariant(gx_[sample], gy_[sample], gd_[sample], sample.astype(np.uint64))
    pick = rng.choice(len(dps), size=512, replace=False)
    check_invariant(dps["x"][pick], None, dps["d"][pick], dps["kidx"][pick])
    # distances only ever grow by table jumps: 192 jumps of at most max(jd) each
    gained = gd_[sample].astype(object)[:, 0] + (gd_[sample].astype(object)[:, 1] << 64) - (
        dev_d[sample].astype(object)[:, 0] + (dev_d[sample].astype(object)[:, 1] << 64))
    jmax = max(array_to_ints(jd))
    jmin = min(array_to_ints(jd))
    assert all(192 * jmin <= int(g) <= 192 * jmax for g in gained)
    # bit-exact replay of a contiguous sub-herd by the oracle (walks are independent)
    sub = slice(4096, 4096 + 2048)
    ox, oy, od = x[sub].copy(), y[sub].copy(), dev_d[sub].copy()
    orc.walk(ox, oy, od, 64 * launches, jd, jx, jy, mask, dp_cap=0)
    assert np.array_equal(gx_[sub], ox) and np.array_equal(gy_[sub], oy) and np.array_equal(gd_[sub], od)


@pytest.mark.parametrize("use_asm", [1, 0])
@pytest.mark.parametrize("share", [8, 4])
@pytest.mark.parametrize("rp", [72, 109, 125])
def test_every_walk_kernel_vs_oracle(kng, orc, share, rp, use_asm):
    """All eight instantiations of the walk kernel -- {low-word distance streaming, both words} x {scheduled asm loop,
    compiler-scheduled loop} x {512-thread blocks with a two-level inversion tree, 256-thread blocks with one level: what
    herds too small to fill the chip get since round 5} -- as the engine itself selects them: a 72-bit range streams only the low word;
    BASELINE configs[3]'s 109-bit range (jump distances around 2^54: a lane's low word carries every ~700 jumps) still
    does with the scheduled loop, which adds the carries in the loop with L2 atomics, and streams both words with the
    compiler loop; configs[4]'s 125-bit range streams both.  States and the exact DP multiset over two launches."""
    dsplit = {72: 1, 109: 1 if use_asm else 0, 125: 0}[rp]
    grid = (4, 4)
    n = grid[0] * grid[1] * 128
    x, y, true_d, wild_offset = _seeded_herd(orc, n, rp, seed=1000 + rp + share)
    jd, jx, jy, _ = orc.jump_table(rp)
    mask = orc.dp_mask(5)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 1 << 16, share=share, group=16, asm=use_asm)
    eng.SetParams(mask, jd, jx, jy)
    assert eng.get_option("dsplit") == dsplit and eng.get_option("share") == share and eng.get_option("asm") == use_asm
    eng.SetWildOffset(wild_offset)
    eng.SetKangaroos(x, y, ints_to_array(true_d))
    ox, oy = x.copy(), y.copy()
    od = ints_to_array(device_distances(true_d, wild_offset), 2)
    key = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
    for _ in range(2):
        eng.callKernel()
        eng.wait()
        got = eng.drain(raw=True)
        want, _total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 20)
        assert sorted(map(key, got)) == sorted(map(key, want))
        gx, gy, gd = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gd, od)
    eng.close()


@pytest.mark.parametrize("ring", [1, 0])
@pytest.mark.parametrize("use_asm", [1, 0])
def test_dp_ring_delivers_the_same_records(kng, orc, use_asm, ring):
    """Option "dp_ring": the kernel writes its DP records straight into pinned, device-mapped host memory (north_star:
    "compaction of distinguished points into a pinned host ring buffer") instead of a device buffer that land_points
    copies.  Same multiset as the oracle over three launches (both buffers of the ring get used), through kng_drain and
    through the zero-copy view; a buffer that is too small loses the same number of points either way.  ring = 0 is the
    device-buffer-and-copy path of rounds 1-2 (still an option)."""
    grid, rp = (4, 8), 72
    n = grid[0] * grid[1] * 128
    x, y, true_d, wild_offset = _seeded_herd(orc, n, rp, seed=4242)
    jd, jx, jy, _ = orc.jump_table(rp)
    mask = orc.dp_mask(3)
    key = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
    ox, oy = x.copy(), y.copy()
    od = ints_to_array(device_distances(true_d, wild_offset), 2)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 1 << 16, asm=use_asm, dp_ring=ring)
    assert eng.get_option("dp_ring") == ring
    eng.SetParams(mask, jd, jx, jy)
    eng.SetWildOffset(wild_offset)
    eng.SetKangaroos(x, y, ints_to_array(true_d))
    for _ in range(3):
        eng.callKernel()
        eng.wait()
        got = eng.drain(raw=True)
        want, _total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 20)
        assert len(got) > 1000 and sorted(map(key, got)) == sorted(map(key, want))
    eng.close()
    # overflow: 512 slots for ~30 000 points
    small = kng.GPUEngine(grid[0], grid[1], 0, 512, asm=use_asm, dp_ring=ring)
    small.SetParams(mask, jd, jx, jy)
    small.SetWildOffset(wild_offset)
    small.SetKangaroos(x, y, ints_to_array(true_d))
    small.callKernel()
    small.wait()
    got = small.drain(raw=True)
    ox, oy = x.copy(), y.copy()
    od = ints_to_array(device_distances(true_d, wild_offset), 2)
    want, total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 20)
    assert len(got) == 512 and small.lastLost == total - 512
    assert set(map(key, got)) <= set(map(key, want))
    small.close()


@pytest.mark.parametrize("rp,dp,dsplit,grid", [(80, 14, 1, (512, 128)), (80, 14, 0, (512, 128)), (109, 25, 1, (512, 128)), (109, 25, 0, (512, 128)),
                                                (80, 12, 1, (512, 1)), (80, 12, 0, (512, 1))])
def test_bench_config_total_parity(kng, orc, rp, dp, dsplit, grid):
    """BASELINE.md 3's gate taken literally, at the bench configuration (80-bit range, grid 512x128 = 2^23 kangaroos,
    auto DP 14, default kernel): after one launch ALL 2^23 (x, y, d) triples and the COMPLETE distinguished-point
    multiset equal the oracle's (walked over a thread pool: kangaroos are independent).  Both distance layouts, and the
    puzzle-#110 table (BASELINE configs[3]: 109-bit range, DP 25, jump distances ~2^54) at the same herd -- in the layout
    the engine picks for it since round 3 (low word streams, ~12 000 carries per jump of the herd go through L2 atomics) and
    with both words streaming.  Round 5 adds BASELINE configs[2] read literally: herd = 2*CU x 128 = 65 536 kangaroos (grid
    512 x 1; DP 12 so that a launch still yields ~1000 points), which the engine walks one kangaroo per lane in 256-thread
    blocks (share 4)."""
    import kangaroo_amd.hostlib as hl

    gx, gy = grid
    n = gx * gy * 128
    literal = n == 65536
    _, kx, ky = hl.pubkey((1 << (rp - 1)) + 0xC0FFEE123456789ABCD)
    jd, jx, jy, _ = hl.jump_table(rp)
    ojd, ojx, ojy, _ = orc.jump_table(rp)
    assert np.array_equal(jd, ojd) and np.array_equal(jx, ojx) and np.array_equal(jy, ojy)
    mask = hl.dp_mask(dp)
    with kng.GPUEngine(gx, gy, 0, 1 << 17, dsplit=dsplit) as eng:
        eng.SetParams(mask, jd, jx, jy)
        assert eng.get_option("dsplit") == dsplit
        assert (eng.get_option("share"), eng.get_option("group")) == ((4, 1) if literal else (8, 64))
        eng.CreateHerdOnDevice(rp, (kx, ky), seed=0xBEEF + dsplit)
        x0, y0, d0 = eng.GetKangaroos(raw=True)
        eng.callKernel()
        eng.wait()
        got = eng.drain(raw=True)
        assert eng.lastLost == 0
        exits = eng.get_option("exact_exits")
        x1, y1, d1 = eng.GetKangaroos(raw=True)
    # the scheduled loop flags a SUPERSET of the operands its short forms are not exact for: ~88 tracked words per jump, each
    # within 1024 of 2^32 (or below 1024) with probability 2^-22, 64 lanes per wave -> ~1.3e-3 of the wave-iterations
    wave_iterations = (n // 64) * 63
    if not literal:  # (1024 waves: a count of the order of 100, too few for a band)
        assert 0.3e-3 * wave_iterations < exits < 4e-3 * wave_iterations, (exits, wave_iterations)
    want = orc.walk_parallel(x0, y0, d0, 64, jd, jx, jy, mask)
    assert np.array_equal(x1, x0) and np.array_equal(y1, y0) and np.array_equal(d1, d0)  # x0.. now hold the oracle's end state
    mean = (n * 64) >> dp
    assert len(got) == len(want) and mean - 6 * mean**0.5 - 1 < len(got) < mean + 6 * mean**0.5 + 1

    def canon(r):  # order by (kidx, d low word): a kangaroo may hit two distinguished points in one launch
        o = np.lexsort((r["d"][:, 0], r["kidx"]))
        return r["kidx"][o], r["x"][o], r["d"][o]

    for a, b in zip(canon(got), canon(want)):
        assert np.array_equal(a, b)


def test_device_built_herd_at_full_size(kng, orc):
    """kng_build_herd at the bench configuration (2^23 kangaroos): every 7th kangaroo -- 1.2 million, both types, every lane
    and batch position -- sits at d*G (tame) / K' + d*G (wild) for the distance the engine reports, recomputed by the oracle
    over a thread pool; all distances are in range and practically all distinct."""
    import kangaroo_amd.hostlib as hl

    rp, gx, gy = 80, 512, 128
    n = gx * gy * 128
    _, kx, ky = hl.pubkey((1 << 79) + 0x5EED5EED5EED)
    jd, jx, jy, _ = hl.jump_table(rp)
    with kng.GPUEngine(gx, gy, 0, 1 << 17) as eng:
        eng.SetParams(hl.dp_mask(14), jd, jx, jy)
        woff = eng.CreateHerdOnDevice(rp, (kx, ky), seed=0xF00D)
        x, y, dd = eng.GetKangaroos(raw=True)
    assert woff == ((1 << rp) - 1) >> 1
    assert np.all(dd[:, 1] < np.uint64(1 << (rp - 64))) and len(np.unique(dd[:, 0])) > 0.999 * n
    idx = np.arange(0, n, 7, dtype=np.uint64)
    d_true = hl.to_true_distances(np.ascontiguousarray(dd[idx]), woff, idx)
    # an odd stride: the sample alternates tame / wild by position, like a herd (Kangaroo.cpp:699), so create_herd applies
    ox, oy = orc.create_herd_parallel(d_true, kx, ky)
    assert np.array_equal(ox, x[idx]) and np.array_equal(oy, y[idx])


def test_allocation_failure_is_reported_not_fatal(kng):
    """KNG_E_ALLOC: a herd that cannot fit the device (2^33 kangaroos, 960 GB of state) fails kng_create with the
    allocation error code and leaks nothing -- a normal engine can be created right after."""
    with pytest.raises(kng.EngineError, match=r"error -2: herd state"):
        kng.GPUEngine(1 << 16, 1024, 0, 65536)
    with kng.GPUEngine(2, 2, 0, 1024) as eng:
        assert eng.nbKangaroo == 512


@pytest.mark.parametrize("program", ["kangaroo_hip", "kangaroo_mi355x"])
def test_reference_gpu_check_harness_on_our_engine(program):
    """The reference's own CPU/GPU parity harness, `kangaroo -gpu -check` (Check.cpp:467-621), unmodified, on our
    engine: SetKangaroos, single SetKangaroo, Launch x2, GetKangaroos against SECPK1 AddDirect, every DP found.
    -g 8,128 keeps the herd below Check's hard-coded maxFound (65536 DPs at dp=8, Check.cpp:418,492)."""
    import subprocess

    exe = ref_binary(program)
    out = subprocess.run([exe, "-gpu", "-g", "8,128", "-check"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "CPU/GPU ok" in out.stdout, out.stdout[-2500:] + out.stderr[-500:]
    assert "DP found" in out.stdout and "warning" not in out.stdout.lower()


# ------------------------------------------------------------------ end to end: actually solve keys
IN_TXT_RANGE_END = 0xFFFFFFFFFFFFFF          # the reference's shipped 56-bit known-answer input (in.txt)
IN_TXT_PUBKEY = "02E9F43F810784FF1E91D8BC7C4FF06BFEE935DA71D7350734C3472FE305FEF82A"
IN_TXT_ANSWER = 0x378ABDEC51BC5D             # README.md:331-357


def _decompress(pub_hex):
    x = int(pub_hex[2:], 16)
    y = pow((x * x * x + 7) % P, (P + 1) // 4, P)
    if (y & 1) != (int(pub_hex[:2], 16) & 1):
        y = P - y
    return x, y


def test_solve_in_txt_with_python_host(kng):
    """Kangaroo collision search driven from Python on the engine: tame/wild DPs into a dict, a
    tame-wild collision on x gives the key (Kangaroo.cpp:218-253).  Known answer of in.txt."""
    import kangaroo_amd.hostlib as hl

    rp = 56
    kx, ky = _decompress(IN_TXT_PUBKEY)            # range start is 0: keyToSearch = K
    gx, gy = 32, 128
    n = gx * gy * 128                               # 2^19 kangaroos
    x, y, d_true, woff = hl.create_herd(n, rp, (kx, ky), seed=11)
    jd, jx, jy, _ = hl.jump_table(rp)
    dp = 9
    table = {}
    found = None
    with kng.GPUEngine(gx, gy, 0, 65536 * 2) as eng:
        eng.SetParams(hl.dp_mask(dp), jd, jx, jy)
        eng.SetWildOffset(woff)
        eng.SetKangaroos(x, y, hl.to_device_distances(d_true, woff))
        eng.callKernel()
        for launch in range(400):                   # expected ~2^29.1 jumps = 2^26 per launch -> ~10 launches
            items = eng.Launch()                    # previous kernel's DPs (true distances), next kernel started
            for it in items:
                xx = tuple(int(v) for v in it["x"])
                kind = int(it["kidx"]) & 1
                dist = sum(int(v) << (64 * i) for i, v in enumerate(it["d"]))
                other = table.get(xx)
                if other is None:
                    table[xx] = (kind, dist)
                elif other[0] != kind:
                    dt, dw = (dist, other[1]) if kind == 0 else (other[1], dist)
                    for cand in ((dt - dw) % N_ORDER, (dt + dw) % N_ORDER, (-dt - dw) % N_ORDER, (dw - dt) % N_ORDER):
                        if cand <= IN_TXT_RANGE_END and hl.pubkey(cand)[1:] == (kx, ky):
                            found = cand
                    if found is not None:
                        break
            if found is not None:
                break
        eng.wait()
    assert found == IN_TXT_ANSWER
    assert launch < 200


@pytest.mark.parametrize("program", ["kangaroo_hip", "kangaroo_hip_ht", "kangaroo_mi355x"])
def test_reference_program_solves_in_txt_on_our_engine(tmp_path, program):
    """The unmodified reference program (oracle/_ref/kangaroo_hip = reference host code + our GPUEngine; kangaroo_hip_ht = the
    same with class HashTable replaced at link time; kangaroo_mi355x = with SolveKeyGPU replaced as well) solving its own
    shipped known-answer input on the MI355X."""
    import subprocess

    exe = ref_binary(program)
    cfg = tmp_path / "in.txt"
    cfg.write_text("0\n%X\n%s\n" % (IN_TXT_RANGE_END, IN_TXT_PUBKEY))
    out = subprocess.run([exe, "-t", "0", "-gpu", "-g", "16,128", str(cfg)], capture_output=True, text=True, timeout=300)
    assert "Priv: 0x%X" % IN_TXT_ANSWER in out.stdout, out.stdout[-2000:] + out.stderr[-500:]


@pytest.mark.parametrize("program", ["kangaroo_hip", "kangaroo_mi355x"])
def test_reference_program_drives_two_engines_concurrently(tmp_path, program):
    """The reference's own thread-per-engine path over the boundary (Kangaroo.cpp:1041-1047: one _SolveKeyGPU pthread per
    -gpuId entry): `-gpuId 0,0` makes the unmodified program create TWO GPUEngine instances and drive them from two
    host threads at once, feeding one HashTable.  SURVEY 8(b): distinct instances from distinct threads must work
    (every entry point selects its own device, nothing is global)."""
    import subprocess

    exe = ref_binary(program)
    cfg = tmp_path / "in.txt"
    cfg.write_text("0\n%X\n%s\n" % (IN_TXT_RANGE_END, IN_TXT_PUBKEY))
    out = subprocess.run([exe, "-t", "0", "-gpu", "-gpuId", "0,0", "-g", "32,128,32,128", str(cfg)], capture_output=True, text=True, timeout=300)
    assert out.stdout.count("GPU: GPU #0") == 2, out.stdout[-2000:]  # two engines were created
    assert "Priv: 0x%X" % IN_TXT_ANSWER in out.stdout, out.stdout[-2000:] + out.stderr[-500:]


@pytest.mark.parametrize("program", ["kangaroo_hip", "kangaroo_mi355x"])
def test_reference_program_solves_64bit_range_on_our_engine(tmp_path, program):
    """BASELINE.json configs[1]: the reference's shipped 64-bit known-answer input (VC_CUDA8/in64.txt,
    answer README.md:194-195) solved by the unmodified reference program on our engine."""
    import subprocess

    exe = ref_binary(program)
    cfg = tmp_path / "in64.txt"
    cfg.write_text("5B3F38AF935A3640D158E871CE6E9666DB862636383386EE0000000000000000\n"
                   "5B3F38AF935A3640D158E871CE6E9666DB862636383386EEFFFFFFFFFFFFFFFF\n"
                   "03BB113592002132E6EF387C3AEBC04667670D4CD40B2103C7D0EE4969E9FF56E4\n")
    out = subprocess.run([exe, "-t", "0", "-gpu", "-g", "64,128", str(cfg)], capture_output=True, text=True, timeout=900)
    assert "Priv: 0x5B3F38AF935A3640D158E871CE6E9666DB862636383386EE510F18CCC3BD72EB" in out.stdout, \
        out.stdout[-2000:] + out.stderr[-500:]


def _run_until_saves(cmd, n_saves, max_seconds):
    """Run the reference program unbuffered, return its output once `n_saves` work-file saves finished."""
    import select
    import shutil
    import subprocess
    import time

    if shutil.which("stdbuf"):
        cmd = ["stdbuf", "-o0", "-e0"] + cmd
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    fd = proc.stdout.fileno()
    buf = b""
    t0 = time.time()
    while time.time() - t0 < max_seconds:
        r, _, _ = select.select([fd], [], [], 0.5)
        if r:
            chunk = os.read(fd, 65536)
            if not chunk:
                break
            buf += chunk
        if buf.count(b"done [") >= n_saves or proc.poll() is not None:
            break
    proc.kill()
    proc.wait()
    return buf.decode(errors="replace")


@pytest.mark.parametrize("program", ["kangaroo_hip", "kangaroo_hip_ht", "kangaroo_mi355x"])
def test_reference_workfile_roundtrip_125bit_on_our_engine(tmp_path, orc, program):
    """BASELINE.json configs[4] on one GPU: 125-bit (maximum) range, `-ws -w f -wi 3` save through
    GetKangaroos, `-winfo` / `-wcheck` of the file, `-i f` restore through SetKangaroos -- all by the
    unmodified reference program (Backup.cpp, Check.cpp) on our engine -- plus an independent check of
    the saved kangaroos: every sampled (x, y, d) satisfies (x,y) = d*G (tame) / K + d*G (wild)."""
    import re
    import subprocess

    exe = ref_binary(program)
    cfg = tmp_path / "in125.txt"
    cfg.write_text("0\n1FFFFFFFFFFFFFFFFFFFFFFFFFFFFFFF\n%s\n" % IN_TXT_PUBKEY)
    f1, f2 = str(tmp_path / "a.work"), str(tmp_path / "b.work")
    nk = 16 * 128 * 128
    out = _run_until_saves([exe, "-t", "0", "-gpu", "-g", "16,128", "-d", "12", "-ws", "-w", f1, "-wi", "3", str(cfg)], 2, 120)
    assert out.count("done [") >= 1, out[-1500:]
    info = subprocess.run([exe, "-winfo", f1], capture_output=True, text=True, timeout=120).stdout
    assert re.search(r"Kangaroos\s*:\s*%d\b" % nk, info), info
    count1 = int(re.search(r"Count\s*:\s*(\d+)", info).group(1))
    chk = subprocess.run([exe, "-wcheck", f1], capture_output=True, text=True, timeout=600).stdout
    assert "100.000% OK" in chk, chk[-500:]

    # the kangaroo section is the tail of the file: u64 count, then count x (32 B x, 32 B y, 32 B d)  (Backup.cpp:525-546)
    raw = np.fromfile(f1, dtype=np.uint8)
    tail = raw[len(raw) - nk * 96:].reshape(nk, 96)
    assert int(np.frombuffer(raw[len(raw) - nk * 96 - 8:len(raw) - nk * 96].tobytes(), dtype=np.uint64)[0]) == nk
    recs = np.frombuffer(tail.tobytes(), dtype=np.uint64).reshape(nk, 12)
    kx, ky = _decompress(IN_TXT_PUBKEY)
    rng = np.random.default_rng(8)
    for i in rng.choice(nk, size=256, replace=False):
        x, y, d = recs[i, 0:4], recs[i, 4:8], np.ascontiguousarray(recs[i, 8:12])
        ox, oy = np.zeros(4, np.uint64), np.zeros(4, np.uint64)
        if i & 1:
            orc.lib.orc_pubkey_add(ox, oy, d, ints_to_array([kx])[0], ints_to_array([ky])[0])
        else:
            orc.lib.orc_pubkey(ox, oy, d)
        assert np.array_equal(ox, x) and np.array_equal(oy, y), f"saved kangaroo {i} is not at its distance"

    # restore (-i) and keep going: the count grows, the herd size stays, the new file checks again
    out = _run_until_saves([exe, "-t", "0", "-gpu", "-g", "16,128", "-i", f1, "-ws", "-w", f2, "-wi", "3"], 1, 120)
    assert "done [" in out, out[-1500:]
    info2 = subprocess.run([exe, "-winfo", f2], capture_output=True, text=True, timeout=120).stdout
    assert re.search(r"Kangaroos\s*:\s*%d\b" % nk, info2), info2
    assert int(re.search(r"Count\s*:\s*(\d+)", info2).group(1)) > count1
    chk2 = subprocess.run([exe, "-wcheck", f2], capture_output=True, text=True, timeout=600).stdout
    assert "100.000% OK" in chk2, chk2[-500:]


def test_standalone_cpp_gpuengine(tmp_path):
    """Our C++ `class GPUEngine` + Int.h + kng_host (no reference code) through the Check.cpp protocol."""
    import subprocess

    from kangaroo_amd.build import build_all

    build_all()
    host = os.path.join(ROOT, "kangaroo_amd", "host")
    lib = os.path.join(ROOT, "kangaroo_amd", "lib")
    exe = str(tmp_path / "test_gpuengine")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", host, os.path.join(ROOT, "tests", "cpp", "test_gpuengine.cpp"),
                           "-o", exe, "-L", lib, "-lkangaroo_host", "-lkangaroo_hip", "-Wl,-rpath," + lib, "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "CPP GPUEngine ok" in out.stdout, out.stdout[-1500:] + out.stderr[-500:]


@pytest.mark.parametrize("grid,lanes", [((3, 5), 448), ((2, 3), 320), ((4, 4), 1984)])
def test_ragged_groups_vs_oracle(kng, orc, grid, lanes):
    """Free lane count ("lanes" option): the herd does not divide evenly, so waves walk ceil or floor
    of N/lanes kangaroos.  Same bit-exact comparison with the oracle, two launches."""
    n = grid[0] * grid[1] * 128
    rp = 72
    x, y, true_d, wild_offset = _seeded_herd(orc, n, rp, seed=lanes)
    jd, jx, jy, _ = orc.jump_table(rp)
    mask = orc.dp_mask(5)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 1 << 16, lanes=lanes)
    assert eng.get_option("lanes") == lanes and n % lanes != 0
    eng.SetParams(mask, jd, jx, jy)
    eng.SetWildOffset(wild_offset)
    eng.SetKangaroos(x, y, ints_to_array(true_d))
    ox, oy = x.copy(), y.copy()
    od = ints_to_array(device_distances(true_d, wild_offset), 2)
    for _ in range(2):
        eng.callKernel()
        eng.wait()
        got = eng.drain(raw=True)
        want, total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 22)
        key = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
        assert sorted(map(key, got)) == sorted(map(key, want))
        gx, gy, gd = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gd, od)
    eng.close()


@pytest.mark.parametrize("share", [8])
def test_distance_low_word_streaming_vs_oracle(kng, orc, share):
    """Option "dsplit": only the low word of the 128-bit distance streams through HBM; the high word is
    read-modified-written when the add carries and fetched when a DP is emitted.  Forced on with jump distances
    just below 2^64 so that about every third jump carries (automatic mode would refuse), start distances
    straddling 2^64, dp 3: kangaroos, distances and DP records must still match the oracle bit for bit."""
    grid = (4, 4)
    n = grid[0] * grid[1] * 128
    rp = 72
    x, y, true_d, wild_offset = _seeded_herd(orc, n, rp, seed=4242)
    _, jx, jy, _ = orc.jump_table(rp)
    rng = np.random.default_rng(9)
    jd = np.zeros((32, 2), np.uint64)
    jd[:, 0] = rng.integers(1 << 61, (1 << 64) - 1, size=32, dtype=np.uint64)
    d = np.zeros((n, 2), np.uint64)
    d[:, 0] = rng.integers(0, (1 << 64) - 1, size=n, dtype=np.uint64)
    d[:, 1] = rng.integers(0, 1 << 40, size=n, dtype=np.uint64)
    d[::7, 1] = np.uint64((1 << 64) - 1)  # high word wraps too (raw 128-bit add, GPUMath.h:119-121)
    mask = orc.dp_mask(3)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 1 << 17, share=share, group=8, dsplit=1)
    eng.SetParams(mask, jd, jx, jy)
    assert eng.get_option("dsplit") == 1
    eng.SetKangaroos(x, y, d)
    ox, oy, od = x.copy(), y.copy(), d.copy()
    for _ in range(3):
        eng.callKernel()
        eng.wait()
        got = eng.drain(raw=True)
        want, total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 22)
        key = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
        assert len(got) == total and sorted(map(key, got)) == sorted(map(key, want))
        gx, gy, gd = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gd, od)
    eng.close()


@pytest.mark.parametrize("share,dsplit", [(8, 1), (8, 0)])
def test_exact_path_exits_of_the_scheduled_loop(kng, orc, share, dsplit):
    """The scheduled asm loop leaves an iteration to the general arithmetic (walk_core) when its short forms may not be
    exact; since round 3 it decides that from a SUPERSET of the conditions (a word below 1024 in a difference, a word within
    1024 of 2^32 in a product).  A herd seeded with kangaroos that raise those flags at every position of a pass -- x a few
    units beside its own jump point (tiny dx: difference words below 1024), x and y with limbs of all ones (borrow ripples,
    products with words near 2^32), first / middle / last of a lane's batch, several neighbours in a row -- must still
    equal the oracle bit for bit, DP multiset included, over three launches."""
    grid = (2, 4)
    n = grid[0] * grid[1] * 128
    rp = 72
    x, y, true_d, wild_offset = _seeded_herd(orc, n, rp, seed=777)
    jd, jx, jy, _ = orc.jump_table(rp)
    if not dsplit:
        jd = jd.copy()
        jd[:, 0] |= np.uint64(1 << 60)  # distances the engine will not stream low-word-only (>= 2^58)
    rng = np.random.default_rng(5)
    d = np.zeros((n, 2), np.uint64)  # raw device distances (no wild offset set: Set/Get pass them through)
    d[:, 0] = rng.integers(0, (1 << 63), size=n, dtype=np.uint64)
    d[:, 1] = rng.integers(0, 1 << 40, size=n, dtype=np.uint64)
    x, y = x.copy(), y.copy()
    jxi = array_to_ints(jx)
    G, L = 8, n // 8  # group 8: kangaroo g of lane t is index g * L + t
    crafted = 0
    for t_ in range(0, L, 3):
        for g in {0: (0,), 1: (G - 1,), 2: (3, 4), 3: (0, 1, 2, G - 1)}[(t_ // 3) % 4]:
            idx = g * L + t_
            kind = (idx // 7) % 3
            if kind == 0:  # a few units beside the jump point its own low bits select: dx = delta < 1024
                j = int(rng.integers(0, 32))
                delta = ((j - (jxi[j] & 31)) % 32) + 32 * int(rng.integers(1, 30))
                x[idx] = ints_to_array([(jxi[j] + delta) & ((1 << 256) - 1)], 4)[0]
                assert (int(x[idx][0]) & 31) == j
            elif kind == 1:  # limbs of all ones: every borrow ripples, products get words within 1024 of 2^32
                x[idx] = np.array([0xFFFFFFFFFFFFFC00 | int(rng.integers(0, 1024)), (1 << 64) - 1, (1 << 64) - 1, 0xFFFFFFFEFFFFFFFF], np.uint64)
                y[idx] = np.array([(1 << 64) - 1, 0xFFFFFFFF00000000, (1 << 64) - 1, (1 << 64) - 1], np.uint64)
            else:  # small values: differences wrap, low words of everything near zero
                x[idx] = np.array([int(rng.integers(1, 1 << 20)), 0, 0, 0], np.uint64)
                y[idx] = np.array([int(rng.integers(1, 1 << 20)), 0, 0, 0], np.uint64)
            crafted += 1
    assert crafted > n // 40
    mask = orc.dp_mask(4)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 1 << 17, share=share, group=G, asm=1)
    eng.SetParams(mask, jd, jx, jy)
    assert eng.get_option("dsplit") == dsplit and eng.get_option("asm") == 1
    eng.SetKangaroos(x, y, d)
    ox, oy, od = x.copy(), y.copy(), d.copy()
    for launch in range(3):
        eng.callKernel()
        eng.wait()
        got = eng.drain(raw=True)
        want, total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 22)
        key = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
        assert len(got) == total and sorted(map(key, got)) == sorted(map(key, want))
        gx, gy, gd = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gd, od)
        if launch == 0:  # the crafted kangaroos take the exact path in their first jump; afterwards they are ordinary points
            assert eng.get_option("exact_exits") >= 2
    eng.close()


def test_low_word_carries_accumulate_over_many_launches_at_109_bits(kng, orc):
    """BASELINE configs[3]'s table (109-bit range, jump distances ~2^54) on the default kernel: the low word of a lane's
    distance carries every ~700 jumps, and each carry is an L2 atomic on the high word from inside the scheduled loop.  50
    launches = 3200 jumps of 131 072 kangaroos (~4.5 carries per kangaroo, ~600 000 atomics, DP records in between fetch the
    high words coherently): every (x, y, d) and the complete DP multiset equal the oracle's walk of the same herd."""
    import kangaroo_amd.hostlib as hl

    rp, gx, gy, dp, launches = 109, 8, 128, 12, 50
    n = gx * gy * 128
    _, kx, ky = hl.pubkey((1 << 108) + 0xC0FFEE123456789ABCD)
    jd, jx, jy, _ = hl.jump_table(rp)
    mask = hl.dp_mask(dp)
    got = []
    with kng.GPUEngine(gx, gy, 0, 1 << 17) as eng:
        eng.SetParams(mask, jd, jx, jy)
        assert eng.get_option("dsplit") == 1 and eng.get_option("asm") == 1 and eng.get_option("share") == 8
        eng.CreateHerdOnDevice(rp, (kx, ky), seed=0x109)
        x0, y0, d0 = eng.GetKangaroos(raw=True)
        for _ in range(launches):
            eng.callKernel()
            eng.wait()
            got.append(eng.drain(raw=True))
            assert eng.lastLost == 0
        x1, y1, d1 = eng.GetKangaroos(raw=True)
    got = np.concatenate(got)
    hi_before = d0[:, 1].copy()
    want = orc.walk_parallel(x0, y0, d0, 64 * launches, jd, jx, jy, mask)
    assert np.array_equal(x1, x0) and np.array_equal(y1, y0) and np.array_equal(d1, d0)  # x0.. now hold the oracle's end state
    carries = int((d0[:, 1] - hi_before).sum())
    assert 2.5 * n < carries < 7 * n, carries  # jD ~ 2^54 of 2^64 per jump, 3200 jumps
    assert len(got) == len(want)

    def canon(r):
        o = np.lexsort((r["d"][:, 1], r["d"][:, 0], r["kidx"]))
        return r["kidx"][o], r["x"][o], r["d"][o]

    for a, b in zip(canon(got), canon(want)):
        assert np.array_equal(a, b)


def test_distance_low_word_streaming_is_chosen_by_the_jump_table(kng, orc):
    eng = kng.GPUEngine(2, 2, 0, 1 << 12)
    # jump distances < 2^(rp/2+1).  Scheduled loop (carries added in the loop by L2 atomics): automatic below 2^58, i.e. up to
    # 115-bit ranges; compiler-scheduled loop (divergent read-modify-write): below 2^50
    for use_asm, cases in ((1, ((72, 1), (98, 1), (100, 1), (109, 1), (115, 1), (116, 0), (125, 0))), (0, ((72, 1), (98, 1), (100, 0), (125, 0)))):
        eng.set_option("asm", use_asm)
        for rp, want in cases:
            jd, jx, jy, _ = orc.jump_table(rp)
            eng.SetParams(orc.dp_mask(8), jd, jx, jy)
            assert eng.get_option("dsplit") == want, (use_asm, rp)
    eng.set_option("asm", 1)
    jd, jx, jy, _ = orc.jump_table(72)
    jd = jd.copy()
    jd[5, 1] = 1  # a high word in the table: never, even when forced
    eng.set_option("dsplit", 1)
    eng.SetParams(orc.dp_mask(8), jd, jx, jy)
    assert eng.get_option("dsplit") == 0
    eng.set_option("dsplit", 0)
    jd, jx, jy, _ = orc.jump_table(72)
    eng.SetParams(orc.dp_mask(8), jd, jx, jy)
    assert eng.get_option("dsplit") == 0
    eng.close()


def test_ranged_set_get_of_the_herd(kng):
    """kng_set_kangaroos_range / kng_get_kangaroos_range: the herd uploaded in uneven slices (crossing the 64 K
    staging chunk) equals a whole-herd upload, slices read back equal the whole-herd read, bad ranges are refused."""
    gx, gy = 40, 16  # 81 920 kangaroos: more than one staging chunk
    n = gx * gy * 128
    rng = np.random.default_rng(77)
    x = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    y = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    x[:, 3] >>= np.uint64(2)
    y[:, 3] >>= np.uint64(2)
    d = rng.integers(0, 1 << 64, size=(n, 2), dtype=np.uint64)
    eng = kng.GPUEngine(gx, gy, 0, 1 << 16)
    cuts = [0, 1, 4097, 65536, 65537, 70001, n]
    with pytest.raises(kng.EngineError):
        eng.GetKangaroosRange(0, 16)  # nothing loaded yet
    for a, b in zip(cuts[:-1], cuts[1:]):
        eng.SetKangaroosRange(a, x[a:b], y[a:b], d[a:b])
    gx_, gy_, gd_ = eng.GetKangaroos(raw=True)
    assert np.array_equal(gx_, x) and np.array_equal(gy_, y) and np.array_equal(gd_, d)
    for a, b in ((0, 5), (65530, 65550), (n - 3, n), (12345, 12345)):
        sx, sy, sd = eng.GetKangaroosRange(a, b - a)
        assert np.array_equal(sx, x[a:b]) and np.array_equal(sy, y[a:b]) and np.array_equal(sd, d[a:b])
    with pytest.raises(kng.EngineError, match="outside the herd"):
        eng.GetKangaroosRange(n - 2, 3)
    with pytest.raises(kng.EngineError, match="outside the herd"):
        eng.SetKangaroosRange(n, x[:1], y[:1], d[:1])
    eng.close()


@pytest.mark.parametrize("share", [8, 4])
@pytest.mark.parametrize("grid,opt", [((4, 4), dict(group=4)), ((4, 4), dict(group=128)), ((3, 5), dict(lanes=448)),
                                      ((2, 3), dict(lanes=320)), ((8, 4), dict(group=2))])
def test_shared_inversion_vs_oracle(kng, orc, grid, opt, share):
    """Option "share": the eight waves of a 512-thread block (or the four of a 256-thread block) invert the product of their lane
    chains once (two-level / one-level tree).  Covers full blocks, a partner wave without work (lanes % 512 != 0) and ragged
    groups; three launches."""
    n = grid[0] * grid[1] * 128
    rp = 72
    x, y, true_d, wild_offset = _seeded_herd(orc, n, rp, seed=n + 7)
    jd, jx, jy, _ = orc.jump_table(rp)
    mask = orc.dp_mask(5)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 1 << 16, share=share, **opt)
    assert eng.get_option("share") == share
    eng.SetParams(mask, jd, jx, jy)
    eng.SetWildOffset(wild_offset)
    eng.SetKangaroos(x, y, ints_to_array(true_d))
    ox, oy = x.copy(), y.copy()
    od = ints_to_array(device_distances(true_d, wild_offset), 2)
    for _ in range(3):
        eng.callKernel()
        eng.wait()
        got = eng.drain(raw=True)
        want, total = orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 22)
        key = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
        assert sorted(map(key, got)) == sorted(map(key, want))
        gx, gy, gd = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gd, od)
    eng.close()


@pytest.mark.parametrize("rp,grid,group", [(72, (2, 2), 16), (125, (2, 3), 64), (20, (1, 2), 1), (64, (4, 4), 128)])
def test_device_herd_creation(kng, orc, rp, grid, group):
    """the herd is built on the GPU (kng_build_herd).  Every kangaroo must sit at
    d*G (tame) / K + d*G (wild) for the distance the engine reports, distances must be in range and
    well spread, and the herd must walk exactly like an uploaded one (bit-exact against the oracle)."""
    import kangaroo_amd.hostlib as hl

    n = grid[0] * grid[1] * 128
    key = (0xC0FFEE << 40) | 0x777
    _, kx, ky = orc.pubkey(key)
    jd, jx, jy, _ = orc.jump_table(rp)
    mask = orc.dp_mask(4)
    with kng.GPUEngine(grid[0], grid[1], 0, 1 << 16, group=group) as eng:
        woff = eng.CreateHerdOnDevice(rp, (kx, ky), seed=99)
        assert woff == ((1 << rp) - 1) >> 1
        eng.SetParams(mask, jd, jx, jy)
        px, py, d_true = eng.GetKangaroos()
        _, _, d_dev = eng.GetKangaroos(raw=True)
        dv = array_to_ints(d_dev)
        assert all(1 <= v < (1 << rp) for v in dv)
        assert len(set(dv)) > 0.99 * min(n, 1 << (rp - 1))          # no stuck generator
        if rp >= 32:
            assert abs(sum(v >> (rp - 8) for v in dv) / n - 127.5) < 12  # top byte roughly uniform
        ox, oy = orc.create_herd(d_true, 0, kx, ky)                  # tame d*G / wild K + d*G
        assert np.array_equal(ox, px) and np.array_equal(oy, py)
        # and it walks like any other herd
        eng.callKernel()
        eng.wait()
        got = eng.drain(raw=True)
        gx, gy, gd = eng.GetKangaroos(raw=True)
    wx, wy, wd = px.copy(), py.copy(), d_dev.copy()
    want, total = orc.walk(wx, wy, wd, 64, jd, jx, jy, mask, dp_cap=1 << 22)
    assert np.array_equal(gx, wx) and np.array_equal(gy, wy) and np.array_equal(gd, wd)
    key_ = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
    assert sorted(map(key_, got)) == sorted(map(key_, want))
    # a different seed gives a different herd, the same seed the same herd
    with kng.GPUEngine(grid[0], grid[1], 0, 16, group=group) as e2:
        e2.CreateHerdOnDevice(rp, (kx, ky), seed=99)
        a = e2.GetKangaroos(raw=True)
        e2.CreateHerdOnDevice(rp, (kx, ky), seed=100)
        b = e2.GetKangaroos(raw=True)
    assert np.array_equal(a[0], px) and not np.array_equal(b[2], d_dev)


@pytest.mark.parametrize("use_asm,rp", [(1, 72), (0, 72), (1, 125)])
def test_products_carried_from_launch_to_launch(kng, orc, use_asm, rp):
    """Round 4: a launch leaves the prefix products of the NEXT jump's dx in the product planes and the following launch
    starts from them instead of recomputing them (WalkArgs.resume) -- unless anything touched the herd, the jump table, the
    geometry or the planes in between.  Every such event, and an odd number of steps (products left in descending order),
    must bring the product pass back; the state and the DP multiset after every launch equal the oracle's."""
    grid, G = (4, 8), 16
    n = grid[0] * grid[1] * 128
    x, y, true_d, wild_offset = _seeded_herd(orc, n, rp, seed=31337 + rp)
    jd, jx, jy, _ = orc.jump_table(rp)
    jd2, jx2, jy2, _ = orc.jump_table(rp - 8)
    mask = orc.dp_mask(5)
    eng = kng.GPUEngine(grid[0], grid[1], 0, 1 << 17, group=G, asm=use_asm)
    eng.SetParams(mask, jd, jx, jy)
    eng.SetWildOffset(wild_offset)
    eng.SetKangaroos(x, y, ints_to_array(true_d))
    ox, oy = x.copy(), y.copy()
    od = ints_to_array(device_distances(true_d, wild_offset), 2)
    key = lambda r: (int(r["kidx"]), tuple(int(v) for v in r["x"]), tuple(int(v) for v in r["d"]))  # noqa: E731
    tab = [jd, jx, jy]

    def launch_and_compare(what, steps=64):
        eng.callKernel()
        eng.wait()
        got = eng.drain(raw=True)
        want, total = orc.walk(ox, oy, od, steps, tab[0], tab[1], tab[2], mask, dp_cap=1 << 22)
        assert len(got) == total and sorted(map(key, got)) == sorted(map(key, want)), what
        gx, gy, gd = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx, ox) and np.array_equal(gy, oy) and np.array_equal(gd, od), what

    launch_and_compare("first launch: product pass")
    launch_and_compare("second launch: resumes")
    launch_and_compare("third launch: resumes")
    # one kangaroo replaced between launches (kng_set_kangaroo): its dx changed, the stored products are stale
    k = 4 * G + 3
    nx, ny, nd = x[k ^ 2].copy(), y[k ^ 2].copy(), od[k ^ 2].copy()  # a valid point of the same type (k ^ 2 keeps parity)
    ox[k], oy[k], od[k] = nx, ny, nd
    from helpers import from_limbs

    eng.SetKangaroo(k, from_limbs(nx), from_limbs(ny), host_distance(from_limbs(nd), k, wild_offset))
    launch_and_compare("after SetKangaroo")
    launch_and_compare("resumes again")
    # the audit borrows the product planes
    _, kx, ky = orc.pubkey(1)
    eng.audit_setup((kx, ky))
    eng.audit_herd(cap=0)  # (mismatches are expected: the herd's key is not 1*G; only the planes matter here)
    launch_and_compare("after an audit")
    # an odd number of steps leaves the products in descending order
    eng.set_option("steps", 33)
    launch_and_compare("33 steps", 33)
    launch_and_compare("33 steps again", 33)
    eng.set_option("steps", 64)
    launch_and_compare("back to 64")
    launch_and_compare("resumes")
    # another jump table
    tab[:] = [jd2, jx2, jy2]
    eng.SetParams(mask, jd2, jx2, jy2)
    launch_and_compare("new table")
    launch_and_compare("resumes with the new table")
    eng.close()
