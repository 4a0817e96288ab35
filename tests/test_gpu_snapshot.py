"""Work-file snapshot of the herd (kng_snapshot*): the kangaroo section of a work file -- 96-byte records
{x, y, true distance mod n}, Backup.cpp:525-546 / :211-231 -- packed and unpacked on the device.

Checked bit-exactly against what the reference computes on the host for the same bytes: GPUEngine::GetKangaroos removes the
wild offset with Int::ModSubK1order (GPUEngine.cu:477), SetKangaroos adds it with ModAddK1order (:406-409); the oracle's
mod-n add / sub are pinned against the reference's own objects (tests/golden/ref_vectors.json, test_oracle_golden.py).
"""
import threading

import numpy as np
import pytest

from helpers import M128, N_ORDER, array_to_ints, device_distances, ints_to_array

pytestmark = pytest.mark.gpu


def _random_state(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    y = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    d = rng.integers(0, 1 << 64, size=(n, 2), dtype=np.uint64)
    return rng, x, y, d


def _true_from_device(orc, dd, woff):
    """what GetKangaroos hands the program: (dd - woff) mod n for odd kIdx, by the oracle's pinned mod-n subtraction"""
    n = dd.shape[0]
    d4 = np.zeros((n, 4), dtype=np.uint64)
    d4[:, :2] = dd
    vals = array_to_ints(d4[1::2])
    d4[1::2] = ints_to_array([orc.sub_order(v, woff) for v in vals])
    return d4


@pytest.mark.parametrize("woff", [(1 << 79) - 1 >> 0, (1 << 124), 0, N_ORDER - 5])
def test_snapshot_records_are_the_work_file_bytes(kng, orc, woff):
    gx, gy = 37, 3  # 14 208 kangaroos: not a multiple of the pack kernel's block
    n = gx * gy * 128
    rng, x, y, dd = _random_state(n, 5)
    # the corners of the mod-n subtraction: below the offset (wraps through n), equal to it, all ones, zero
    w2 = woff & M128
    for i, v in ((1, 0), (3, max(w2 - 1, 0)), (5, w2), (7, M128), (9, (w2 + 1) & M128), (0, M128), (2, 0)):
        dd[i] = ints_to_array([v], 2)[0]
    with kng.GPUEngine(gx, gy, 0, 1 << 16) as eng:
        with pytest.raises(kng.EngineError, match="no herd"):
            eng.Snapshot()
        eng.SetKangaroos(x, y, dd)
        eng.SetWildOffset(woff)
        with pytest.raises(kng.EngineError, match="no snapshot"):
            eng.SnapshotRead(0, 4)
        mem0 = eng.GetMemory()
        eng.Snapshot()
        assert eng.GetMemory() == mem0 + 96 * n  # the second buffer is part of GetMemory()
        rec = eng.SnapshotRead()
        assert rec.shape == (n, 12)
        assert np.array_equal(rec[:, 0:4], x) and np.array_equal(rec[:, 4:8], y)
        want = _true_from_device(orc, dd, woff) if woff else np.concatenate([dd, np.zeros((n, 2), np.uint64)], axis=1)
        assert np.array_equal(rec[:, 8:12], want)
        # ... and by plain integers, for the corners
        for i in (0, 1, 2, 3, 5, 7, 9):
            v = int(dd[i, 0]) | int(dd[i, 1]) << 64
            assert array_to_ints(rec[i:i + 1, 8:12])[0] == ((v - woff) % N_ORDER if (i & 1 and woff) else v), i
        # slices, an empty one, a refused one
        for a, b in ((0, 1), (4095, 4099), (n - 1, n), (77, 77)):
            assert np.array_equal(eng.SnapshotRead(a, b - a), rec[a:b])
        with pytest.raises(kng.EngineError, match="outside the herd"):
            eng.SnapshotRead(n - 1, 2)
        # without the offset: the zero-extended device distances
        eng.Snapshot(with_offset=False)
        raw = eng.SnapshotRead()
        assert np.array_equal(raw[:, 8:10], dd) and not raw[:, 10:12].any()
        eng.SnapshotRelease()
        assert eng.GetMemory() == mem0
        with pytest.raises(kng.EngineError, match="no snapshot"):
            eng.SnapshotRead(0, 1)


def test_snapshot_is_frozen_between_two_launches_while_the_walk_goes_on(kng, orc):
    """wait L, snapshot, launch L+1, read the snapshot from ANOTHER thread while L+1 runs: the records are the state after L
    (oracle), and L+1 is not disturbed (state after it = oracle)."""
    from test_gpu_parity import _seeded_herd

    gx, gy, rp = 16, 8, 72  # 16 384 kangaroos
    n = gx * gy * 128
    x, y, true_d, woff = _seeded_herd(orc, n, rp, seed=61)
    jd, jx, jy, _ = orc.jump_table(rp)
    mask = orc.dp_mask(6)
    with kng.GPUEngine(gx, gy, 0, 1 << 16) as eng:
        eng.SetParams(mask, jd, jx, jy)
        eng.SetWildOffset(woff)
        eng.SetKangaroos(x, y, ints_to_array(true_d))
        ox, oy = x.copy(), y.copy()
        od = ints_to_array(device_distances(true_d, woff), 2)
        eng.callKernel()
        eng.wait()
        eng.drain(raw=True)
        orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 22)
        sx, sy, sd = ox.copy(), oy.copy(), od.copy()  # the state the snapshot must hold
        eng.Snapshot()
        eng.set_option("steps", 64 * 20)  # a long launch: the reader overlaps it
        eng.callKernel()
        got = {}

        def reader():
            parts = [eng.SnapshotRead(a, min(4096, n - a)) for a in range(0, n, 4096)]
            got["rec"] = np.concatenate(parts)
            got["still_running"] = bool(kng.load_library().kng_outstanding(eng._h))

        th = threading.Thread(target=reader)
        th.start()
        th.join()
        eng.wait()
        rec = got["rec"]
        assert np.array_equal(rec[:, 0:4], sx) and np.array_equal(rec[:, 4:8], sy)
        assert np.array_equal(rec[:, 8:12], _true_from_device(orc, sd, woff))
        # every saved kangaroo satisfies the invariant a work file promises: (x, y) = d*G (tame) / K + d*G (wild), spot-checked
        orc.walk(ox, oy, od, 64 * 20, jd, jx, jy, mask, dp_cap=1 << 24)
        gx_, gy_, gd_ = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx_, ox) and np.array_equal(gy_, oy) and np.array_equal(gd_, od)


def test_restore_from_records(kng, orc):
    """records -> kng_snapshot_write (uneven slices) -> kng_snapshot_restore = SetKangaroos of the same kangaroos; a distance
    that does not fit the device is an error with its index; a restored herd walks like an uploaded one."""
    from test_gpu_parity import _seeded_herd

    gx, gy, rp = 5, 6, 125
    n = gx * gy * 128
    x, y, true_d, woff = _seeded_herd(orc, n, rp, seed=62)
    rec = np.concatenate([x, y, ints_to_array(true_d)], axis=1)
    jd, jx, jy, _ = orc.jump_table(rp)
    mask = orc.dp_mask(5)
    dev = ints_to_array(device_distances(true_d, woff), 2)
    with kng.GPUEngine(gx, gy, 0, 1 << 16) as eng:
        eng.SetParams(mask, jd, jx, jy)
        eng.SetWildOffset(woff)
        with pytest.raises(kng.EngineError, match="nothing uploaded"):
            eng.SnapshotRestore()
        cuts = [0, 1, 130, 2049, n]
        for a, b in zip(cuts[:-1], cuts[1:]):
            eng.SnapshotWrite(a, rec[a:b])
        with pytest.raises(kng.EngineError, match="no herd"):
            eng.GetKangaroos(raw=True)
        eng.SnapshotRestore(0, 2049)  # the head first: not yet a herd
        with pytest.raises(kng.EngineError, match="no herd"):
            eng.GetKangaroos(raw=True)
        eng.SnapshotRestore(2049, n - 2049)
        gx_, gy_, gd_ = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx_, x) and np.array_equal(gy_, y) and np.array_equal(gd_, dev)
        # walks like an uploaded herd
        ox, oy, od = x.copy(), y.copy(), dev.copy()
        eng.callKernel()
        eng.wait()
        eng.drain(raw=True)
        orc.walk(ox, oy, od, 64, jd, jx, jy, mask, dp_cap=1 << 22)
        gx_, gy_, gd_ = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx_, ox) and np.array_equal(gy_, oy) and np.array_equal(gd_, od)
        # pack(unpack(records)) = records
        eng.SnapshotWrite(0, rec)
        eng.SnapshotRestore()
        eng.Snapshot()
        assert np.array_equal(eng.SnapshotRead(), rec)
        # distances that cannot live in 128 bits on the device: a tame one with a high limb, a wild one whose sum overflows
        bad = rec.copy()
        bad[10, 10] = 1
        bad[33, 8:12] = ints_to_array([(M128 - woff + 1) % N_ORDER])[0]
        eng.SnapshotWrite(0, bad)
        with pytest.raises(kng.EngineError, match=r"2 restored distances do not fit.*kangaroo 10"):
            eng.SnapshotRestore()
        # the largest wild distance that still fits
        ok = rec.copy()
        ok[33, 8:12] = ints_to_array([(M128 - woff) % N_ORDER])[0]
        eng.SnapshotWrite(0, ok)
        eng.SnapshotRestore()
        _, _, gd_ = eng.GetKangaroosRange(33, 1)
        assert array_to_ints(np.concatenate([gd_, np.zeros((1, 2), np.uint64)], axis=1))[0] == M128


def test_snapshot_of_the_default_herd(kng):
    """2^23 kangaroos: pack kernel + 805 MB read in pieces, timed (printed); unpack restores the identical planes"""
    import time

    gx, gy = 512, 128
    n = gx * gy * 128
    rng, x, y, dd = _random_state(n, 9)
    woff = (1 << 79) - 1
    with kng.GPUEngine(gx, gy, 0, 1 << 16) as eng:
        eng.SetKangaroos(x, y, dd)
        eng.SetWildOffset(woff)
        eng.Snapshot()  # allocates
        t0 = time.perf_counter()
        eng.Snapshot()
        first = eng.SnapshotRead(0, 1)
        t1 = time.perf_counter()
        rec = eng.SnapshotRead()
        t2 = time.perf_counter()
        print(f"\nsnapshot of 2^23 kangaroos: pack + first record {1e3 * (t1 - t0):.2f} ms, read of {rec.nbytes / 1e6:.0f} MB into pageable memory {t2 - t1:.3f} s")
        assert np.array_equal(rec[0], first[0])
        assert np.array_equal(rec[:, 0:4], x) and np.array_equal(rec[:, 4:8], y)
        assert np.array_equal(rec[0::2, 8:10], dd[0::2]) and not rec[0::2, 10:12].any()
        k = np.arange(1, n, 2)[:: 4099]
        for i in k[:200]:
            v = int(dd[i, 0]) | int(dd[i, 1]) << 64
            assert array_to_ints(rec[i:i + 1, 8:12])[0] == (v - woff) % N_ORDER
        eng.SetKangaroos(y, x, dd[::-1].copy())  # something else
        eng.SnapshotWrite(0, rec)
        eng.SnapshotRestore()
        gx_, gy_, gd_ = eng.GetKangaroos(raw=True)
        assert np.array_equal(gx_, x) and np.array_equal(gy_, y) and np.array_equal(gd_, dd)
