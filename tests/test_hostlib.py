"""CPU tests of the product-side host library (kangaroo_amd/host) against the oracle and the
vectors produced by the reference's own objects."""
import numpy as np
import pytest

from helpers import N_ORDER, P, array_to_ints, ints_to_array

H = lambda s: int(s, 16)  # noqa: E731


@pytest.fixture(scope="module")
def hl():
    from kangaroo_amd.build import build_all

    build_all()
    from kangaroo_amd import hostlib

    hostlib.load()
    return hostlib


def test_pubkey_matches_reference(hl, golden):
    for k, x, y in golden["pubkey"]:
        assert hl.pubkey(H(k)) == (0, H(x), H(y))
    assert hl.pubkey(0)[0] == -1 and hl.pubkey(N_ORDER)[0] == -1


@pytest.mark.parametrize("rp", ["32", "56", "64", "80", "109", "125"])
def test_jump_table_matches_reference(hl, golden, rp):
    t = golden["jump_tables"][rp]
    jd, jx, jy, avg = hl.jump_table(int(rp))
    assert array_to_ints(jd) == [H(v) for v in t["jd"]]
    assert array_to_ints(jx) == [H(v) for v in t["jx"]]
    assert array_to_ints(jy) == [H(v) for v in t["jy"]]


def test_order_arithmetic(hl, golden):
    lib = hl.load()
    for a, b, s, d in golden["order"]:
        r = np.zeros(4, np.uint64)
        lib.kngh_add_order(hl.limbs(H(a)), hl.limbs(H(b)), r)
        assert hl.to_int(r) == H(s)
        lib.kngh_sub_order(hl.limbs(H(a)), hl.limbs(H(b)), r)
        assert hl.to_int(r) == H(d)


def test_dp_mask_and_suggest_dp(hl, orc):
    for dp in (0, 1, 8, 14, 25, 33, 63, 64, 70):
        assert hl.dp_mask(dp) == orc.dp_mask(dp)
    # 2^23 kangaroos on 80 bits -> 14 ; 2^26 on 109 bits -> 25 ; 2^26 on 125 bits -> 33
    assert hl.suggest_dp(80, 2**23) == 14
    assert hl.suggest_dp(109, 2**26) == 25
    assert hl.suggest_dp(125, 2**26) == 33


@pytest.mark.parametrize("rp,n,threads", [(64, 1000, 1), (80, 5000, 3), (125, 2049, 0), (20, 600, 2)])
def test_create_herd_is_valid(hl, orc, rp, n, threads):
    key = (0xC0FFEE << 40) | 12345
    _, kx, ky = orc.pubkey(key)
    x, y, d, woff = hl.create_herd(n, rp, (kx, ky), first_type=0, seed=42, nthreads=threads)
    assert woff == ((1 << rp) - 1) >> 1
    # every kangaroo sits where its distance says: tame = d*G, wild = K + d*G  (Kangaroo.cpp:707-725)
    ox, oy = orc.create_herd(d, 0, kx, ky)
    assert np.array_equal(ox, x) and np.array_equal(oy, y)
    dv = array_to_ints(d)
    for i in range(0, n, 97):
        if i & 1:
            signed = dv[i] if dv[i] < (1 << 200) else dv[i] - N_ORDER
            assert -woff - 1 <= signed <= woff + 1
        else:
            assert 0 <= dv[i] < (1 << rp)
        assert hl.on_curve(*array_to_ints([x[i], y[i]]))
    # deterministic in the seed and independent of the thread count
    x2, y2, d2, _ = hl.create_herd(n, rp, (kx, ky), first_type=0, seed=42, nthreads=1)
    assert np.array_equal(x, x2) and np.array_equal(d, d2)
    # device <-> true distance mapping (GPUEngine.cu:409,477)
    dev = hl.to_device_distances(d, woff)
    assert np.array_equal(hl.to_true_distances(dev, woff), d)
    assert all(int(v) < (1 << 64) for v in dev[:, 1])


def test_point_add(hl, orc):
    _, gx, gy = orc.pubkey(1)
    _, x2, y2 = orc.pubkey(2)
    _, x3, y3 = orc.pubkey(3)
    assert hl.point_add((gx, gy), (gx, gy)) == (0, x2, y2)  # doubling
    assert hl.point_add((gx, gy), (x2, y2)) == (0, x3, y3)
    assert hl.point_add((gx, gy), (gx, P - gy))[0] == -1  # P + (-P) = infinity


def test_telemetry_degrades_to_unavailable_without_a_device():
    """kangaroo_amd/telemetry.py (package power / GFX clock for the bench line) must never raise: on a box without an
    AMD GPU it reports itself unavailable, with a reason, and the sampler still measures its window."""
    import time

    from kangaroo_amd.telemetry import GpuSampler, power_cap_w

    with GpuSampler([0], hz=100.0) as s:
        time.sleep(0.05)
    out = s.summary()
    assert isinstance(out, dict) and "available" in out
    if not out["available"]:
        assert out["reason"] and power_cap_w(0) is None
    else:  # a GPU box: the fields the bench line carries
        dev = out["devices"][0]
        assert dev["device"] == 0 and out["window_s"] >= 0.05 and dev["power_cap_w"]


def test_bench_power_bound_arithmetic():
    """bench.py's second bound: (P_cap - P_static) / E_dyn_per_jump from the recorded instruction and byte energies."""
    import bench

    fake = {"available": True, "devices": [{"device": 0, "power_cap_w": 1400.0, "power_w": {"median": 1360.0}}]}
    pb = bench._power_bound(fake, 208.0, 25000.0)
    e_dyn = (410 * 1.38 + 475 * 0.75 + 140 * 0.36) / 64 + 208.0 * 0.1
    assert abs(pb["e_dyn_nj_per_jump"] - e_dyn) < 0.01
    assert abs(pb["value_mks"] - (1400.0 - 342.0) / (e_dyn * 1e-9) / 1e6) < 1.0
    assert abs(pb["measured_nj_per_jump"] - 1360.0 / 25000e6 * 1e9) < 0.01
    assert 0.8 < pb["frac_of_bound_at_measured_power"] < 1.0
    assert "value_mks" not in bench._power_bound({"available": False}, 208.0, 25000.0)
    assert bench.effective_cpus() >= 1.0
