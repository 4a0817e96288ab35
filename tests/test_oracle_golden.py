"""Pin the oracle (oracle/kng_oracle.c) against vectors produced by the reference's own objects.

tests/golden/ref_vectors.json is written by oracle/refprobe.cpp, which links the unmodified
reference SECPK1 + Kangaroo objects (tools/make_golden.py).  An independent pure-Python
restatement (pow(x,-1,p), textbook affine addition) cross-checks both.
"""
import os

import numpy as np
import pytest

from helpers import (M128, N_ORDER, P, array_to_ints, device_distances, dp_multiset, host_distance,
                     ints_to_array, walk_fixture)

H = lambda s: int(s, 16)  # noqa: E731
M256 = (1 << 256) - 1


def test_modmul_matches_reference(golden, orc):
    for a, b, r in golden["modmul"]:
        assert orc.modmul(H(a), H(b)) == H(r)


def test_modmul_is_the_lazy_fold(golden, orc):
    """Results are NOT always canonical, and for extreme operands not even congruent: the fold
    drops the last carry and never compares with p (IntMod.cpp:944 'very very unlikely').
    The engine must reproduce exactly this, so pin the behaviour."""
    non_canonical = dropped_carry = 0
    for a, b, r in golden["modmul"]:
        r = H(r)
        assert r <= M256
        diff = (H(a) * H(b) - r) % P
        assert diff in (0, (1 << 256) % P)
        dropped_carry += diff != 0
        non_canonical += r >= P
    # the edge operands (p, p+1, 2^256-1 ...) do exercise both corners
    assert non_canonical > 0 and dropped_carry > 0


def test_modsqr_matches_reference(golden, orc):
    for a, r in golden["modsqr"]:
        assert orc.modsqr(H(a)) == H(r)
        assert orc.modsqr(H(a)) == orc.modmul(H(a), H(a))  # GPUEngine.cu:76-79 GPU_CHECK identity


def test_modsub_matches_reference_low_256(golden, orc):
    # the reference Int is 320-bit signed; the device works on 256 bits (GPUMath.h:476-494):
    # compare the low 256 bits (limb 4 of the reference is only a sign extension)
    for a, b, r in golden["modsub"]:
        assert orc.modsub(H(a), H(b)) == H(r) & M256


def test_modinv_matches_reference_and_python(golden, orc):
    for a, r in golden["modinv"]:
        got = orc.modinv(H(a))
        assert got == H(r)
        if H(a) % P:
            assert got == pow(H(a), -1, P)
        else:
            assert got == 0


def test_batch_inverse_matches_reference(golden, orc):
    bi = golden["batch_inv"]
    assert orc.batch_inv([H(v) for v in bi["in"]]) == [H(v) for v in bi["out"]]


def test_batch_inverse_zero_poisons_batch(orc):
    # IntGroup.cpp:41-56: a zero anywhere zeroes every output
    vals = [3, 5, 0, 7]
    assert orc.batch_inv(vals) == [0, 0, 0, 0]


def test_order_arithmetic_matches_reference(golden, orc):
    for a, b, s, d in golden["order"]:
        assert orc.add_order(H(a), H(b)) == H(s)
        assert orc.sub_order(H(a), H(b)) == H(d)
        assert H(s) == (H(a) + H(b)) % N_ORDER
        assert H(d) == (H(a) - H(b)) % N_ORDER


def test_rng_matches_reference(golden, orc):
    r = golden["rand"]
    orc.rseed(r["seed"])
    assert [orc.rndl() for _ in range(8)] == r["first_rndl"]
    for nbit, v in r["int_rand"]:
        assert orc.int_rand(nbit) == H(v)


def test_pubkey_matches_reference(golden, orc):
    for k, x, y in golden["pubkey"]:
        rc, gx, gy = orc.pubkey(H(k))
        assert rc == 0 and (gx, gy) == (H(x), H(y))
        assert (gy * gy - gx * gx * gx - 7) % P == 0


@pytest.mark.parametrize("rp", ["32", "56", "64", "80", "109", "125"])
def test_jump_table_matches_reference(golden, orc, rp):
    t = golden["jump_tables"][rp]
    jd, jx, jy, avg = orc.jump_table(int(rp))
    assert array_to_ints(jd) == [H(v) for v in t["jd"]]
    assert array_to_ints(jx) == [H(v) for v in t["jx"]]
    assert array_to_ints(jy) == [H(v) for v in t["jy"]]
    jump_bit = min(128, int(rp) // 2 + 1)
    assert jump_bit - 1.05 < avg < jump_bit - 0.95


def test_survey_appendix_c_vectors(orc):
    """values produced from the reference objects during the survey."""
    jd, jx, jy, _ = orc.jump_table(80)
    assert array_to_ints(jd)[:3] == [0xB2B5FF2560, 0x1E3893C08DF, 0x94FBEE16A6]
    assert array_to_ints(jx)[0] == 0x9F6CAE9F006747AF8588F0C350FE1153E0918DE240E00DB54C47FFB285D3F169
    d0 = 0x123456789ABCDEF
    rc, x, y = orc.pubkey(d0)
    xs, ys, ds = ints_to_array([x]), ints_to_array([y]), ints_to_array([d0], 2)
    orc.walk(xs, ys, ds, 64, jd, jx, jy, 0, dp_cap=0)
    assert array_to_ints(xs)[0] == 0x823D31F7DBC87485CDE8CFE8536F2B7A0CCE6B57A4EA6AB0D18F883CB1F3C95D
    assert array_to_ints(ys)[0] == 0x8CC4B15A435FAD0B81DB55D870B77CD18B2939B61E7AE5A1FE30B6F4CBB970D9
    assert array_to_ints(ds)[0] == 0x1238E96E2F4FE0A
    assert orc.pubkey(0x1238E96E2F4FE0A)[1:] == (array_to_ints(xs)[0], array_to_ints(ys)[0])


@pytest.mark.parametrize("name", ["walk_check64", "walk_80", "walk_125"])
def test_walks_match_reference(golden, orc, name):
    """The -check scenario (Check.cpp:472-586) replayed by both oracle walks."""
    w = walk_fixture(golden[name])
    jd, jx, jy, _ = orc.jump_table(w["range_power"])
    n = len(w["start"])
    true_d = [d for _, _, d in w["start"]]

    # herd creation semantics (Kangaroo.cpp:670-738)
    hx, hy = orc.create_herd(ints_to_array(true_d), 0, *w["key_to_search"])
    assert array_to_ints(hx) == [x for x, _, _ in w["start"]]
    assert array_to_ints(hy) == [y for _, y, _ in w["start"]]

    # host view: AddDirect per point, distances mod n (Check.cpp:534-556)
    x = ints_to_array([s[0] for s in w["start"]])
    y = ints_to_array([s[1] for s in w["start"]])
    d4 = ints_to_array(true_d)
    dps, total = orc.walk_direct(x, y, d4, w["nsteps"], jd, jx, jy, w["dp_mask"])
    assert total == len(w["dps"])
    got = dp_multiset((r["kidx"], array_to_ints([r["x"]])[0], (int(r["d"][1]) << 64) | int(r["d"][0])) for r in dps)
    want = dp_multiset((k, xx, dd & M128) for k, xx, dd in w["dps"])
    assert got == want
    assert list(zip(array_to_ints(x), array_to_ints(y), array_to_ints(d4))) == w["end"]

    # device view: batched inverse, 128-bit distances with the wild offset folded in
    x = ints_to_array([s[0] for s in w["start"]])
    y = ints_to_array([s[1] for s in w["start"]])
    d2 = ints_to_array(device_distances(true_d, w["wild_offset"]), 2)
    dps, total = orc.walk(x, y, d2, w["nsteps"], jd, jx, jy, w["dp_mask"])
    assert total == len(w["dps"])
    got = dp_multiset(
        (r["kidx"], array_to_ints([r["x"]])[0],
         host_distance((int(r["d"][1]) << 64) | int(r["d"][0]), int(r["kidx"]), w["wild_offset"]))
        for r in dps)
    assert got == dp_multiset(w["dps"])
    end_d = [host_distance(d, i, w["wild_offset"]) for i, d in enumerate(array_to_ints(d2))]
    assert list(zip(array_to_ints(x), array_to_ints(y), end_d)) == w["end"]


def test_walk_empty_herd(orc):
    jd, jx, jy, _ = orc.jump_table(64)
    e4 = np.zeros((0, 4), dtype=np.uint64)
    e2 = np.zeros((0, 2), dtype=np.uint64)
    dps, total = orc.walk(e4.copy(), e4.copy(), e2, 64, jd, jx, jy, 0)
    assert total == 0 and len(dps) == 0


def test_dp_mask(orc):
    # Kangaroo.cpp:154-164
    assert orc.dp_mask(0) == 0
    assert orc.dp_mask(8) == 0xFF00000000000000
    assert orc.dp_mask(14) == 0xFFFC000000000000
    assert orc.dp_mask(64) == 0xFFFFFFFFFFFFFFFF
    assert orc.dp_mask(70) == 0xFFFFFFFFFFFFFFFF


# ---------------------------------------------------------------- BASELINE configs[0]: 32-bit CPU plumbing
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG0_START, CFG0_END = 0x80000000, 0xFFFFFFFF
CFG0_PUB = "0209C58240E50E3BA3F833C82655E8725C037A2294E14CF5D73A5DF8D56159DE69"
CFG0_PRIV = 0xB862A62E  # SURVEY 8(d) config 1 (puzzle32.txt itself holds no 32-bit key)


def test_config0_reference_cpu_program_solves_the_32bit_key(tmp_path):
    """The unmodified reference, CPU target, `-t 1` (SolveKeyCPU, Kangaroo.cpp:334-506) on BASELINE configs[0]."""
    import subprocess

    exe = os.path.join(ROOT, "oracle", "_ref", "kangaroo_cpu")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/kangaroo_cpu not built (needs /root/reference at build time)")
    cfg = tmp_path / "in32.txt"
    cfg.write_text("%X\n%X\n%s\n" % (CFG0_START, CFG0_END, CFG0_PUB))
    out = subprocess.run([exe, "-t", "1", str(cfg)], capture_output=True, text=True, timeout=120)
    assert "Priv: 0x%X" % CFG0_PRIV in out.stdout, out.stdout[-1500:]


def test_config0_oracle_walk_finds_the_same_key(orc):
    """The same input through the oracle alone -- herd (CreateHerd), jump table, batched walk (orc_walk), a Python dict
    as the DP table, CollisionCheck's four sign cases -- i.e. the whole SolveKeyCPU data flow restated: same answer."""
    from helpers import ints_to_array

    rp = 31
    x = int(CFG0_PUB[2:], 16)
    y = pow((x * x * x + 7) % P, (P + 1) // 4, P)
    if (y & 1) != (int(CFG0_PUB[:2], 16) & 1):
        y = P - y
    # keyToSearch = K - start*G (Kangaroo.cpp:892-909)
    _, sx, sy = orc.pubkey(CFG0_START)
    kx, ky = (np.zeros(4, np.uint64) for _ in range(2))
    orc.lib.orc_add_direct(kx, ky, ints_to_array([x])[0], ints_to_array([y])[0], ints_to_array([sx])[0], ints_to_array([P - sy])[0])
    ksx, ksy = array_to_ints([kx])[0], array_to_ints([ky])[0]
    n, woff = 256, ((1 << rp) - 1) >> 1
    rng = np.random.default_rng(32)
    true_d = [int(rng.integers(0, 1 << rp)) if i % 2 == 0 else (int(rng.integers(0, 1 << rp)) - woff) % N_ORDER for i in range(n)]
    hx, hy = orc.create_herd(ints_to_array(true_d), 0, ksx, ksy)
    dev = ints_to_array([(d + woff) % N_ORDER if i & 1 else d for i, d in enumerate(true_d)], 2)
    jd, jx, jy, _ = orc.jump_table(rp)
    table, found = {}, None
    for _ in range(400):
        dps, _total = orc.walk(hx, hy, dev, 16, jd, jx, jy, orc.dp_mask(3))
        for r in dps:
            k = int(r["kidx"])
            d = array_to_ints([r["d"]])[0]
            d = (d - woff) % N_ORDER if k & 1 else d
            key = tuple(int(v) for v in r["x"])
            other = table.get(key)
            if other is None:
                table[key] = (k & 1, d)
            elif other[0] != (k & 1):
                td, wd = (d, other[1]) if k & 1 == 0 else (other[1], d)
                for cand in ((td - wd) % N_ORDER, (td + wd) % N_ORDER, (-td - wd) % N_ORDER, (wd - td) % N_ORDER):
                    if orc.pubkey(cand)[1:] == (ksx, ksy):
                        found = cand + CFG0_START
        if found is not None:
            break
    assert found == CFG0_PRIV
