"""Shared helpers for the parity tests (test-side only; may use the oracle)."""
from __future__ import annotations

import numpy as np

from oracle.binding import N_ORDER, P, array_to_ints, from_limbs, ints_to_array, to_limbs  # noqa: F401

M128 = (1 << 128) - 1


def walk_fixture(g):
    """Decode a golden 'walk_*' record -> dict of python ints / arrays."""
    start = [(int(x, 16), int(y, 16), int(d, 16)) for x, y, d in g["start"]]
    end = [(int(x, 16), int(y, 16), int(d, 16)) for x, y, d in g["end"]]
    dps = [(int(i), int(x, 16), int(d, 16)) for i, x, d in g["dps"]]
    return dict(
        start=start, end=end, dps=dps, nsteps=g["nsteps"], dp_mask=int(g["dp_mask"], 16),
        wild_offset=int(g["wild_offset"], 16), range_power=g["range_power"],
        key_to_search=tuple(int(v, 16) for v in g["key_to_search"]),
    )


def device_distances(true_d, wild_offset):
    """GPUEngine.cu:406-411: odd (wild) indices carry d + wildOffset mod n on the device."""
    out = []
    for i, d in enumerate(true_d):
        if i % 2 == 1:
            d = (d + wild_offset) % N_ORDER
        assert d <= M128, "device distance must fit 128 bits"
        out.append(d)
    return out


def host_distance(dev_d, kidx, wild_offset):
    """GPUEngine.cu:477,672: wild distances have the offset removed mod n on the way back."""
    return (dev_d - wild_offset) % N_ORDER if kidx % 2 == 1 else dev_d


def dp_multiset(dps):
    """Canonical sorted list of (kidx, x, d) for exact multiset comparison."""
    return sorted((int(k), int(x), int(d)) for k, x, d in dps)


def random_field_elems(rng: np.random.Generator, n: int) -> np.ndarray:
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    return a


# ---- operands that drive kng_field.h's fe_fold32 through each rarely-taken branch ----
_M32, _M64 = (1 << 32) - 1, (1 << 64) - 1


def fold_rare_flags(a: int, b: int) -> set:
    """Which rare conditions of fe_fold32's single-chain form (kng_field.h) the product a*b raises: the Python
    restatement of its dataflow, used only to prove that the test vectors reach every branch."""
    w = [((a * b) >> (32 * i)) & _M32 for i in range(16)]
    flags = set()
    e, o = [], []
    for j in range(4):
        v = w[8 + 2 * j] * 977 + (w[2 * j] | (w[2 * j + 1] << 32))
        if v >> 64:
            flags.add(f"ce{j}")
        e.append(v & _M64)
        v = w[9 + 2 * j] * 977 + (w[8 + 2 * j] | (w[9 + 2 * j] << 32))
        if v >> 64:
            flags.add(f"co{j}")
        o.append(v & _M64)
    if flags:
        return flags  # the chain below is only meaningful without dropped carries
    s = sum(e[j] << (64 * j) for j in range(4)) + (sum(o[j] << (64 * j) for j in range(4)) << 32)
    sl = [(s >> (32 * i)) & _M32 for i in range(9)]
    if s >> 288:
        flags.add("top")
    r01 = sl[8] * 977 + (sl[0] | (sl[1] << 32))
    if r01 >> 64:
        flags.add("c1")
    r1 = ((r01 >> 32) & _M32) + sl[8]
    if (sl[2] + (r1 >> 32)) >> 32:
        flags.add("c3")
    return flags


def fold_rare_vectors(rng: np.random.Generator):
    """(a, b) pairs reaching every rare branch of the fold, found by a seeded structured search."""
    full = (1 << 256) - 1
    want = {f"ce{j}" for j in range(4)} | {f"co{j}" for j in range(4)} | {"top", "c1", "c3"}
    found, out = set(), []

    def rnd():
        return int.from_bytes(rng.bytes(32), "little")

    def consider(a, b):
        f = fold_rare_flags(a, b)
        if f - found:
            found.update(f)
            out.append((a, b))

    # a * (2^256 - 1) = a*2^256 - a : hi = a - 1, lo = 2^256 - a  -> shape the limbs of a
    for j in range(4):
        for _ in range(64):
            a = rnd()
            a &= ~(_M64 << (64 * j))
            a |= int(rng.integers(1, 1 << 9)) << (64 * j)  # limb pair (small, 0): lo pair close to 2^64
            consider(a | 1, full)
            a2 = rnd() | (_M32 << (32 * (2 * j + 1)))  # odd limb all ones: (hi << 32) pair overflows
            consider(a2 | 1, full)
    # the rest only depends on S = lo + hi*K; with b = 2^256 - 1: S = 2^256 - K + a*(K - 1), K - 1 = 16 * (2^28 + 61),
    # so the low 96 bits of S can be chosen freely (up to that factor 16) through the low bits of a
    K = 0x1000003D1
    odd_inv = pow((K - 1) >> 4, -1, 1 << 92)

    def a_for_low_bits_of_s(target96, top_limb):
        assert (target96 + K) % 16 == 0
        low = (((target96 + K) % (1 << 96)) >> 4) * odd_inv % (1 << 92)
        mid = rnd() & (((1 << 224) - 1) ^ ((1 << 96) - 1))
        return low | mid | (top_limb << 224)

    # c1: the second fold's MAD s8*977 + (s0, s1) overflows: (s0, s1) = 2^64 - 17, T = s8 ~ 1000
    consider(a_for_low_bits_of_s((rnd() & (_M32 << 64)) | ((1 << 64) - 17), 1000), full)
    # c3: no MAD overflow, r1 + s8 carries, limb 2 all ones: s = (15, 2^32 - 2, 2^32 - 1)
    consider(a_for_low_bits_of_s((_M32 << 64) | ((_M32 - 1) << 32) | 15, 1000), full)
    # top: T needs 33 bits: hi_7 = 2^32 - 977 and hi_6 in [953552, 954529) (see the derivation in fe_fold32)
    for h6 in (953552, 954000, 954528):
        consider((((_M32 - 976) << 224) | (h6 << 192) | (rnd() & ((1 << 192) - 1))) + 1, full)
    return out, found, want


def ref_binary(name: str) -> str:
    """Path of oracle/_ref/<name> (the reference program built from /root/reference by oracle/Makefile; the binaries travel
    to the GPU box with the tree).  Missing: a box WITH a device fails -- there the reference-program rows (a14, g) must be
    tested, not skipped -- a box without one skips.  KNG_REQUIRE_REF=1/0 overrides the detection."""
    import os

    import pytest

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", name)
    if os.path.exists(exe):
        return exe
    need = os.environ.get("KNG_REQUIRE_REF")
    if need is None:
        try:
            import kangaroo_amd

            need = "1" if kangaroo_amd.device_count() > 0 else "0"
        except Exception:
            need = "0"
    msg = f"oracle/_ref/{name} not built (python -c 'import __graft_entry__ as g; g.build()' where /root/reference exists)"
    if need == "1":
        pytest.fail(msg + ": on a box with a device the reference-program tests must run, not skip")
    pytest.skip(msg)
