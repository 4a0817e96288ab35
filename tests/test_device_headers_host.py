"""The DEVICE field arithmetic (kangaroo_amd/csrc/kng_field.h, kng_modinv.h), compiled for
the HOST with the ROCm clang++ and checked against the reference's golden vectors.

No GPU needed: the headers are written so that everything except the inline-asm fast paths also compiles
as plain C++.  This catches arithmetic regressions in the kernel source before a GPU box is involved; the
asm paths themselves are covered by the `-m gpu` primitive tests.
"""
from __future__ import annotations

import os
import shutil
import subprocess

import pytest

from tests.helpers import P

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = shutil.which("clang++") or "/opt/rocm/lib/llvm/bin/clang++"


@pytest.fixture(scope="module")
def host_field(tmp_path_factory):
    if not os.path.exists(CLANG):
        pytest.skip("clang++ not available")
    exe = tmp_path_factory.mktemp("hostfield") / "host_field_test"
    subprocess.run([CLANG, "-O2", "-std=c++17", "-o", str(exe), os.path.join(ROOT, "tools", "host_field_test.cpp")],
                   check=True)

    def run(lines):
        out = subprocess.run([str(exe)], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True)
        return [int(v, 16) for v in out.stdout.split()]

    return run


def test_modmul_modsqr_bit_exact_with_reference(golden, host_field):
    """IntMod.cpp ModMulK1/ModSquareK1 incl. the lazy (non-canonical) fold: low 256 bits bit-exact."""
    mul = golden["modmul"]
    got = host_field([f"mul {a} {b}" for a, b, _ in mul])
    assert got == [int(r, 16) & ((1 << 256) - 1) for _, _, r in mul]
    sqr = golden["modsqr"]
    got = host_field([f"sqr {a} {a}" for a, _ in sqr])
    assert got == [int(r, 16) & ((1 << 256) - 1) for _, r in sqr]


def test_modsub_bit_exact_with_reference(golden, host_field):
    sub = golden["modsub"]
    got = host_field([f"sub {a} {b}" for a, b, _ in sub])
    assert got == [int(r, 16) & ((1 << 256) - 1) for _, _, r in sub]


def test_modinv_safegcd_and_fermat(golden, host_field):
    inv = [(a, r) for a, r in golden["modinv"] if int(a, 16) % P]
    want = [int(r, 16) % P for _, r in inv]
    assert host_field([f"inv {a} {a}" for a, _ in inv]) == want
    assert host_field([f"invf {a} {a}" for a, _ in inv]) == want
    # the packed 2 x 15-step division steps against Fermat on seeded values of every size
    import random

    rnd = random.Random(0xD1F5)
    vals = [rnd.getrandbits(rnd.choice((8, 31, 64, 129, 200, 255, 256))) % P or 1 for _ in range(3000)]
    vals += [1, 2, P - 1, P - 2, (P + 1) // 2, 1 << 255, (1 << 256) - 1 - P]
    assert host_field([f"inv {v:064x} {v:064x}" for v in vals]) == [pow(v, P - 2, P) for v in vals]


def test_fold_rare_branches_bit_exact(host_field, orc):
    """fe_fold32's single-chain form leaves through fe_fold32_full when a MAD overflows, T needs 33 bits or the
    second fold ripples: vectors built to raise each of the 11 conditions, against the oracle's ModMulK1."""
    import numpy as np

    from tests.helpers import array_to_ints, fold_rare_vectors, ints_to_array

    vecs, found, want = fold_rare_vectors(np.random.default_rng(5))
    assert found >= want, want - found
    a, b = ints_to_array([v[0] for v in vecs]), ints_to_array([v[1] for v in vecs])
    exp = np.zeros_like(a)
    for i in range(len(vecs)):
        orc.lib.orc_modmul(exp[i], a[i], b[i])
    assert host_field([f"mul {x:064x} {y:064x}" for x, y in vecs]) == array_to_ints(exp)
