"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/kangaroo_hip.h declares, and fails loudly (no fallback) when no GPU is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    with open(os.path.join(ROOT, "include", "kangaroo_hip.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(kng_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import ctypes

    from kangaroo_amd.build import build_engine

    lib = ctypes.CDLL(build_engine())
    names = _declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/kangaroo_hip.h but not exported"


def test_host_library_exports_every_declared_symbol_and_headers_are_plain_c():
    """libkangaroo_host.so: every kngh_/kngt_/kngw_/kngs_ function of kangaroo_amd/host/*.h is exported and the
    headers compile as C99 (they are the binding surface for the host pipeline)."""
    import ctypes
    import subprocess
    import tempfile

    from kangaroo_amd import hostlib

    lib = hostlib.load()
    assert isinstance(lib, ctypes.CDLL)
    host = os.path.join(ROOT, "kangaroo_amd", "host")
    total = 0
    for hdr, prefix in (("kng_host.h", "kngh_"), ("kng_dptable.h", "kngt_"), ("kng_workfile.h", "kngw_"), ("kng_solver.h", "kngs_")):
        with open(os.path.join(host, hdr)) as f:
            src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
        names = sorted(set(re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, src)))
        assert names, hdr
        total += len(names)
        for name in names:
            assert hasattr(lib, name), f"{name} declared in {hdr} but not exported"
    assert total >= 40
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        with open(src, "w") as f:
            f.write('#include "kng_host.h"\n#include "kng_dptable.h"\n#include "kng_workfile.h"\n#include "kng_solver.h"\n'
                    "int main(void){return sizeof(kngt_entry)==32 ? 0 : 1;}\n")
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", host, src, "-o", exe])
        assert subprocess.call([exe]) == 0


def test_header_is_plain_c():
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "t.c")
        with open(src, "w") as f:
            f.write('#include "kangaroo_hip.h"\nint main(void){kng_item it; (void)it; return sizeof(kng_item)==56?0:1;}\n')
        exe = os.path.join(td, "t")
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe])
        assert subprocess.call([exe]) == 0


def test_no_cpu_fallback_without_device():
    import kangaroo_amd

    if kangaroo_amd.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(kangaroo_amd.EngineError, match="no HIP device"):
        kangaroo_amd.GPUEngine(1, 1)
    import numpy as np

    with pytest.raises(kangaroo_amd.EngineError):
        kangaroo_amd.test_fieldop("modmul", np.ones((1, 4), np.uint64))


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under kangaroo_amd/ may import, include, link
    or call it (comments that merely mention it are fine)."""
    pat = re.compile(r"(import\s+oracle|from\s+oracle|liboracle|kng_oracle|\borc_[a-z]|-loracle|#include\s*[<\"][^>\"]*oracle)")
    bad = []
    for dp, _, files in os.walk(os.path.join(ROOT, "kangaroo_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp", ".c")) or fn == "Makefile":
                with open(os.path.join(dp, fn), errors="ignore") as f:
                    if pat.search(f.read()):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
