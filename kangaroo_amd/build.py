"""Build the HIP engine (libkangaroo_hip.so) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the built library travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libkangaroo_hip.so")
HOSTLIB = os.path.join(LIBDIR, "libkangaroo_host.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall"]


def _sources(dirname, exts):
    return sorted(os.path.join(dirname, f) for f in os.listdir(dirname) if f.endswith(exts))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def hipcc_path() -> str:
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def build_engine(force: bool = False, verbose: bool = False) -> str:
    """Compile kangaroo_amd/csrc/*.hip into kangaroo_amd/lib/libkangaroo_hip.so."""
    os.makedirs(LIBDIR, exist_ok=True)
    deps = _sources(CSRC, (".hip", ".h")) + [os.path.join(ROOT, "include", "kangaroo_hip.h")]
    if force or _stale(LIB, deps):
        cmd = [hipcc_path(), *HIPCC_FLAGS, "-o", LIB, *_sources(CSRC, (".hip",))]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


def build_all(force: bool = False, verbose: bool = False) -> dict:
    out = {"engine": build_engine(force, verbose)}
    host_dir = os.path.join(PKG, "host")
    if os.path.exists(os.path.join(host_dir, "Makefile")):
        subprocess.check_call(["make", "-s", "-C", host_dir] + (["-B"] if force else []))
        out["host"] = host_dir
    return out


if __name__ == "__main__":
    print(build_all(verbose=True))
