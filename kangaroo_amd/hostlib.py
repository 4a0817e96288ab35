"""ctypes binding of libkangaroo_host.so (kangaroo_amd/host/kng_host.h): reference-compatible
jump table, herd builder, DP mask / auto-DP and distance bookkeeping.  Product-side host code
(C++), used by bench.py and the tools; the test-suite checks it against the oracle."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_PKG, "lib", "libkangaroo_host.so")
_U64P = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
_M64 = (1 << 64) - 1
_lib = None


def limbs(v: int, n: int = 4) -> np.ndarray:
    return np.array([(v >> (64 * i)) & _M64 for i in range(n)], dtype=np.uint64)


def to_int(a) -> int:
    return sum(int(x) << (64 * i) for i, x in enumerate(a))


def load() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError(f"{_PATH} is missing: run `python -m kangaroo_amd.build` first")
        # the host library links the engine library (GPUEngine class); load that first
        from .engine import load_library

        load_library()
        L = C.CDLL(_PATH)
        L.kngh_pubkey.argtypes = [_U64P, _U64P, _U64P]
        L.kngh_point_add.argtypes = [_U64P] * 6
        L.kngh_on_curve.argtypes = [_U64P, _U64P]
        L.kngh_add_order.argtypes = [_U64P, _U64P, _U64P]
        L.kngh_add_order.restype = None
        L.kngh_sub_order.argtypes = [_U64P, _U64P, _U64P]
        L.kngh_sub_order.restype = None
        L.kngh_dp_mask.argtypes = [C.c_int]
        L.kngh_dp_mask.restype = C.c_uint64
        L.kngh_jump_table.argtypes = [C.c_int, _U64P, _U64P, _U64P]
        L.kngh_jump_table.restype = C.c_double
        L.kngh_suggest_dp.argtypes = [C.c_int, C.c_double]
        L.kngh_create_herd.argtypes = [C.c_uint64, C.c_int, _U64P, C.c_void_p, C.c_void_p, C.c_int, C.c_uint64, C.c_int,
                                       _U64P, _U64P, _U64P]
        L.kngh_herd_params.argtypes = [C.c_int, _U64P, C.c_void_p, C.c_void_p, C.c_uint64, _U64P, _U64P, _U64P, _U64P]
        L.kngh_to_device_distances.argtypes = [_U64P, C.c_uint64, _U64P, _U64P]
        L.kngh_to_true_distances.argtypes = [_U64P, C.c_void_p, C.c_uint64, _U64P, _U64P]
        L.kngh_to_true_distances.restype = None
        _lib = L
    return _lib


def pubkey(k: int):
    x = np.zeros(4, np.uint64)
    y = np.zeros(4, np.uint64)
    rc = load().kngh_pubkey(limbs(k), x, y)
    return rc, to_int(x), to_int(y)


def point_add(p1, p2):
    x = np.zeros(4, np.uint64)
    y = np.zeros(4, np.uint64)
    rc = load().kngh_point_add(limbs(p1[0]), limbs(p1[1]), limbs(p2[0]), limbs(p2[1]), x, y)
    return rc, to_int(x), to_int(y)


def on_curve(x: int, y: int) -> bool:
    return bool(load().kngh_on_curve(limbs(x), limbs(y)))


def dp_mask(dp: int) -> int:
    return int(load().kngh_dp_mask(dp))


def suggest_dp(range_power: int, total_kangaroos: float) -> int:
    return int(load().kngh_suggest_dp(range_power, float(total_kangaroos)))


def jump_table(range_power: int):
    jd = np.zeros((32, 2), np.uint64)
    jx = np.zeros((32, 4), np.uint64)
    jy = np.zeros((32, 4), np.uint64)
    avg = load().kngh_jump_table(range_power, jd, jx, jy)
    return jd, jx, jy, avg


def create_herd(n: int, range_power: int, key_xy=None, first_type: int = 0, seed: int = 1, nthreads: int = 0):
    """Returns x (n,4), y (n,4), d_true (n,4), wild_offset (int)."""
    wild_offset = ((1 << range_power) - 1) >> 1  # Kangaroo.cpp:877-890 rangeWidthDiv2
    x = np.zeros((n, 4), np.uint64)
    y = np.zeros((n, 4), np.uint64)
    d = np.zeros((n, 4), np.uint64)
    if key_xy is None:
        kx = ky = None
    else:
        kxa, kya = limbs(key_xy[0]), limbs(key_xy[1])
        kx, ky = kxa.ctypes.data, kya.ctypes.data
    rc = load().kngh_create_herd(n, range_power, limbs(wild_offset), kx, ky, first_type, seed & _M64, nthreads, x, y, d)
    if rc != 0:
        raise RuntimeError("kngh_create_herd failed")
    return x, y, d, wild_offset


def herd_params(range_power: int, key_xy=None, seed: int = 1):
    """Inputs of GPUEngine.CreateHerdOnDevice / kng_build_herd: (table, windows, base_tame, base_wild, final_add, wild_offset)."""
    wild_offset = ((1 << range_power) - 1) >> 1
    windows = (range_power + 7) // 8
    table = np.zeros(windows * 256 * 8, np.uint64)
    bt, bw, fin = np.zeros(8, np.uint64), np.zeros(8, np.uint64), np.zeros(8, np.uint64)
    if key_xy is None:
        kx = ky = None
    else:
        kxa, kya = limbs(key_xy[0]), limbs(key_xy[1])
        kx, ky = kxa.ctypes.data, kya.ctypes.data
    if load().kngh_herd_params(range_power, limbs(wild_offset), kx, ky, seed & _M64, table, bt, bw, fin) != 0:
        raise RuntimeError("kngh_herd_params failed")
    return table, windows, bt, bw, fin, wild_offset


def audit_params(key_xy, wild_offset: int, seed: int = 0xA0D17):
    """Inputs of kng_audit_setup: the full 16-window table and the offset points for a key whose wild herd was built
    with `wild_offset` (kngh_herd_params at 128 bits, but with the caller's offset instead of 2^127)."""
    table = np.zeros(16 * 256 * 8, np.uint64)
    bt, bw, fin = np.zeros(8, np.uint64), np.zeros(8, np.uint64), np.zeros(8, np.uint64)
    if key_xy is None:
        kx = ky = None
    else:
        kxa, kya = limbs(key_xy[0]), limbs(key_xy[1])
        kx, ky = kxa.ctypes.data, kya.ctypes.data
    if load().kngh_herd_params(128, limbs(wild_offset), kx, ky, seed & _M64, table, bt, bw, fin) != 0:
        raise RuntimeError("kngh_herd_params failed")
    return table, 16, bt, bw, fin


def to_device_distances(d_true: np.ndarray, wild_offset: int) -> np.ndarray:
    out = np.zeros((d_true.shape[0], 2), np.uint64)
    if load().kngh_to_device_distances(np.ascontiguousarray(d_true), d_true.shape[0], limbs(wild_offset), out) != 0:
        raise RuntimeError("a device distance does not fit 128 bits")
    return out


def to_true_distances(d_dev: np.ndarray, wild_offset: int, kidx: np.ndarray | None = None) -> np.ndarray:
    out = np.zeros((d_dev.shape[0], 4), np.uint64)
    kp = None if kidx is None else np.ascontiguousarray(kidx, dtype=np.uint64).ctypes.data
    load().kngh_to_true_distances(np.ascontiguousarray(d_dev), kp, d_dev.shape[0], limbs(wild_offset), out)
    return out
