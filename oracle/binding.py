"""ctypes binding of oracle/liboracle.so (the C restatement of the reference jump path).

TEST INFRASTRUCTURE ONLY -- see oracle/kng_oracle.h.  Arrays are numpy uint64, little-endian
limbs, shape (n,4) for field elements / 256-bit scalars and (n,2) for 128-bit device distances.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_U64P = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")

DP_DTYPE = np.dtype([("x", np.uint64, 4), ("d", np.uint64, 2), ("kidx", np.uint64)])

P = 2**256 - 0x1000003D1
N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141


def to_limbs(v: int, n: int = 4) -> np.ndarray:
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)], dtype=np.uint64)


def from_limbs(a) -> int:
    return sum(int(x) << (64 * i) for i, x in enumerate(a))


def ints_to_array(vals, n: int = 4) -> np.ndarray:
    out = np.zeros((len(vals), n), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = to_limbs(v, n)
    return out


def array_to_ints(a) -> list:
    return [from_limbs(r) for r in a]


class Oracle:
    def __init__(self, path: str):
        self.lib = L = C.CDLL(path)
        for name in ("orc_modmul", "orc_modsub", "orc_add_order", "orc_sub_order"):
            getattr(L, name).argtypes = [_U64P, _U64P, _U64P]
            getattr(L, name).restype = None
        for name in ("orc_modsqr", "orc_modinv"):
            getattr(L, name).argtypes = [_U64P, _U64P]
            getattr(L, name).restype = None
        L.orc_batch_inv.argtypes = [_U64P, C.c_size_t]
        L.orc_batch_inv.restype = None
        L.orc_add_direct.argtypes = [_U64P] * 6
        L.orc_add_direct.restype = None
        L.orc_pubkey.argtypes = [_U64P, _U64P, _U64P]
        L.orc_pubkey.restype = C.c_int
        L.orc_pubkey_add.argtypes = [_U64P] * 5
        L.orc_pubkey_add.restype = C.c_int
        L.orc_rseed.argtypes = [C.c_uint32]
        L.orc_rseed.restype = None
        L.orc_rndl.argtypes = []
        L.orc_rndl.restype = C.c_uint32
        L.orc_int_rand.argtypes = [_U64P, C.c_int]
        L.orc_int_rand.restype = None
        L.orc_dp_mask.argtypes = [C.c_int]
        L.orc_dp_mask.restype = C.c_uint64
        L.orc_jump_table.argtypes = [C.c_int, _U64P, _U64P, _U64P, C.POINTER(C.c_double)]
        L.orc_jump_table.restype = None
        walk_args = [_U64P, _U64P, _U64P, C.c_size_t, C.c_int, _U64P, _U64P, _U64P, C.c_uint64,
                     C.c_void_p, C.c_size_t]
        L.orc_walk.argtypes = walk_args
        L.orc_walk.restype = C.c_size_t
        # the same entry point with raw addresses, for walks over slices of one big array (walk_parallel)
        L.orc_walk_ptr = C.CFUNCTYPE(C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, _U64P, _U64P, _U64P,
                                     C.c_uint64, C.c_void_p, C.c_size_t)(("orc_walk", L))
        L.orc_walk_direct.argtypes = walk_args
        L.orc_walk_direct.restype = C.c_size_t
        L.orc_create_herd.argtypes = [_U64P, _U64P, _U64P, C.c_size_t, C.c_int, _U64P, _U64P]
        L.orc_create_herd.restype = None

    # ---- scalar helpers on python ints -------------------------------------------------
    def _bin(self, fn, a: int, b: int) -> int:
        r = np.zeros(4, dtype=np.uint64)
        fn(r, to_limbs(a), to_limbs(b))
        return from_limbs(r)

    def modmul(self, a, b): return self._bin(self.lib.orc_modmul, a, b)
    def modsub(self, a, b): return self._bin(self.lib.orc_modsub, a, b)
    def add_order(self, a, b): return self._bin(self.lib.orc_add_order, a, b)
    def sub_order(self, a, b): return self._bin(self.lib.orc_sub_order, a, b)

    def modsqr(self, a):
        r = np.zeros(4, dtype=np.uint64)
        self.lib.orc_modsqr(r, to_limbs(a))
        return from_limbs(r)

    def modinv(self, a):
        r = np.zeros(4, dtype=np.uint64)
        self.lib.orc_modinv(r, to_limbs(a))
        return from_limbs(r)

    def batch_inv(self, vals):
        a = ints_to_array(vals)
        self.lib.orc_batch_inv(a, len(vals))
        return array_to_ints(a)

    def pubkey(self, k: int):
        x = np.zeros(4, dtype=np.uint64)
        y = np.zeros(4, dtype=np.uint64)
        rc = self.lib.orc_pubkey(x, y, to_limbs(k))
        return rc, from_limbs(x), from_limbs(y)

    def rseed(self, s): self.lib.orc_rseed(s & 0xFFFFFFFF)
    def rndl(self): return self.lib.orc_rndl()

    def int_rand(self, nbit):
        r = np.zeros(4, dtype=np.uint64)
        self.lib.orc_int_rand(r, nbit)
        return from_limbs(r)

    def dp_mask(self, dp): return int(self.lib.orc_dp_mask(dp))

    def jump_table(self, range_power: int):
        jd = np.zeros((32, 2), dtype=np.uint64)
        jx = np.zeros((32, 4), dtype=np.uint64)
        jy = np.zeros((32, 4), dtype=np.uint64)
        avg = C.c_double(0)
        self.lib.orc_jump_table(range_power, jd, jx, jy, C.byref(avg))
        return jd, jx, jy, avg.value

    # ---- walks on arrays ------------------------------------------------------------------
    def walk(self, x, y, d, nsteps, jd, jx, jy, dpmask, dp_cap=1 << 20):
        """Device-view walk (128-bit raw distances).  Updates x,y,d in place; returns (dps, total)."""
        n = x.shape[0]
        assert x.shape == (n, 4) and y.shape == (n, 4) and d.shape == (n, 2)
        dps = np.zeros(dp_cap, dtype=DP_DTYPE)
        total = self.lib.orc_walk(x, y, d, n, nsteps, jd, jx, jy, dpmask, dps.ctypes.data, dp_cap)
        return dps[: min(total, dp_cap)], total

    def walk_parallel(self, x, y, d, nsteps, jd, jx, jy, dpmask, threads=None, chunk=1 << 14):
        """orc_walk over a thread pool: kangaroos are independent, so contiguous chunks walked separately give the
        same states and the same distinguished points (ctypes releases the GIL; the C code keeps no global state in
        the walk).  Updates x, y, d in place; returns every DP as one DP_DTYPE array with herd-wide kidx."""
        from concurrent.futures import ThreadPoolExecutor

        n = x.shape[0]
        assert x.shape == (n, 4) and y.shape == (n, 4) and d.shape == (n, 2)
        assert x.flags.c_contiguous and y.flags.c_contiguous and d.flags.c_contiguous
        threads = threads or min(os.cpu_count() or 1, 128)
        addr = lambda a, i: C.c_void_p(a.ctypes.data + i * a.strides[0])  # noqa: E731
        fn = self.lib.orc_walk_ptr

        def one(c0):
            m = min(chunk, n - c0)
            # 4x the expected number of points, and room for a burst in a small chunk
            cap = 64 + 4 * int(m * nsteps * (1.0 if dpmask == 0 else 2.0 ** -bin(dpmask).count("1")))
            dps = np.zeros(cap, dtype=DP_DTYPE)
            total = fn(addr(x, c0), addr(y, c0), addr(d, c0), m, nsteps, jd, jx, jy, dpmask, dps.ctypes.data, cap)
            assert total <= cap, "DP buffer of a chunk overflowed"
            dps = dps[:total]
            dps["kidx"] += np.uint64(c0)
            return dps

        with ThreadPoolExecutor(threads) as ex:
            parts = list(ex.map(one, range(0, n, chunk)))
        return np.concatenate(parts) if parts else np.zeros(0, dtype=DP_DTYPE)

    def walk_direct(self, x, y, d4, nsteps, jd, jx, jy, dpmask, dp_cap=1 << 20):
        """Host-view walk of Check.cpp (AddDirect, 256-bit distances mod n)."""
        n = x.shape[0]
        assert x.shape == (n, 4) and y.shape == (n, 4) and d4.shape == (n, 4)
        dps = np.zeros(dp_cap, dtype=DP_DTYPE)
        total = self.lib.orc_walk_direct(x, y, d4, n, nsteps, jd, jx, jy, dpmask, dps.ctypes.data, dp_cap)
        return dps[: min(total, dp_cap)], total

    def create_herd(self, d4, first_type, kx: int, ky: int):
        n = d4.shape[0]
        x = np.zeros((n, 4), dtype=np.uint64)
        y = np.zeros((n, 4), dtype=np.uint64)
        self.lib.orc_create_herd(x, y, np.ascontiguousarray(d4), n, first_type, to_limbs(kx), to_limbs(ky))
        return x, y


def _create_herd_parallel(self, d4, kx: int, ky: int, threads=None, chunk=1 << 12):
    """create_herd (tame d*G at even, wild K + d*G at odd indices) over a thread pool: points are independent"""
    from concurrent.futures import ThreadPoolExecutor

    n = d4.shape[0]
    d4 = np.ascontiguousarray(d4)
    x = np.zeros((n, 4), dtype=np.uint64)
    y = np.zeros((n, 4), dtype=np.uint64)
    fn = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, _U64P, _U64P)(("orc_create_herd", self.lib))
    kxl, kyl = to_limbs(kx), to_limbs(ky)
    addr = lambda a, i: C.c_void_p(a.ctypes.data + i * a.strides[0])  # noqa: E731

    def one(c0):
        m = min(chunk, n - c0)
        fn(addr(x, c0), addr(y, c0), addr(d4, c0), m, c0 & 1, kxl, kyl)

    with ThreadPoolExecutor(threads or min(os.cpu_count() or 1, 256)) as ex:
        list(ex.map(one, range(0, n, chunk)))
    return x, y


Oracle.create_herd_parallel = _create_herd_parallel


def build_oracle() -> str:
    """Compile oracle/liboracle.so (gcc, <1 s).  Building the checker is not using it."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    return os.path.join(_HERE, "liboracle.so")


_cached = None


def load_oracle() -> Oracle:
    global _cached
    if _cached is None:
        path = os.path.join(_HERE, "liboracle.so")
        src = os.path.join(_HERE, "kng_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build_oracle()
        _cached = Oracle(path)
    return _cached
