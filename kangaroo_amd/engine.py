"""ctypes binding of libkangaroo_hip.so (include/kangaroo_hip.h) + a Python mirror of the
reference's `class GPUEngine` (GPU/GPUEngine.h:40-64) for the test-suite and bench.py.

The mirror keeps the reference's method names and call protocol (SetParams, SetWildOffset,
SetKangaroos, GetKangaroos, SetKangaroo, callKernel, Launch, GetNbThread, GetGroupSize,
GetMemory) so the parity tests read like Check.cpp:467-621.  Distances given to / returned by
this class are TRUE distances mod n; the wild offset is added / removed here exactly as
GPUEngine.cu:406-411,477,672 do it (the C++ class in kangaroo_amd/host does the same for C++
callers).  All compute happens in the HIP library; a missing library or device raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

KNG_NB_JUMP = 32
KNG_NB_RUN = 64
KNG_GRP_SIZE = 128

N_ORDER = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
_M64 = (1 << 64) - 1

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_PKG, "lib", "libkangaroo_hip.so")
_U64P = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")

ITEM_DTYPE = np.dtype([("x", np.uint64, 4), ("d", np.uint64, 2), ("kidx", np.uint64)])
RECORD_DTYPE = np.dtype([("x", np.uint64, 4), ("d", np.uint64, 2), ("kidx", np.uint64), ("reserved", np.uint64)])  # kng_dp_record


class EngineError(RuntimeError):
    pass


_lib = None


def load_library(path: str | None = None) -> C.CDLL:
    """Load the HIP engine.  Fails loudly if it has not been built (no fallback of any kind)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or os.environ.get("KNG_LIB_PATH") or _LIB_PATH  # KNG_LIB_PATH: A/B runs of two builds of the engine
    if not os.path.exists(path):
        raise EngineError(f"{path} is missing: run `python -m kangaroo_amd.build` (hipcc, gfx950) first")
    L = C.CDLL(path)
    L.kng_last_error.restype = C.c_char_p
    L.kng_version.restype = C.c_char_p
    L.kng_device_count.restype = C.c_int
    L.kng_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_uint64),
                                  C.c_char_p, C.c_size_t]
    L.kng_default_grid.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.kng_create.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_void_p)]
    L.kng_destroy.argtypes = [C.c_void_p]
    L.kng_destroy.restype = None
    L.kng_nb_kangaroos.argtypes = [C.c_void_p]
    L.kng_nb_kangaroos.restype = C.c_uint64
    L.kng_memory_bytes.argtypes = [C.c_void_p]
    L.kng_memory_bytes.restype = C.c_uint64
    L.kng_set_params.argtypes = [C.c_void_p, C.c_uint64, _U64P, _U64P, _U64P]
    L.kng_set_kangaroos.argtypes = [C.c_void_p, _U64P, C.c_size_t, _U64P, C.c_size_t, _U64P, C.c_size_t, C.c_uint64]
    L.kng_get_kangaroos.argtypes = [C.c_void_p, _U64P, C.c_size_t, _U64P, C.c_size_t, _U64P, C.c_size_t, C.c_uint64]
    L.kng_set_kangaroos_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, _U64P, C.c_size_t, _U64P, C.c_size_t, _U64P, C.c_size_t]
    L.kng_get_kangaroos_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, _U64P, C.c_size_t, _U64P, C.c_size_t, _U64P, C.c_size_t]
    L.kng_set_kangaroo.argtypes = [C.c_void_p, C.c_uint64, _U64P, _U64P, _U64P]
    L.kng_build_herd.argtypes = [C.c_void_p, C.c_int, C.c_uint64, _U64P, C.c_uint32, _U64P, _U64P, _U64P]
    L.kng_build_herd.restype = C.c_int
    L.kng_launch.argtypes = [C.c_void_p]
    L.kng_wait.argtypes = [C.c_void_p, C.c_int]
    L.kng_drain.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.kng_last_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.kng_set_option.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.kng_get_option.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]
    L.kng_test_fieldop.argtypes = [C.c_int, C.c_int, _U64P, _U64P, _U64P, C.c_uint64]
    L.kng_audit_setup.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, _U64P]
    L.kng_audit_herd.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint32]
    L.kng_audit_points.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_void_p, C.c_uint32]
    L.kng_drain_view.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.kng_outstanding.argtypes = [C.c_void_p]
    L.kng_snapshot.argtypes = [C.c_void_p, C.c_void_p]
    L.kng_snapshot_read.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.kng_snapshot_write.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p]
    L.kng_snapshot_restore.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint64)]
    L.kng_snapshot_release.argtypes = [C.c_void_p]
    for name in ("kng_device_info", "kng_default_grid", "kng_create", "kng_set_params", "kng_set_kangaroos",
                 "kng_get_kangaroos", "kng_set_kangaroos_range", "kng_get_kangaroos_range", "kng_set_kangaroo", "kng_launch", "kng_wait", "kng_drain",
                 "kng_last_kernel_ms", "kng_set_option", "kng_get_option", "kng_test_fieldop", "kng_audit_setup", "kng_audit_herd",
                 "kng_audit_points", "kng_drain_view", "kng_outstanding", "kng_snapshot", "kng_snapshot_read", "kng_snapshot_write",
                 "kng_snapshot_restore", "kng_snapshot_release"):
        getattr(L, name).restype = C.c_int
    _lib = L
    return L


def _check(rc: int) -> None:
    if rc != 0:
        raise EngineError(f"kangaroo_hip error {rc}: {load_library().kng_last_error().decode()}")


def device_count() -> int:
    return load_library().kng_device_count()


def device_info(dev: int = 0) -> dict:
    L = load_library()
    name = C.create_string_buffer(256)
    arch = C.create_string_buffer(256)
    cu = C.c_int(0)
    mem = C.c_uint64(0)
    _check(L.kng_device_info(dev, name, 256, C.byref(cu), C.byref(mem), arch, 256))
    return {"name": name.value.decode(), "arch": arch.value.decode(), "cu_count": cu.value, "mem_bytes": mem.value}


def device_free_bytes(dev: int = 0) -> tuple:
    """(free, total) device memory in bytes (kng_device_free_bytes)."""
    L = load_library()
    L.kng_device_free_bytes.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    f, t = C.c_uint64(0), C.c_uint64(0)
    _check(L.kng_device_free_bytes(dev, C.byref(f), C.byref(t)))
    return int(f.value), int(t.value)


def default_grid(dev: int = 0, x: int = 0, y: int = 0) -> tuple:
    """GPUEngine::GetGridSize (GPUEngine.cu:280-308)."""
    cx, cy = C.c_int(x), C.c_int(y)
    _check(load_library().kng_default_grid(dev, C.byref(cx), C.byref(cy)))
    return cx.value, cy.value


OPS = {"modmul": 0, "modsqr": 1, "modsub": 2, "modinv": 3}


def test_fieldop(op: str, a: np.ndarray, b: np.ndarray | None = None, dev: int = 0) -> np.ndarray:
    """Run a 256-bit primitive on the GPU over n x 4 limb arrays (device self-test entry point)."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = a if b is None else np.ascontiguousarray(b, dtype=np.uint64)
    r = np.zeros_like(a)
    _check(load_library().kng_test_fieldop(dev, OPS[op], a, b, r, a.shape[0]))
    return r


test_fieldop.__test__ = False  # not a pytest test


def _limbs(v: int, n: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & _M64 for i in range(n)], dtype=np.uint64)


class GPUEngine:
    """Python mirror of the reference's GPUEngine (GPU/GPUEngine.h:40-64) over the C ABI."""

    def __init__(self, nbThreadGroup: int, nbThreadPerGroup: int, gpuId: int = 0, maxFound: int = 65536 * 2,
                 **options):
        self._L = load_library()
        self._h = C.c_void_p()
        _check(self._L.kng_create(gpuId, nbThreadGroup, nbThreadPerGroup, maxFound, C.byref(self._h)))
        self.gpuId = gpuId
        self.maxFound = maxFound
        self.nbThread = nbThreadGroup * nbThreadPerGroup
        self.nbThreadPerGroup = nbThreadPerGroup
        self.wildOffset = 0
        self.lostWarning = False
        self.lastLost = 0
        self._outstanding = False
        self._items = np.zeros(maxFound, dtype=ITEM_DTYPE)
        info = device_info(gpuId)
        # GPUEngine.cu:176-182 banner; "(CUs x lanes)" instead of the CUDA-core table
        self.deviceName = (f"GPU #{gpuId} {info['name']} ({info['cu_count']}x64 lanes) "
                           f"Grid({nbThreadGroup}x{nbThreadPerGroup})")
        for k, v in options.items():
            self.set_option(k, v)

    # -- lifetime --------------------------------------------------------------------------
    def close(self):
        if self._h:
            self._L.kng_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- reference surface -------------------------------------------------------------------
    def GetNbThread(self) -> int:
        return self.nbThread

    def GetGroupSize(self) -> int:
        return KNG_GRP_SIZE

    def GetMemory(self) -> int:
        return int(self._L.kng_memory_bytes(self._h))

    @property
    def nbKangaroo(self) -> int:
        return int(self._L.kng_nb_kangaroos(self._h))

    def SetWildOffset(self, offset: int) -> None:
        self.wildOffset = int(offset)

    def SetParams(self, dpMask: int, jd: np.ndarray, jx: np.ndarray, jy: np.ndarray) -> None:
        """jd: (32,2), jx/jy: (32,4) uint64 limbs (GPUEngine.cu:559-590)."""
        jd = np.ascontiguousarray(jd, dtype=np.uint64)
        jx = np.ascontiguousarray(jx, dtype=np.uint64)
        jy = np.ascontiguousarray(jy, dtype=np.uint64)
        assert jd.shape == (32, 2) and jx.shape == (32, 4) and jy.shape == (32, 4)
        _check(self._L.kng_set_params(self._h, dpMask & _M64, jd, jx, jy))

    def _to_device_d(self, d_true: np.ndarray) -> np.ndarray:
        """(n,4) true distances mod n -> (n,2) device distances (wild += offset mod n)."""
        n = d_true.shape[0]
        out = np.ascontiguousarray(d_true[:, :2]).copy()
        if self.wildOffset == 0 and not d_true[:, 2:].any():
            return out
        for i in range(n):
            v = sum(int(d_true[i, k]) << (64 * k) for k in range(4))
            if i & 1:
                v = (v + self.wildOffset) % N_ORDER
            if v >> 128:
                raise EngineError(f"kangaroo {i}: device distance does not fit 128 bits")
            out[i, 0] = v & _M64
            out[i, 1] = v >> 64
        return out

    def _to_true_d(self, d_dev: np.ndarray, kidx: np.ndarray | None = None) -> np.ndarray:
        n = d_dev.shape[0]
        out = np.zeros((n, 4), dtype=np.uint64)
        out[:, :2] = d_dev
        if self.wildOffset == 0:
            return out
        idx = np.arange(n) if kidx is None else kidx
        for i in np.nonzero(np.asarray(idx) & 1)[0]:
            v = (int(d_dev[i, 0]) | (int(d_dev[i, 1]) << 64))
            v = (v - self.wildOffset) % N_ORDER
            out[i] = _limbs(v, 4)
        return out

    def SetKangaroos(self, px: np.ndarray, py: np.ndarray, d: np.ndarray) -> None:
        """px, py: (n,4); d: (n,4) true distances mod n, or (n,2) device distances."""
        px = np.ascontiguousarray(px, dtype=np.uint64)
        py = np.ascontiguousarray(py, dtype=np.uint64)
        d = np.ascontiguousarray(d, dtype=np.uint64)
        dd = self._to_device_d(d) if d.shape[1] == 4 else d
        _check(self._L.kng_set_kangaroos(self._h, px, 4, py, 4, dd, 2, px.shape[0]))

    def GetKangaroos(self, raw: bool = False):
        n = self.nbKangaroo
        px = np.zeros((n, 4), dtype=np.uint64)
        py = np.zeros((n, 4), dtype=np.uint64)
        dd = np.zeros((n, 2), dtype=np.uint64)
        _check(self._L.kng_get_kangaroos(self._h, px, 4, py, 4, dd, 2, n))
        return px, py, (dd if raw else self._to_true_d(dd))

    def SetKangaroosRange(self, first: int, px: np.ndarray, py: np.ndarray, dd: np.ndarray) -> None:
        """Upload kangaroos first .. first+len-1 (kng_set_kangaroos_range); dd: (m,2) DEVICE distances.  The herd
        counts as loaded once a range ending at the last kangaroo has been set."""
        px = np.ascontiguousarray(px, dtype=np.uint64)
        py = np.ascontiguousarray(py, dtype=np.uint64)
        dd = np.ascontiguousarray(dd, dtype=np.uint64)
        _check(self._L.kng_set_kangaroos_range(self._h, first, px.shape[0], px, 4, py, 4, dd, 2))

    def GetKangaroosRange(self, first: int, count: int):
        """(x, y, device distances) of kangaroos first .. first+count-1 (kng_get_kangaroos_range)."""
        px = np.zeros((count, 4), dtype=np.uint64)
        py = np.zeros((count, 4), dtype=np.uint64)
        dd = np.zeros((count, 2), dtype=np.uint64)
        _check(self._L.kng_get_kangaroos_range(self._h, first, count, px, 4, py, 4, dd, 2))
        return px, py, dd

    # -- work-file snapshot (kng_snapshot*): the kangaroo section of a work file, 96-byte records ---------------------
    def _woff_limbs(self, with_offset: bool):
        return _limbs(self.wildOffset % N_ORDER, 4) if (with_offset and self.wildOffset) else None

    def Snapshot(self, with_offset: bool = True) -> None:
        """Freeze the herd as work-file records {x, y, true distance mod n} in a second device buffer; returns at once
        (stream-ordered between two launches)."""
        w = self._woff_limbs(with_offset)
        _check(self._L.kng_snapshot(self._h, None if w is None else w.ctypes.data))

    def SnapshotRead(self, first: int = 0, count: int | None = None) -> np.ndarray:
        """records first .. first+count-1 of the last snapshot as a (count, 12) uint64 array: x[4] y[4] d[4]"""
        count = self.nbKangaroo - first if count is None else count
        out = np.zeros((count, 12), dtype=np.uint64)
        _check(self._L.kng_snapshot_read(self._h, first, count, out.ctypes.data))
        return out

    def SnapshotWrite(self, first: int, records: np.ndarray) -> None:
        records = np.ascontiguousarray(records, dtype=np.uint64)
        assert records.ndim == 2 and records.shape[1] == 12
        _check(self._L.kng_snapshot_write(self._h, first, records.shape[0], records.ctypes.data))

    def SnapshotRestore(self, first: int = 0, count: int | None = None, with_offset: bool = True) -> None:
        """records first .. first+count-1 (uploaded with SnapshotWrite) become herd state; wild distances get the offset back"""
        count = self.nbKangaroo - first if count is None else count
        w = self._woff_limbs(with_offset)
        bad = C.c_uint64(0)
        _check(self._L.kng_snapshot_restore(self._h, first, count, None if w is None else w.ctypes.data, C.byref(bad)))

    def SnapshotRelease(self) -> None:
        _check(self._L.kng_snapshot_release(self._h))

    def CreateHerdOnDevice(self, range_power: int, key_xy=None, seed: int = 1) -> int:
        """Build the whole herd on the GPU (kng_build_herd; replaces Kangaroo::CreateHerd + SetKangaroos).
        Sets and returns the wild offset (N/2).  key_xy: the (shifted) public key for the wild herd."""
        from . import hostlib

        table, windows, bt, bw, fin, woff = hostlib.herd_params(range_power, key_xy, seed)
        _check(self._L.kng_build_herd(self._h, range_power, seed & _M64, table, windows, bt, bw, fin))
        self.wildOffset = woff
        return woff

    def SetKangaroo(self, kIdx: int, px: int, py: int, d: int) -> None:
        if kIdx & 1:
            d = (d + self.wildOffset) % N_ORDER
        if d >> 128:
            raise EngineError("device distance does not fit 128 bits")
        _check(self._L.kng_set_kangaroo(self._h, kIdx, _limbs(px, 4), _limbs(py, 4), _limbs(d, 2)))

    def callKernel(self) -> bool:
        _check(self._L.kng_launch(self._h))
        self._outstanding = True
        return True

    def wait(self, spin: bool = False) -> None:
        _check(self._L.kng_wait(self._h, 1 if spin else 0))
        self._outstanding = False

    def drain(self, raw: bool = False):
        """Items of the most recently waited launch: structured array (x, d, kidx)."""
        n = C.c_uint32(0)
        lost = C.c_uint32(0)
        _check(self._L.kng_drain(self._h, self._items.ctypes.data, self.maxFound, C.byref(n), C.byref(lost)))
        if lost.value and not self.lostWarning:
            print(f"\nWarning, {lost.value} items lost\nHint: Search with less threads (-g) or increse dp (-d)")
            self.lostWarning = True
        items = self._items[: n.value].copy()
        self.lastLost = lost.value
        if not raw and self.wildOffset and n.value:
            true_d = self._to_true_d(items["d"], items["kidx"])
            out = np.zeros(n.value, dtype=np.dtype([("x", np.uint64, 4), ("d", np.uint64, 4), ("kidx", np.uint64)]))
            out["x"], out["d"], out["kidx"] = items["x"], true_d, items["kidx"]
            return out
        return items

    def Launch(self, spinWait: bool = False, raw: bool = False):
        """GPUEngine::Launch (GPUEngine.cu:607-679): DPs of the PREVIOUS kernel, then start the next.
        Unlike the reference the next kernel is started BEFORE the DP copy, so the copy overlaps it."""
        had_launch = self._outstanding  # False on the very first call (Check.cpp:526)
        if had_launch:
            self.wait(spinWait)
        self.callKernel()
        return self.drain(raw=raw) if had_launch else self._items[:0].copy()

    def callKernelAndWait(self) -> bool:
        self.callKernel()
        self.wait()
        return True

    # -- measurement / tuning -----------------------------------------------------------------
    def last_kernel_ms(self) -> float:
        ms = C.c_float(0)
        _check(self._L.kng_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def set_option(self, key: str, value: int) -> None:
        _check(self._L.kng_set_option(self._h, key.encode(), int(value)))

    def get_option(self, key: str) -> int:
        v = C.c_int64(0)
        _check(self._L.kng_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    # -- whole-run audit (kng_audit_*) ------------------------------------------------------------
    def audit_setup(self, key_xy=None, seed: int = 0xA0D17, wild_offset: int | None = None) -> None:
        """Upload the inputs of the device audit for this key: the 16-window table of G and the offset points
        (hostlib.herd_params at 128 bits).  key_xy = the (shifted) public key the wild herd walks from; the wild offset
        defaults to the one SetWildOffset / CreateHerdOnDevice left."""
        from . import hostlib

        woff = self.wildOffset if wild_offset is None else wild_offset
        table, windows, bt, bw, fin = hostlib.audit_params(key_xy, woff, seed)
        assert windows == 16
        _check(self._L.kng_audit_setup(self._h, table, bt, bw, fin))

    def audit_herd(self, cap: int = 16):
        """(mismatches, first mismatching kIdx ...): every kangaroo re-derived from its device distance, x and y compared."""
        bad = C.c_uint64(0)
        idx = np.zeros(max(cap, 1), np.uint64)
        _check(self._L.kng_audit_herd(self._h, C.byref(bad), idx.ctypes.data, cap))
        # the device keeps at most 1024 indices per audit launch (2^21 kangaroos each): never hand back the zero padding
        kept = 1024 * max(1, -(-self.nbKangaroo // (1 << 21)))
        return int(bad.value), [int(v) for v in idx[: min(cap, bad.value, kept)]]

    def audit_points(self, records: np.ndarray, cap: int = 16):
        """(mismatches, first mismatching positions): records of RECORD_DTYPE (reserved = 0: full x, 1: table-entry bits)."""
        assert records.dtype == RECORD_DTYPE and records.flags.c_contiguous
        bad = C.c_uint64(0)
        idx = np.zeros(max(cap, 1), np.uint64)
        _check(self._L.kng_audit_points(self._h, records.ctypes.data, len(records), C.byref(bad), idx.ctypes.data, cap))
        kept = 1024 * max(1, -(-len(records) // (1 << 21)))
        return int(bad.value), [int(v) for v in idx[: min(cap, bad.value, kept)]]

    def drain_records(self) -> np.ndarray:
        """kng_drain_view: the 64-byte records of the most recently waited launch, copied out of the landing buffer."""
        p = C.c_void_p()
        n = C.c_uint32(0)
        lost = C.c_uint32(0)
        _check(self._L.kng_drain_view(self._h, C.byref(p), C.byref(n), C.byref(lost)))
        self.lastLost = lost.value
        if not n.value:
            return np.zeros(0, RECORD_DTYPE)
        buf = (C.c_char * (n.value * RECORD_DTYPE.itemsize)).from_address(p.value)
        return np.frombuffer(buf, dtype=RECORD_DTYPE).copy()
