"""ctypes binding of the host pipeline in libkangaroo_host.so (kangaroo_amd/host/):

  DpTable   -- kng_dptable.h: the distinguished-point table, serialisation-compatible with the reference's
               HashTable (HashTable.cpp:75-100,262-307,375-396)
  WorkFile  -- kng_workfile.h: work files in the reference's format (Backup.cpp:368-407,497-552)
  Solver    -- kng_solver.h: multi-GPU SolveKeyGPU replacement (Kangaroo.cpp:510-644) with an asynchronous
               DP drain, sharded table, stream-ordered kangaroo replacement, save/restore

Product-side code; the heavy lifting is C++.  Nothing here falls back to the CPU: Solver.start() raises when
an engine cannot be created.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import hostlib
from .hostlib import limbs, to_int

_U64P = np.ctypeslib.ndpointer(dtype=np.uint64, flags="C_CONTIGUOUS")
ADD_OK, ADD_DUPLICATE, ADD_COLLISION = 0, 1, 2
HEADW, HEADK = 0xFA6A8001, 0xFA6A8002
N_BUCKETS = 1 << 18
MAX_GPUS = 16

ENTRY_DTYPE = np.dtype([("x", np.uint64, (2,)), ("d", np.uint64, (2,))])


class _Header(C.Structure):
    _fields_ = [("magic", C.c_uint32), ("version", C.c_uint32), ("dp_size", C.c_uint32), ("reserved", C.c_uint32),
                ("range_start", C.c_uint64 * 4), ("range_end", C.c_uint64 * 4), ("key_x", C.c_uint64 * 4),
                ("key_y", C.c_uint64 * 4), ("total_count", C.c_uint64), ("total_seconds", C.c_double)]


class _Config(C.Structure):
    _fields_ = [("range_start", C.c_uint64 * 4), ("range_end", C.c_uint64 * 4), ("key_x", C.c_uint64 * 4),
                ("key_y", C.c_uint64 * 4), ("dp", C.c_int32), ("n_gpus", C.c_int32), ("gpu_ids", C.c_int32 * MAX_GPUS),
                ("grid_x", C.c_int32), ("grid_y", C.c_int32), ("max_found", C.c_uint32), ("consumers", C.c_int32),
                ("seed", C.c_uint64), ("max_launches", C.c_uint64), ("warmup_launches", C.c_uint32), ("flags", C.c_uint32)]


class _Stats(C.Structure):
    _fields_ = [("jumps", C.c_uint64), ("launches", C.c_uint64), ("dps", C.c_uint64), ("dps_lost", C.c_uint64),
                ("same_herd", C.c_uint64), ("wrong_collisions", C.c_uint64), ("table_items", C.c_uint64),
                ("kangaroos", C.c_uint64), ("seconds", C.c_double), ("kernel_ms_avg", C.c_double), ("dp", C.c_int32),
                ("range_power", C.c_int32), ("solved", C.c_int32), ("running", C.c_int32), ("seed", C.c_uint64),
                ("herd_loaded", C.c_uint64), ("herd_created", C.c_uint64), ("table_bytes", C.c_uint64),
                ("warmup_jumps", C.c_uint64), ("audits", C.c_uint64), ("audited_kangaroos", C.c_uint64),
                ("audit_mismatches", C.c_uint64)]


class _HostStats(C.Structure):
    _fields_ = [("host_ms_max", C.c_double), ("host_ms_mean", C.c_double), ("ingest_ms_max", C.c_double),
                ("late_launches", C.c_uint64), ("queue_high_points", C.c_uint64), ("consumer_busy_max", C.c_double),
                ("consumer_busy_mean", C.c_double), ("run_seconds", C.c_double), ("consumers", C.c_uint32), ("numa_nodes", C.c_uint32),
                ("consumer_cpu_s", C.c_double), ("consumer_runq_s", C.c_double), ("consumer_busy_s", C.c_double),
                ("consumer_nvcsw", C.c_uint64), ("consumer_nivcsw", C.c_uint64), ("effective_cpus", C.c_double),
                ("pin_failures", C.c_uint64)]


class _AuditResult(C.Structure):
    _fields_ = [("kangaroos", C.c_uint64), ("kangaroo_mismatches", C.c_uint64), ("table_points", C.c_uint64),
                ("table_mismatches", C.c_uint64), ("herd_ms", C.c_double), ("table_ms", C.c_double), ("seconds", C.c_double),
                ("n_first_bad", C.c_uint32), ("reserved", C.c_uint32), ("first_bad", C.c_uint64 * 8)]


DP_RECORD_DTYPE = np.dtype([("x", np.uint64, (4,)), ("d", np.uint64, (2,)), ("kidx", np.uint64), ("reserved", np.uint64)])


_bound = False


def _lib() -> C.CDLL:
    global _bound
    L = hostlib.load()
    if not _bound:
        L.kngt_create.restype = C.c_void_p
        L.kngt_destroy.argtypes = [C.c_void_p]
        L.kngt_destroy.restype = None
        L.kngt_reset.argtypes = [C.c_void_p]
        L.kngt_reset.restype = None
        L.kngt_add.argtypes = [C.c_void_p, _U64P, _U64P, C.c_uint32, _U64P, C.POINTER(C.c_uint32)]
        L.kngt_count.argtypes = [C.c_void_p]
        L.kngt_count.restype = C.c_uint64
        L.kngt_bucket_count.argtypes = [C.c_void_p, C.c_uint32]
        L.kngt_bucket_count.restype = C.c_uint32
        L.kngt_bucket_entries.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32]
        L.kngt_bucket_entries.restype = C.c_uint32
        L.kngt_serialised_size.argtypes = [C.c_void_p]
        L.kngt_serialised_size.restype = C.c_uint64
        L.kngw_create.argtypes = [C.c_char_p, C.POINTER(_Header), C.c_void_p, C.c_uint64]
        L.kngw_create.restype = C.c_void_p
        L.kngw_put_kangaroos.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_uint64]
        L.kngw_open.argtypes = [C.c_char_p, C.POINTER(_Header), C.c_void_p, C.POINTER(C.c_uint64)]
        L.kngw_open.restype = C.c_void_p
        L.kngw_get_kangaroos.argtypes = [C.c_void_p, _U64P, _U64P, _U64P, C.c_uint64]
        L.kngw_close.argtypes = [C.c_void_p]
        L.kngw_last_error.restype = C.c_char_p
        L.kngs_create.argtypes = [C.POINTER(_Config), C.POINTER(C.c_void_p)]
        L.kngs_destroy.argtypes = [C.c_void_p]
        L.kngs_destroy.restype = None
        L.kngs_load.argtypes = [C.c_void_p, C.c_char_p]
        L.kngs_start.argtypes = [C.c_void_p]
        L.kngs_prepare.argtypes = [C.c_void_p]
        L.kngs_wait.argtypes = [C.c_void_p, C.c_double]
        L.kngs_stop.argtypes = [C.c_void_p]
        L.kngs_result.argtypes = [C.c_void_p, _U64P]
        L.kngs_get_stats.argtypes = [C.c_void_p, C.POINTER(_Stats)]
        L.kngs_save.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.kngs_audit.argtypes = [C.c_void_p, C.c_int, C.POINTER(_AuditResult)]
        L.kngs_host_stats.argtypes = [C.c_void_p, C.POINTER(_HostStats)]
        L.kngs_collision_key.argtypes = [C.c_void_p, _U64P, _U64P, _U64P]
        L.kngs_last_error.restype = C.c_char_p
        L.kngs_gpu_stats.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
        L.kngs_consumer_load.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.kngs_start_ingest.argtypes = [C.c_void_p, C.c_int]
        L.kngs_ingest.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint32]
        L.kngs_drained.argtypes = [C.c_void_p, C.c_double]
        L.kngs_table.argtypes = [C.c_void_p]
        L.kngs_table.restype = C.c_void_p
        L.kngt_memory_bytes.argtypes = [C.c_void_p]
        L.kngt_memory_bytes.restype = C.c_uint64
        L.kngt_encode_device.argtypes = [_U64P, _U64P, _U64P, C.c_uint64, C.POINTER(C.c_uint32), C.c_void_p]
        L.kngt_encode_device.restype = None
        L.kngt_encode.argtypes = [_U64P, _U64P, C.c_uint32, C.POINTER(C.c_uint32), C.c_void_p]
        L.kngt_encode.restype = None
        _bound = True
    return L


class HostError(RuntimeError):
    pass


class DpTable:
    """HashTable (HashTable.h:58-100): Add / GetNbItem / Reset + access to the sorted buckets."""

    def __init__(self, _borrowed=None):
        self._L = _lib()
        self._own = _borrowed is None
        self._h = self._L.kngt_create() if self._own else _borrowed
        if not self._h:
            raise MemoryError("kngt_create")

    def close(self):
        if self._h and self._own:
            self._L.kngt_destroy(self._h)
        self._h = None

    def memory_bytes(self) -> int:
        return int(self._L.kngt_memory_bytes(self._h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add(self, x: int, d: int, ktype: int):
        """Returns (status, other_d, other_type); other_* describe the stored kangaroo on ADD_COLLISION."""
        od = np.zeros(4, np.uint64)
        ot = C.c_uint32(0)
        st = self._L.kngt_add(self._h, limbs(x), limbs(d), ktype, od, C.byref(ot))
        if st < 0:
            raise MemoryError("kngt_add")
        return (st, to_int(od), int(ot.value)) if st == ADD_COLLISION else (st, None, None)

    def count(self) -> int:
        return int(self._L.kngt_count(self._h))

    def reset(self):
        self._L.kngt_reset(self._h)

    def bucket(self, h: int) -> np.ndarray:
        n = int(self._L.kngt_bucket_count(self._h, h))
        out = np.zeros(n, ENTRY_DTYPE)
        if n:
            self._L.kngt_bucket_entries(self._h, h, out.ctypes.data, n)
        return out

    def serialised_size(self) -> int:
        return int(self._L.kngt_serialised_size(self._h))


def _hdr_to_dict(h: _Header) -> dict:
    return dict(magic=h.magic, version=h.version, dp=h.dp_size, range_start=to_int(h.range_start), range_end=to_int(h.range_end),
                key=(to_int(h.key_x), to_int(h.key_y)), count=int(h.total_count), seconds=float(h.total_seconds))


def write_workfile(path: str, *, dp: int, range_start: int, range_end: int, key, count: int, seconds: float, table: DpTable,
                   kangaroos=None, magic: int = HEADW) -> None:
    """kangaroos: None or (x, y, d_true) arrays of shape (n, 4)."""
    L = _lib()
    h = _Header(magic=magic, version=0, dp_size=dp, total_count=count, total_seconds=seconds)
    for name, v in (("range_start", range_start), ("range_end", range_end), ("key_x", key[0]), ("key_y", key[1])):
        for i in range(4):
            getattr(h, name)[i] = (v >> (64 * i)) & ((1 << 64) - 1)
    n = 0 if kangaroos is None else len(kangaroos[0])
    f = L.kngw_create(path.encode(), C.byref(h), table._h if table is not None else None, n)
    if not f:
        raise HostError(L.kngw_last_error().decode())
    if n:
        x, y, d = (np.ascontiguousarray(a, dtype=np.uint64) for a in kangaroos)
        if L.kngw_put_kangaroos(f, x, y, d, n) != 0:
            L.kngw_close(f)
            raise HostError(L.kngw_last_error().decode())
    if L.kngw_close(f) != 0:
        raise HostError(L.kngw_last_error().decode())


def read_workfile(path: str, table: DpTable | None = None, with_kangaroos: bool = True):
    """Returns (header dict, n_kangaroos, (x, y, d_true) or None); fills `table` when given."""
    L = _lib()
    h = _Header()
    n = C.c_uint64(0)
    f = L.kngw_open(path.encode(), C.byref(h), table._h if table is not None else None, C.byref(n))
    if not f:
        raise HostError(L.kngw_last_error().decode())
    kang = None
    if with_kangaroos and n.value:
        x, y, d = (np.zeros((n.value, 4), np.uint64) for _ in range(3))
        if L.kngw_get_kangaroos(f, x, y, d, n.value) != 0:
            L.kngw_close(f)
            raise HostError(L.kngw_last_error().decode())
        kang = (x, y, d)
    L.kngw_close(f)
    return _hdr_to_dict(h), int(n.value), kang


class Solver:
    """Kangaroo::SolveKeyGPU for N GPUs (Kangaroo.cpp:510-644, :1019-1063) over the C-ABI engine."""

    def __init__(self, range_start: int, range_end: int, key, *, gpus=(0,), grid=(0, 0), dp: int = -1, max_found: int = 0,
                 consumers: int = 0, seed: int = 0, max_launches: int = 0, warmup_launches: int = 0, flags: int = 0):
        """seed 0 (default) draws a herd seed like the reference does from the clock (main.cpp:177); stats()["seed"]
        reports it.  Pass a fixed seed only for tests and benchmarks: equal seeds rebuild equal herds."""
        self._L = _lib()
        cfg = _Config(dp=dp, n_gpus=len(gpus), grid_x=grid[0], grid_y=grid[1], max_found=max_found, consumers=consumers,
                      seed=seed & ((1 << 64) - 1), max_launches=max_launches, warmup_launches=warmup_launches, flags=flags)
        for name, v in (("range_start", range_start), ("range_end", range_end), ("key_x", key[0]), ("key_y", key[1])):
            for i in range(4):
                getattr(cfg, name)[i] = (v >> (64 * i)) & ((1 << 64) - 1)
        for i, g in enumerate(gpus):
            cfg.gpu_ids[i] = g
        self._h = C.c_void_p()
        if self._L.kngs_create(C.byref(cfg), C.byref(self._h)) != 0:
            raise HostError(self._L.kngs_last_error().decode())

    def _check(self, rc: int) -> int:
        if rc < 0:
            raise HostError(self._L.kngs_last_error().decode())
        return rc

    def close(self):
        if self._h:
            self._L.kngs_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load(self, path: str):
        self._check(self._L.kngs_load(self._h, path.encode()))

    def prepare(self):
        """engines, herds and the warm-up launches; start() does it when it has not been done"""
        self._check(self._L.kngs_prepare(self._h))

    def start(self):
        self._check(self._L.kngs_start(self._h))

    def wait(self, seconds: float) -> int:
        """1 = solved, 2 = every GPU reached max_launches, 0 = timeout."""
        return self._check(self._L.kngs_wait(self._h, seconds))

    def stop(self):
        self._check(self._L.kngs_stop(self._h))

    def result(self) -> int:
        out = np.zeros(4, np.uint64)
        self._check(self._L.kngs_result(self._h, out))
        return to_int(out)

    def stats(self) -> dict:
        st = _Stats()
        self._check(self._L.kngs_get_stats(self._h, C.byref(st)))
        return {k: getattr(st, k) for k, _ in _Stats._fields_}

    def collision_key(self, tame_d: int, wild_d: int):
        """CollisionCheck/CheckKey (Kangaroo.cpp:233-329): the private key from the true distances (mod n) of a tame
        and a wild kangaroo on the same point, or None when the collision does not resolve."""
        out = np.zeros(4, np.uint64)
        rc = self._check(self._L.kngs_collision_key(self._h, limbs(tame_d), limbs(wild_d), out))
        return to_int(out) if rc == 1 else None

    def gpu_stats(self, gpu: int) -> dict:
        """launches, summed kernel milliseconds (HIP events) and herd size of one GPU of this run"""
        l, ms, n = C.c_uint64(0), C.c_double(0), C.c_uint64(0)
        self._check(self._L.kngs_gpu_stats(self._h, gpu, C.byref(l), C.byref(ms), C.byref(n)))
        return {"launches": int(l.value), "kernel_ms_sum": float(ms.value), "kangaroos": int(n.value)}

    def gpu_option(self, gpu: int, key: str) -> int:
        """an option of one GPU's engine (kng_get_option): group, lanes, share, dsplit, asm, ..."""
        v = C.c_int64(0)
        self._check(self._L.kngs_gpu_option(self._h, gpu, key.encode(), C.byref(v)))
        return int(v.value)

    def consumer_load(self) -> list:
        buf = (C.c_uint64 * 64)()
        n = self._check(self._L.kngs_consumer_load(self._h, buf, 64))
        return [int(buf[i]) for i in range(min(n, 64))]

    # ---- the host path without engines (kng_solver.h: measurements and tests of the DP ingest) ----
    def start_ingest(self, feeders: int):
        self._check(self._L.kngs_start_ingest(self._h, feeders))

    def ingest(self, feeder: int, records: np.ndarray):
        assert records.dtype == DP_RECORD_DTYPE and records.flags.c_contiguous
        self._check(self._L.kngs_ingest(self._h, feeder, records.ctypes.data, len(records)))

    def drained(self, seconds: float) -> bool:
        return self._check(self._L.kngs_drained(self._h, seconds)) == 1

    def table(self) -> "DpTable":
        """read-only view of the solver's table (exact while no point is in flight)"""
        return DpTable(_borrowed=self._L.kngs_table(self._h))

    def save(self, path: str, with_kangaroos: bool = True):
        self._check(self._L.kngs_save(self._h, path.encode(), 1 if with_kangaroos else 0))

    def audit(self, with_table: bool = True) -> dict:
        """Whole-run audit on the device (kngs_audit): every kangaroo of every herd -- and every table entry -- re-derived
        from its distance.  Returns the counts; a clean run has kangaroo_mismatches == table_mismatches == 0."""
        r = _AuditResult()
        self._check(self._L.kngs_audit(self._h, 1 if with_table else 0, C.byref(r)))
        out = {k: getattr(r, k) for k, _ in _AuditResult._fields_ if k not in ("first_bad", "reserved", "n_first_bad")}
        out["first_bad"] = [(int(v) >> 56, int(v) & ((1 << 56) - 1)) for v in list(r.first_bad)[: r.n_first_bad]]
        return out

    def host_stats(self) -> dict:
        """kngs_host_stats: is the host keeping up (host ms per launch, late launches, queue high-water, consumer load)"""
        h = _HostStats()
        self._check(self._L.kngs_host_stats(self._h, C.byref(h)))
        return {k: (round(getattr(h, k), 4) if isinstance(getattr(h, k), float) else getattr(h, k)) for k, _ in _HostStats._fields_ if k != "reserved"}
