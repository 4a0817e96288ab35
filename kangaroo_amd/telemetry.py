"""Package power and GFX clock of the devices a measurement runs on, sampled from a thread (librocm_smi64 through ctypes,
no subprocess: a sample costs microseconds, so 50 Hz is cheap), plus the device's energy accumulator read at both ends
of the window -- the average power that does not depend on how many samples fell into a short window.

Why this is part of the bench line: the walk kernel sits at the package power cap, and "the
remaining distance is power, not scheduling"  has to be checkable from the driver-run JSON itself.
Plumbing only; nothing on the data path.  Every call degrades to "unavailable" instead of raising.
"""
from __future__ import annotations

import ctypes as C
import threading
import time

_RSMI_MAX_FREQ = 33


class _Freqs(C.Structure):
    _fields_ = [("has_deep_sleep", C.c_bool), ("num_supported", C.c_uint32), ("current", C.c_uint32),
                ("frequency", C.c_uint64 * _RSMI_MAX_FREQ)]


_lib = None
_lib_err = None


def _rsmi():
    global _lib, _lib_err
    if _lib is not None or _lib_err is not None:
        return _lib
    for name in ("librocm_smi64.so", "/opt/rocm/lib/librocm_smi64.so", "librocm_smi64.so.1"):
        try:
            L = C.CDLL(name)
            if L.rsmi_init(C.c_uint64(0)) != 0:
                _lib_err = "rsmi_init failed"
                return None
            _lib = L
            return L
        except OSError as e:
            _lib_err = str(e)
    return None


def _power_w(L, dev):
    v = C.c_uint64(0)
    if L.rsmi_dev_current_socket_power_get(C.c_uint32(dev), C.byref(v)) == 0 and v.value:
        return v.value * 1e-6
    if L.rsmi_dev_power_ave_get(C.c_uint32(dev), C.c_uint32(0), C.byref(v)) == 0 and v.value:
        return v.value * 1e-6
    return None


def _sclk_mhz(L, dev):
    f = _Freqs()
    if L.rsmi_dev_gpu_clk_freq_get(C.c_uint32(dev), C.c_int(0), C.byref(f)) != 0:  # RSMI_CLK_TYPE_SYS
        return None
    if f.current < f.num_supported and f.current < _RSMI_MAX_FREQ:
        return f.frequency[f.current] * 1e-6
    return None


def _energy_j(L, dev):
    e, res, ts = C.c_uint64(0), C.c_float(0), C.c_uint64(0)
    try:
        if L.rsmi_dev_energy_count_get(C.c_uint32(dev), C.byref(e), C.byref(res), C.byref(ts)) != 0:
            return None
    except AttributeError:
        return None
    return e.value * float(res.value) * 1e-6  # counter x resolution (micro-joules)


def power_cap_w(dev: int = 0):
    L = _rsmi()
    if L is None:
        return None
    v = C.c_uint64(0)
    if L.rsmi_dev_power_cap_get(C.c_uint32(dev), C.c_uint32(0), C.byref(v)) == 0 and v.value:
        return v.value * 1e-6
    return None


def _stats(vals):
    vals = sorted(v for v in vals if v is not None)
    if not vals:
        return None
    return {"median": round(vals[len(vals) // 2], 1), "mean": round(sum(vals) / len(vals), 1), "min": round(vals[0], 1),
            "max": round(vals[-1], 1), "samples": len(vals)}


class GpuSampler:
    """with GpuSampler([0]) as s: ...timed region...;  s.summary() -> per-device power / clock over the region."""

    def __init__(self, devices=(0,), hz: float = 50.0):
        self.devices = tuple(devices)
        self.period = 1.0 / hz
        self.hz = hz
        self._L = _rsmi()
        self._stop = threading.Event()
        self._th = None
        self._samples = {d: [] for d in self.devices}
        self._e0 = self._e1 = None
        self._t0 = self._t1 = None

    def start(self):
        self._t0 = time.perf_counter()
        if self._L is None:
            return self
        self._e0 = {d: _energy_j(self._L, d) for d in self.devices}
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def _run(self):
        nxt = time.perf_counter()
        while not self._stop.is_set():
            for d in self.devices:
                self._samples[d].append((_power_w(self._L, d), _sclk_mhz(self._L, d)))
            nxt += self.period
            delay = nxt - time.perf_counter()
            if delay > 0:
                self._stop.wait(delay)
            else:
                nxt = time.perf_counter()

    def stop(self):
        self._t1 = time.perf_counter()
        if self._L is not None:
            self._e1 = {d: _energy_j(self._L, d) for d in self.devices}
        self._stop.set()
        if self._th is not None:
            self._th.join()
        return self

    __enter__ = start

    def __exit__(self, *exc):
        self.stop()

    def summary(self) -> dict:
        if self._L is None:
            return {"available": False, "reason": _lib_err or "librocm_smi64 not found"}
        out = {"available": True, "hz": self.hz, "window_s": round((self._t1 or time.perf_counter()) - self._t0, 4), "devices": []}
        for d in self.devices:
            pw = _stats([s[0] for s in self._samples[d]])
            ck = _stats([s[1] for s in self._samples[d]])
            row = {"device": d, "power_w": pw, "sclk_mhz": ck, "power_cap_w": power_cap_w(d)}
            e0, e1 = (self._e0 or {}).get(d), (self._e1 or {}).get(d)
            if e0 is not None and e1 is not None and e1 > e0 and self._t1:
                row["energy_j"] = round(e1 - e0, 3)
                row["power_w_from_energy_counter"] = round((e1 - e0) / (self._t1 - self._t0), 1)
            out["devices"].append(row)
        return out
