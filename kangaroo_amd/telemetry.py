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


def _call(L, name, *args):
    """rsmi status of L.<name>(*args), or None when this librocm_smi64 does not export the symbol (older / newer ROCm): the
    module's promise is "unavailable", never an exception in the sampler thread or inside a timed region."""
    try:
        fn = getattr(L, name)
    except AttributeError:
        return None
    try:
        return fn(*args)
    except Exception:  # noqa: BLE001  (ctypes.ArgumentError and friends)
        return None


def _power_w(L, dev):
    v = C.c_uint64(0)
    if _call(L, "rsmi_dev_current_socket_power_get", C.c_uint32(dev), C.byref(v)) == 0 and v.value:
        return v.value * 1e-6
    if _call(L, "rsmi_dev_power_ave_get", C.c_uint32(dev), C.c_uint32(0), C.byref(v)) == 0 and v.value:
        return v.value * 1e-6
    return None


def _sclk_mhz(L, dev):
    f = _Freqs()
    if _call(L, "rsmi_dev_gpu_clk_freq_get", C.c_uint32(dev), C.c_int(0), C.byref(f)) != 0:  # RSMI_CLK_TYPE_SYS
        return None
    if f.current < f.num_supported and f.current < _RSMI_MAX_FREQ:
        return f.frequency[f.current] * 1e-6
    return None


def _energy_j(L, dev):
    e, res, ts = C.c_uint64(0), C.c_float(0), C.c_uint64(0)
    if _call(L, "rsmi_dev_energy_count_get", C.c_uint32(dev), C.byref(e), C.byref(res), C.byref(ts)) != 0:
        return None
    return e.value * float(res.value) * 1e-6  # counter x resolution (micro-joules)


def power_cap_w(dev: int = 0):
    """Package power cap of rocm_smi device `dev` (an SMI index: see smi_index_of)."""
    L = _rsmi()
    if L is None:
        return None
    v = C.c_uint64(0)
    if _call(L, "rsmi_dev_power_cap_get", C.c_uint32(dev), C.c_uint32(0), C.byref(v)) == 0 and v.value:
        return v.value * 1e-6
    return None


# ---- which rocm_smi device is HIP device i?  Not "i": HIP numbers the devices the process may see (HIP_VISIBLE_DEVICES /
# ROCR_VISIBLE_DEVICES renumber and reorder them), rocm_smi numbers the cards of the machine.  The PCI address ties them.
def bdf_of_smi_id(bdfid: int) -> str:
    """rsmi_dev_pci_id_get's packed id -> "dddd:bb:dd.f" (domain bits 63-32, bus 15-8, device 7-3, function 2-0; the partition
    id that newer ROCm keeps in bits 31-28 is not part of the address)."""
    return "%04x:%02x:%02x.%x" % ((bdfid >> 32) & 0xFFFFFFFF, (bdfid >> 8) & 0xFF, (bdfid >> 3) & 0x1F, bdfid & 0x7)


def smi_bdfs(L=None):
    """PCI address of every rocm_smi device, by SMI index ([] when unavailable)."""
    L = L or _rsmi()
    if L is None:
        return []
    n = C.c_uint32(0)
    if _call(L, "rsmi_num_monitor_devices", C.byref(n)) != 0:
        return []
    out = []
    for i in range(n.value):
        v = C.c_uint64(0)
        out.append(bdf_of_smi_id(v.value) if _call(L, "rsmi_dev_pci_id_get", C.c_uint32(i), C.byref(v)) == 0 else None)
    return out


def hip_bdfs(devices):
    """PCI address of each HIP device index (kng_device_pci_bdf), None where it cannot be read."""
    out = {}
    try:
        from . import load_library

        lib = load_library()
        for d in devices:
            buf = C.create_string_buffer(64)
            out[d] = buf.value.decode().lower() if lib.kng_device_pci_bdf(int(d), buf, 64) == 0 else None
    except Exception:  # noqa: BLE001
        out = {d: None for d in devices}
    return out


def map_devices(hip: dict, smi: list) -> dict:
    """{hip index: (smi index, how)}: by PCI address where both sides know it, else the same index ("index (assumed)"), else
    None when rocm_smi has no such device.  Pure: tests feed it made-up address lists."""
    out = {}
    for d, bdf in hip.items():
        if bdf is not None and bdf in smi:
            out[d] = (smi.index(bdf), "pci")
        elif smi and all(b is None for b in smi) or bdf is None:
            out[d] = (d, "index (assumed)") if (not smi or d < len(smi)) else (None, "no such rocm_smi device")
        else:
            out[d] = (None, "pci address %s not among rocm_smi's devices" % bdf)
    return out


def _stats(vals):
    vals = sorted(v for v in vals if v is not None)
    if not vals:
        return None
    return {"median": round(vals[len(vals) // 2], 1), "mean": round(sum(vals) / len(vals), 1), "min": round(vals[0], 1),
            "max": round(vals[-1], 1), "samples": len(vals)}


class GpuSampler:
    """with GpuSampler([0]) as s: ...timed region...;  s.summary() -> per-device power / clock over the region."""

    def __init__(self, devices=(0,), hz: float = 50.0):
        self.devices = tuple(devices)  # HIP indices
        self.period = 1.0 / hz
        self.hz = hz
        self._L = _rsmi()
        self._bdf = hip_bdfs(self.devices) if self._L is not None else {d: None for d in self.devices}
        self._map = map_devices(self._bdf, smi_bdfs(self._L)) if self._L is not None else {d: (None, "unavailable") for d in self.devices}
        self._stop = threading.Event()
        self._th = None
        self._samples = {d: [] for d in self.devices}
        self._e0 = self._e1 = None
        self._t0 = self._t1 = None

    def start(self):
        self._t0 = time.perf_counter()
        if self._L is None:
            return self
        self._e0 = {d: self._read(_energy_j, d) for d in self.devices}
        self._th = threading.Thread(target=self._run, daemon=True)
        self._th.start()
        return self

    def _read(self, fn, d):
        smi = self._map[d][0]
        return None if smi is None else fn(self._L, smi)

    def _run(self):
        nxt = time.perf_counter()
        while not self._stop.is_set():
            for d in self.devices:
                self._samples[d].append((self._read(_power_w, d), self._read(_sclk_mhz, d)))
            nxt += self.period
            delay = nxt - time.perf_counter()
            if delay > 0:
                self._stop.wait(delay)
            else:
                nxt = time.perf_counter()

    def stop(self):
        self._t1 = time.perf_counter()
        if self._L is not None:
            self._e1 = {d: self._read(_energy_j, d) for d in self.devices}
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
            smi, how = self._map[d]
            row = {"device": d, "pci": self._bdf.get(d), "smi_index": smi, "smi_mapping": how, "power_w": pw, "sclk_mhz": ck,
                   "power_cap_w": power_cap_w(smi) if smi is not None else None}
            e0, e1 = (self._e0 or {}).get(d), (self._e1 or {}).get(d)
            if e0 is not None and e1 is not None and e1 > e0 and self._t1:
                row["energy_j"] = round(e1 - e0, 3)
                row["power_w_from_energy_counter"] = round((e1 - e0) / (self._t1 - self._t0), 1)
            out["devices"].append(row)
        return out
