"""kangaroo_amd -- MI355X-native (gfx950) Pollard's-kangaroo jump engine.

One hot path of JeanLucPons/Kangaroo (the per-herd random walk behind `class GPUEngine`,
GPU/GPUEngine.h:40-84) rebuilt from scratch as hand-written HIP behind a C ABI
(include/kangaroo_hip.h).  This Python package is plumbing only: it builds the library and binds
the C ABI with ctypes for the test-suite and bench.py.  There is no CPU fallback: importing works
anywhere, creating an engine needs a gfx950 device.
"""
from .engine import (KNG_GRP_SIZE, KNG_NB_JUMP, KNG_NB_RUN, EngineError, GPUEngine, device_count,  # noqa: F401
                     device_info, device_free_bytes, default_grid, load_library, test_fieldop)

__all__ = ["GPUEngine", "EngineError", "device_count", "device_info", "device_free_bytes", "default_grid", "load_library",
           "test_fieldop", "KNG_NB_JUMP", "KNG_NB_RUN", "KNG_GRP_SIZE"]
