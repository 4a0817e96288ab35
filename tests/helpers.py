"""Shared helpers for the parity tests (test-side only; may use the oracle)."""
from __future__ import annotations

import numpy as np

from oracle.binding import N_ORDER, P, array_to_ints, from_limbs, ints_to_array, to_limbs  # noqa: F401

M128 = (1 << 128) - 1


def walk_fixture(g):
    """Decode a golden 'walk_*' record -> dict of python ints / arrays."""
    start = [(int(x, 16), int(y, 16), int(d, 16)) for x, y, d in g["start"]]
    end = [(int(x, 16), int(y, 16), int(d, 16)) for x, y, d in g["end"]]
    dps = [(int(i), int(x, 16), int(d, 16)) for i, x, d in g["dps"]]
    return dict(
        start=start, end=end, dps=dps, nsteps=g["nsteps"], dp_mask=int(g["dp_mask"], 16),
        wild_offset=int(g["wild_offset"], 16), range_power=g["range_power"],
        key_to_search=tuple(int(v, 16) for v in g["key_to_search"]),
    )


def device_distances(true_d, wild_offset):
    """GPUEngine.cu:406-411: odd (wild) indices carry d + wildOffset mod n on the device."""
    out = []
    for i, d in enumerate(true_d):
        if i % 2 == 1:
            d = (d + wild_offset) % N_ORDER
        assert d <= M128, "device distance must fit 128 bits"
        out.append(d)
    return out


def host_distance(dev_d, kidx, wild_offset):
    """GPUEngine.cu:477,672: wild distances have the offset removed mod n on the way back."""
    return (dev_d - wild_offset) % N_ORDER if kidx % 2 == 1 else dev_d


def dp_multiset(dps):
    """Canonical sorted list of (kidx, x, d) for exact multiset comparison."""
    return sorted((int(k), int(x), int(d)) for k, x, d in dps)


def random_field_elems(rng: np.random.Generator, n: int) -> np.ndarray:
    a = rng.integers(0, 1 << 64, size=(n, 4), dtype=np.uint64)
    return a
