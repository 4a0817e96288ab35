import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    """Vectors produced by the reference's own objects (tools/make_golden.py, oracle/refprobe.cpp)."""
    with open(os.path.join(ROOT, "tests", "golden", "ref_vectors.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def orc():
    from oracle import load_oracle

    return load_oracle()


@pytest.fixture(scope="session")
def kng():
    """The engine library on a real device (GPU tests only): raises instead of falling back."""
    import kangaroo_amd

    kangaroo_amd.load_library()  # raises if the HIP library was not built: no fallback
    assert kangaroo_amd.device_count() >= 1, "no HIP device visible"
    info = kangaroo_amd.device_info(0)
    assert "gfx950" in info["arch"], info
    return kangaroo_amd
