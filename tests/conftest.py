import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Compile libclpgpu.so (hipcc, cross-compiles on CPU) and the oracle once per session."""
    import __graft_entry__ as g

    g.build()
    return True


@pytest.fixture(scope="session")
def afiro():
    from clp_amd.mps import read_mps

    return read_mps(os.path.join(ROOT, "tests", "golden", "afiro.mps"))
