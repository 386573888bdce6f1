import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

GOLD = os.path.join(ROOT, "tests", "golden")
SCENES = os.path.join(ROOT, "scenes")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """Incremental `make` of every native piece (no-op when the shipped .so files are current)."""
    import __graft_entry__ as ge
    return ge.build()


@pytest.fixture(scope="session")
def oracle(built):
    import oracle as orc  # oracle/oracle.py (test infrastructure)
    orc.lib()
    return orc


@pytest.fixture(scope="session")
def pt(built):
    """The product: ctypes binding over the C-ABI (include/pt_amd.h)."""
    return built
