import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built with gcc on first use."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def meao_lib():
    """libmeao_hip.so through ctypes (the product's C ABI)."""
    from miniengineao_amd import _lib
    return _lib.load()
