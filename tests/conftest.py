import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure). Built on demand with gcc."""
    from oracle import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def reg():
    """The HIP library through its ctypes binding; GPU tests only."""
    from mandala_mapping_amd import binding
    return binding
