import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle_lib():
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def cgo():
    """The product package; builds libcgo_hip.so with hipcc if it is not there (cross-compiles on CPU)."""
    import cgo_amd
    from cgo_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return cgo_amd


@pytest.fixture(scope="session")
def gpu_ctx(cgo):
    return cgo.default_context()
