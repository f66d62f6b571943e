import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def built_lib():
    """The in-tree HIP library (built on demand; hipcc cross-compiles without a GPU)."""
    from wespeaker_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from wespeaker_amd import build
        build.build(verbose=False)
    return _lib.lib()
