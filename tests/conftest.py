import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def sb():
    """The product package (directory name has a hyphen, so it is loaded by path)."""
    import __graft_entry__ as ge
    ge.build()  # make is a no-op when the in-tree .so files are current; a fresh checkout gets them built
    return ge.load_package()


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.build()
    return orc
