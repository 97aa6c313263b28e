import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def pkg():
    """The cortex.llamacpp_amd package (directory name has a dot, so it is loaded by path)."""
    import __graft_entry__ as ge
    return ge.load_pkg()


@pytest.fixture(scope="session")
def tmp_models(tmp_path_factory):
    return tmp_path_factory.mktemp("models")
