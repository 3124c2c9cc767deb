import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Every test session starts from built artefacts (HIP library + CPU oracle); a no-op when they are up to date."""
    import __graft_entry__ as ge
    ge.build()


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle

    oracle.build()
    return oracle
