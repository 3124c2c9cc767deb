import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

try:
    # PyTorch bundles its own libamdhip64 / libhsa-runtime64.  Whichever copy of the HIP runtime is mapped first serves every later
    # user of that soname, so importing torch before libsmr_hip.so is loaded keeps ONE runtime in the process whatever subset of
    # the tests is selected; the other order leaves two, and a torch stream handed to smr_ctx_create then belongs to the wrong one.
    import torch  # noqa: F401
except ImportError:  # the C ABI itself does not need torch
    pass


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
