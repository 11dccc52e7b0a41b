import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """tests/test_aa_*.py start child processes (ranks, a world-1 RCCL group) before this process initialises the GPU: keep them
    ahead of every other test whatever the collection order."""
    items.sort(key=lambda it: 0 if "test_aa_" in it.nodeid else 1)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (plain C via ctypes). Test infrastructure only."""
    from oracle.qgtc_oracle import Oracle

    return Oracle()


@pytest.fixture(scope="session")
def qgtc():
    """The compiled QGTC extension on a real GPU. No fallback: a missing .so is an error."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    import QGTC

    return QGTC


@pytest.fixture(autouse=True)
def _library_switches_back_to_default(request):
    """Process-wide switches of the binding (engine, zero-tile skipping) never leak from one GPU test into the
    next: every test starts on the shipped defaults."""
    yield
    if request.node.get_closest_marker("gpu") is None:
        return
    mod = sys.modules.get("QGTC")
    if mod is not None:
        mod.set_engine("auto")
        mod.set_zero_skip(True)
