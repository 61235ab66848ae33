import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the oracle runs on the CPU: keep torch's thread pool within the container's CPU quota (cavp_amd/hostinfo.py - 128 threads on
    # a 16-CPU quota made the oracle comparisons, and every eager launch sequence beside them, several times slower)
    from cavp_amd.hostinfo import cap_torch_threads
    cap_torch_threads()


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(REPO, "tests", "golden")


@pytest.fixture
def deterministic():
    """cavp_set_deterministic for the duration of one test (fixed-order reductions: run-to-run bit-reproducible steps), so that
    comparisons between two runs of the training step can be held to rounding instead of to the f32-atomics noise."""
    import torch
    from cavp_amd import _lib
    _lib.set_deterministic(True, torch.device("cuda", 0))
    try:
        yield
    finally:
        _lib.set_deterministic(False)
