import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "bayes.js_amd"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "node: needs the `node` binary")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Build the oracle and the HIP library once per session (no-op when up to date)."""
    import __graft_entry__ as ge
    ge.build()


@pytest.fixture(scope="session", autouse=True)
def _torch_sees_the_gpu_first():
    """On a GPU box: let torch start its HIP context before the first sampler is created, so that the tests which hand torch device
    buffers to the library (bench helpers) do not depend on the order the test files run in."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:      # no torch / no GPU: the CPU suite does not need it
        pass
