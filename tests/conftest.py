import os
import sys

import pytest

# torch BEFORE libamwg.so is loaded.  The torch wheel carries its own copy of the HIP runtime; libamwg.so links the system one.  Whichever
# is loaded first serves both (same soname) -- but only if torch comes first: loaded after libamwg.so, torch brings a second runtime into the
# process, and the second HIP runtime to start finds no device on the GPU box ("No HIP GPUs are available" / "no ROCm-capable device is
# detected", depending on who lost).  The tests that hand torch buffers to the library (bench helpers) then depended on which test files
# pytest happened to collect.  bench.py imports torch first for the same reason.
try:
    import torch  # noqa: F401
except Exception:      # the CPU-only suite does not need it
    pass

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

