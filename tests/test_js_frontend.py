"""The JavaScript host side (north_star: host code stays JavaScript over an N-API addon).

test_frontend.js: parameter completion KATs, option merging, model recognition, errors (no GPU).
test_gpu.js (-m gpu): the README programs, verbatim, through bayes.js_amd on the GPU reproduce the
seeded reference runs of tests/golden/ bit for bit.
"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NODE = shutil.which("node")
needs_node = pytest.mark.skipif(NODE is None, reason="node is not installed")


def run_node(script, timeout):
    p = subprocess.run([NODE, os.path.join(ROOT, "tests", "js", script)], cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout + "\n" + p.stderr
    return p.stdout


@needs_node
def test_js_frontend_host_logic():
    assert "frontend ok" in run_node("test_frontend.js", 120)


@needs_node
@pytest.mark.gpu
def test_js_frontend_on_gpu_matches_reference_goldens():
    assert "gpu frontend ok" in run_node("test_gpu.js", 600)


@needs_node
@pytest.mark.gpu
def test_translated_closures_on_gpu_match_reference_goldens():
    """Arbitrary user closures (real/int/binary params, every scalar ld.*, derived quantities) through translate.js + hiprtc."""
    assert "gpu user models ok" in run_node("test_gpu_user.js", 1200)
