"""The gfx950 code objects of the benched step-kernel instantiations carry no spilled VGPR and no scratch traffic inside their loops
(tools/isa_audit.py: hipcc cross-compiles here, no GPU needed).  Round 2 compiled every instantiation for 1024-thread workgroups and
carried 31 scratch instructions through the slot loop of the cfg4 kernel."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_benched_kernels_have_no_vgpr_spills_or_scratch_in_their_loops():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_audit.py")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
    assert "isa audit ok" in p.stdout
