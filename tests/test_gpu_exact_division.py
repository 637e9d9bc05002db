"""The exact-division identities of dvo_device_math.h, checked over all 2^32 float bit patterns on the GPU."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_exact_division_identities_exhaustive():
    exe = os.path.join(ROOT, "tools", "exhaustive", "bin", "div_tricks")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ALL IDENTITIES HOLD" in out.stdout, out.stdout + out.stderr
