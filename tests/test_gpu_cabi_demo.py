"""GPU: the C ABI driven from a plain C++ program (hipMalloc'ed buffers, no Python, no torch):
tests/cabi/scan_demo.cpp, built by __graft_entry__.build()."""
import os
import subprocess

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_cabi_scan_demo_is_bit_exact():
    exe = os.path.join(ROOT, "tests", "cabi", "scan_demo")
    if not os.path.exists(exe):
        import __graft_entry__
        __graft_entry__.build_cabi_demo()
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("bit-exact") == 5 and "labels bit-exact" in out.stdout
    assert "bad m -> rc=-1" in out.stdout
