"""CPU tier (needs hipcc, no GPU): static checks of the generated gfx950 code that the pipelined kernels depend on --
tools/check_mp_isa.py: no register spill inside conv_mp's counted-wait K loop, and no compiler-inserted full vmcnt wait (or
spill) inside wgrad_wide_kernel's three-stage loop; and, over every translation unit, no VALU write to the data registers of a
12-/16-B buffer store fewer than two wait states behind it (the compiler skips that hazard for register-soffset stores, gfx950
does not)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_pipelined_kernels_keep_their_counted_waits():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_mp_isa.py")], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "wgrad_wide_kernel" in r.stdout and "conv_mp_kernel" in r.stdout
    assert "STORE-DATA HAZARD" not in r.stdout and "conv_mp.hip" in r.stdout
