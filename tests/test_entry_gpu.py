"""GPU tier: the driver's entry points in ONE process, build() first (it imports the package and loads the library before
any CUDA work) and then smoke() -- the order that used to leave two HIP runtimes in the process (rotate-yolov3_amd/_lib.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_build_then_smoke_in_one_process(cuda_dev):
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke(); print('ENTRY_OK')"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "ENTRY_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
