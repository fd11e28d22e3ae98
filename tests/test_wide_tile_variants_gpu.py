"""GPU tier: the training-path tests once with every eligible 3x3 layer forced onto conv_mp.hip and once onto conv_mq.hip
(RYOLO_CONV3X3 = mp | mq; the automatic choice is a cost model, so a given test shape exercises only one of them).  The
statistics (training forward) and data-gradient instantiations of BOTH kernels see the same cases and the same bars."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("mode", ["mp", "mq"])
def test_training_ops_with_the_wide_tile_forced(cuda_dev, mode):
    env = dict(os.environ, RYOLO_CONV3X3=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(ROOT, "tests", "test_train_ops_gpu.py"),
                        os.path.join(ROOT, "tests", "test_train_engine_gpu.py"),
                        "-k", "conv_stats or dgrad or composed_backward or train_step_matches or second_step"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
