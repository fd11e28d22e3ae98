"""A Darknet .weights file and a checkpoint dict WRITTEN BY THE REFERENCE (model/model_utils.py:save_weights :95-118,
torch.save of the train.py:323-363 checkpoint layout) for a small rotated-YOLO cfg, plus the parameter values that went in.
The product must load both (tests/test_oracle_model.py).  Only data is stored: the cfg text below is this script's own.

    python tests/golden/gen_weights_golden.py        (needs /root/reference)
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

MINI = """[net]
width=64
height=64
channels=3

[convolutional]
batch_normalize=1
filters=8
size=3
stride=2
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=3
stride=2
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=8
size=1
stride=1
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=3
stride=1
pad=1
activation=leaky

[shortcut]
from=-3
activation=linear

[convolutional]
size=1
stride=1
pad=1
filters=21
activation=linear

[yolo]
mask = 6-8
anchors = ara 800, 3000, 9000 / 4.0 / -60, 0, 60
classes=1
num=9

[route]
layers = -3

[convolutional]
batch_normalize=1
filters=8
size=1
stride=1
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=21
activation=linear

[yolo]
mask = 3-5
anchors = ara 800, 3000, 9000 / 4.0 / -60, 0, 60
classes=1
num=9

[route]
layers = -3

[convolutional]
size=1
stride=1
pad=1
filters=21
activation=linear

[yolo]
mask = 0-2
anchors = ara 800, 3000, 9000 / 4.0 / -60, 0, 60
classes=1
num=9
"""


def main():
    from tests.golden.gen_model_golden import install_stubs
    install_stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from model import models as rmodels
    from model import model_utils as rmu
    tmp = tempfile.mkdtemp()
    cfg = os.path.join(tmp, "mini.cfg")
    open(cfg, "w").write(MINI)
    model = rmodels.Darknet(cfg, {"context_factor": 1.0})
    g = torch.Generator().manual_seed(21)
    with torch.no_grad():
        for k, t in model.state_dict().items():
            if t.dtype.is_floating_point:
                t.copy_(torch.randn(t.shape, generator=g) * 0.5 + (1.0 if "running_var" in k else 0.0))
                if "running_var" in k:
                    t.abs_()
    model.seen = np.array([12345], dtype=np.int64)
    wpath = os.path.join(OUT, "ref_saved_mini.weights")
    rmu.save_weights(model, wpath, cutoff=-1)
    ck = {"epoch": 3, "best_fitness": 0.25, "training_results": "row\n", "model": model.state_dict(), "optimizer": None}
    torch.save(ck, os.path.join(OUT, "ref_saved_mini.pt"))
    np.savez_compressed(os.path.join(OUT, "ref_saved_mini_state.npz"), cfg=np.array(MINI),
                        **{k: v.numpy() for k, v in model.state_dict().items()})
    print("wrote", wpath, os.path.getsize(wpath), "bytes;", len(model.state_dict()), "state entries")


if __name__ == "__main__":
    main()
