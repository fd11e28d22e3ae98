"""CPU tier: train.py end to end on a tiny cfg with synthetic data (1 epoch), checkpoint + results.txt formats."""
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_train_py_runs_and_saves_reference_shaped_checkpoint(tmp_path):
    from tests.test_dp_gloo import CFG
    cfg = tmp_path / "m.cfg"
    cfg.write_text(CFG)
    hyp = tmp_path / "hyp.py"
    hyp.write_text("giou: 0.1\ncls: 27.76\ncls_pw: 1.446\nobj: 20.35\nobj_pw: 3.941\niou_t: 0.3\nang_t: 3.1415926/12\n"
                   "reg: 1.0\nfl_gamma: 0.5\ncontext_factor: 1.0\nlr0: 0.0001\nmultiplier:10\nwarm_epoch:1\nmomentum: 0.97\n"
                   "weight_decay: 0.0004569\nepochs: 2\nbatch_size: 2\nsave_interval: 300\ntest_interval: 5\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "train.py"), "--cfg", str(cfg), "--hyp", str(hyp), "--img-size", "64",
                        "--synthetic", "4", "--device", "cpu", "--wdir", str(tmp_path / "w")], cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    ck = torch.load(str(tmp_path / "w" / "best.pt"))
    assert set(ck) == {"epoch", "best_fitness", "training_results", "model", "optimizer"} and ck["epoch"] == 1
    assert any(k.endswith("Conv2d.weight") for k in ck["model"]) and any("BatchNorm2d.running_mean" in k for k in ck["model"])
    rows = (tmp_path / "results.txt").read_text().strip().split("\n")
    assert len(rows) == 2 and len(rows[0].split()) == 15     # epoch, mem, 4 losses, targets, img_size, 7 results


def test_lr_schedule_shape():
    sys.path.insert(0, ROOT)
    from train import lr_factor
    f = [lr_factor(e, 100, 10.0, 5) for e in range(100)]
    assert f[0] == 1.0 and abs(f[5] - 10.0) < 1e-12 and f[50] == 10.0 and abs(f[86] - 1.0) < 1e-9 and abs(f[96] - 0.1) < 1e-9
