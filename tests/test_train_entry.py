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
    assert os.path.isfile(str(tmp_path / "w" / "last.pt"))
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


def test_resume_does_not_compound_the_lr_factor():
    """optimizer.state_dict() stores the lr already scaled by the epoch's factor; a resumed run must scale lr0
    again, not that value (with multiplier 10 a resume after warm-up would otherwise train at 100 x lr0)."""
    sys.path.insert(0, ROOT)
    from train import init_schedule, lr_factor, set_epoch_lr
    hyp = {"lr0": 1e-3, "multiplier": 10.0, "warm_epoch": 1, "momentum": 0.9}
    epochs = 20
    w = torch.nn.Parameter(torch.zeros(3))
    opt = torch.optim.SGD([w], lr=hyp["lr0"], momentum=0.9)
    opt.add_param_group({"params": [torch.nn.Parameter(torch.zeros(2))], "weight_decay": 1e-4})
    init_schedule(opt, hyp)
    for k in (0, 1, 2, 17):                  # warm-up, plateau, after the first milestone
        set_epoch_lr(opt, hyp, k, epochs)
        sd = opt.state_dict()                 # what the checkpoint of epoch k carries
        opt2 = torch.optim.SGD([torch.nn.Parameter(torch.zeros(3))], lr=hyp["lr0"], momentum=0.9)
        opt2.add_param_group({"params": [torch.nn.Parameter(torch.zeros(2))], "weight_decay": 1e-4})
        opt2.load_state_dict(sd)
        init_schedule(opt2, hyp)              # train(): after load_state_dict
        set_epoch_lr(opt2, hyp, k + 1, epochs)
        want = hyp["lr0"] * lr_factor(k + 1, epochs, 10.0, 1)
        assert all(abs(g["lr"] - want) < 1e-15 for g in opt2.param_groups), (k, [g["lr"] for g in opt2.param_groups], want)


def test_resume_without_a_checkpoint_fails_instead_of_restarting(tmp_path):
    """`--resume` with a mistyped --wdir used to start at epoch 0 and overwrite last.pt / best.pt / results.txt; the
    reference fails in torch.load (train.py:406, :104).  A missing plain --weights file only prints a NOTE (like test.py / detect.py)."""
    from tests.test_dp_gloo import CFG
    cfg = tmp_path / "m.cfg"
    cfg.write_text(CFG)
    hyp = tmp_path / "hyp.py"
    hyp.write_text("giou: 0.1\ncls: 27.76\ncls_pw: 1.446\nobj: 20.35\nobj_pw: 3.941\niou_t: 0.3\nang_t: 3.1415926/12\n"
                   "reg: 1.0\nfl_gamma: 0.5\ncontext_factor: 1.0\nlr0: 0.0001\nmultiplier:10\nwarm_epoch:1\nmomentum: 0.97\n"
                   "weight_decay: 0.0004569\nepochs: 1\nbatch_size: 2\nsave_interval: 300\ntest_interval: 5\n")
    base = [sys.executable, os.path.join(ROOT, "train.py"), "--cfg", str(cfg), "--hyp", str(hyp), "--img-size", "64", "--synthetic", "2",
            "--device", "cpu", "--wdir", str(tmp_path / "nowhere")]
    r = subprocess.run(base + ["--resume"], cwd=str(tmp_path), capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "no checkpoint at" in r.stderr, r.stderr[-1500:]
    assert not os.path.exists(str(tmp_path / "nowhere" / "last.pt")) and not os.path.exists(str(tmp_path / "results.txt"))
    r = subprocess.run(base + ["--weights", str(tmp_path / "missing.weights")], cwd=str(tmp_path), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0 and "NOTE: weights file" in r.stdout, (r.stdout[-800:], r.stderr[-800:])
