"""Which small ATen launches (copies, fills, elementwise) sit inside the timed train step?  Builds the bench's bs-64 step,
switches the engine and the fused loss to eager launches (what the hipGraphs replay) and lists, for two steps under torch.profiler, every
aten:: op that launches a device kernel or copy, with its count per step and the innermost repo frame that issued it.
    python tools/step_small_ops.py [--bs 64]"""
import argparse
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rotate_yolov3_amd  # noqa: E402,F401
from bench import init_bench_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--size", type=int, default=608)
    a = ap.parse_args()
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.dist import GradientAllReducer
    from rotate_yolov3_amd.model.loss import compute_loss
    from rotate_yolov3_amd.model.models import Darknet
    from rotate_yolov3_amd.utils.synthetic import synthetic_targets
    from train import make_optimizer
    dev = torch.device("cuda:0")
    hyp = {"giou": 0.1, "cls": 27.76, "cls_pw": 1.0, "obj": 20.35, "obj_pw": 1.0, "iou_t": 0.5, "ang_t": 3.1415926 / 12,
           "reg": 1.0, "fl_gamma": 0.5, "context_factor": 1.0, "lr0": 1e-4, "momentum": 0.97, "weight_decay": 0.0004569, "riou": 1}
    torch.manual_seed(0)
    model = init_bench_weights(Darknet(make_cfg.darknet53(a.size, a.size), hyp), seed=0).to(dev).train()
    model.nc, model.arc, model.hyp = 1, "default", hyp
    model.enable_fused_loss(capacity=max(256, 8 * a.bs))
    opt = make_optimizer(model, hyp)
    dp = GradientAllReducer(model)
    x = torch.rand(a.bs, 3, a.size, a.size, device=dev)
    tg = synthetic_targets(a.bs, seed=1, device=dev)

    def step():
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            pred = model(x)
        loss, _ = compute_loss([p.float() for p in pred], tg.clone(), model, hyp)
        loss.backward()
        dp.finish()
        opt.step()
        dp.zero_grad()
    os.environ["RYOLO_NO_GRAPH"] = "1"
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    rows = collections.OrderedDict()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.device_time_total <= 0 or ev.cpu_children:
            continue
        frame = "?"
        for fr in (ev.stack or []):
            if ROOT in fr and "site-packages" not in fr:
                frame = fr.replace(ROOT + "/", "")
                break
        k = (ev.name, frame)
        r = rows.setdefault(k, [0, 0.0])
        r[0] += 1
        r[1] += ev.device_time_total
    print("%-28s %-90s %9s %10s" % ("op", "issued from", "per step", "us / step"))
    tot_n, tot_us = 0, 0.0
    for (name, frame), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        print("%-28s %-90s %9.1f %10.1f" % (name, frame[:90], n / 2.0, us / 2.0))
        tot_n += n
        tot_us += us
    print("total: %.1f launches and %.1f us of device time per step outside the library's own kernels" % (tot_n / 2.0, tot_us / 2.0))


if __name__ == "__main__":
    main()
