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
    # device-side view: every kernel / copy of the two steps that is NOT one of the library's own kernels
    ours = ("ryolo", "conv_", "wgrad", "bn_act", "bn_finalize", "yolo_", "sgd_batch", "pack_batch", "build_targets", "nchw_f32", "upsample", "add_nhwc",
            "riou", "dgrad3x3", "conv0_", "conv3x3", "pgrad")
    dev_rows = {}
    for ev in prof.events():
        if ev.device_type is None or str(ev.device_type).endswith("CPU"):
            continue
        if any(t in ev.name for t in ours):
            continue
        r = dev_rows.setdefault(ev.name[:110], [0, 0.0])
        r[0] += 1
        r[1] += ev.device_time_total if ev.device_time_total else (ev.time_range.end - ev.time_range.start)
    print("%-112s %9s %10s" % ("device kernel / copy (not the library's)", "per step", "us / step"))
    tn, tu = 0, 0.0
    for name, (n, us) in sorted(dev_rows.items(), key=lambda kv: -kv[1][1]):
        print("%-112s %9.1f %10.1f" % (name, n / 2.0, us / 2.0))
        tn += n
        tu += us
    print("total: %.1f launches, %.1f us per step" % (tn / 2.0, tu / 2.0))
    # host-side view: the aten ops that issue them, with the innermost repo frame
    rows = collections.OrderedDict()
    for ev in prof.events():
        if not ev.name.startswith("aten::") or ev.name in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::detach", "aten::alias",
                                                            "aten::slice", "aten::select", "aten::empty_strided", "aten::to", "aten::_to_copy",
                                                            "aten::item", "aten::_local_scalar_dense", "aten::lift_fresh", "aten::reshape", "aten::expand"):
            continue
        if any(c.name.startswith("aten::") for c in (ev.cpu_children or [])):
            continue                     # count leaf ops only
        frame = "?"
        for fr in (ev.stack or []):
            if ROOT in fr and "site-packages" not in fr and "tools/step_small_ops" not in fr:
                frame = fr.replace(ROOT + "/", "")
                break
        r = rows.setdefault((ev.name, frame), [0])
        r[0] += 1
    print()
    print("%-28s %-100s %9s" % ("leaf aten op", "issued from", "per step"))
    for (name, frame), (n,) in sorted(rows.items(), key=lambda kv: -kv[1][0]):
        print("%-28s %-100s %9.1f" % (name, frame[:100], n / 2.0))


if __name__ == "__main__":
    main()
