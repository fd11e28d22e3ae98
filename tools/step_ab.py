"""A/B timing of the bs-64 training step under two library settings in ONE process on ONE box (boxes of the pool differ by +-4 %,
more than most single changes): two model instances, each captured into its own hipGraphs while its setting is active, then
timed in interleaved rounds.  Settings are debug knobs of the ABLATION build.   python tools/step_ab.py [--bs 64] [--rounds 5]"""
import argparse
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get("RYOLO_AB_PRODUCT_LIB", "1") != "1":      # the ablation build's debug knobs are only needed by settings that call them
    os.environ.setdefault("RYOLO_HIP_LIB", os.path.join(ROOT, "rotate-yolov3_amd", "libryolo_hip_ablation.so"))
import torch  # noqa: E402

import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd import _lib  # noqa: E402
from bench import init_bench_weights  # noqa: E402

L = _lib.lib()
if hasattr(L, "ryolo_debug_bn_set"):
    L.ryolo_debug_bn_set.argtypes = [C.c_int] * 5
    L.ryolo_debug_bn_set.restype = None
    L.ryolo_debug_conv_nt_min.argtypes = [C.c_longlong]
    L.ryolo_debug_conv_nt_min.restype = None

MB = 1 << 20
# engine-level A/B: the setting is an environment variable read when the engine is built; library-level knobs of the
# ablation build (ryolo_debug_conv_nt_min(bytes), ryolo_debug_bn_set(...)) can be called from a setting the same way
SETTINGS = {
    "bn_reduce_separate": lambda: os.environ.__setitem__("RYOLO_BN_REDUCE_FUSION", "0"),
    "bn_reduce_in_dgrad": lambda: os.environ.__setitem__("RYOLO_BN_REDUCE_FUSION", "1"),
}


def settings_from_args(specs):
    """--ab name=VAR:VALUE[,VAR:VALUE...] (repeatable): a setting is a set of environment variables the library / engine reads when a
    launch is made or captured, e.g.  --ab new=RYOLO_WGRAD_REDUCE4:1 --ab old=RYOLO_WGRAD_REDUCE4:0"""
    out = {}
    for spec in specs:
        name, rest = spec.split("=", 1)
        pairs = [kv.split(":", 1) for kv in rest.split(",") if kv]

        def setter(pairs=pairs):
            for k, v in pairs:
                os.environ[k] = v                      # engine-level switches (read when an engine is built) and the measurement build's knobs
                if k in _lib.TUNING_SWITCHES:          # the library's own six are read from the environment once: set them in-process
                    _lib.set_tuning(k, v)
                if k == "RYOLO_DEBUG_WGRAD":           # measurement build: weight-gradient dispatch variants (csrc/train.hip: ryolo_debug_wgrad_set)
                    L.ryolo_debug_wgrad_set.argtypes = [C.c_int]
                    L.ryolo_debug_wgrad_set.restype = None
                    L.ryolo_debug_wgrad_set(int(v))
        out[name] = setter
    return out


def make(args, dev, setting):
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.model.loss import compute_loss
    from rotate_yolov3_amd.model.models import Darknet
    from rotate_yolov3_amd.utils.synthetic import synthetic_targets
    from train import make_optimizer
    hyp = {"giou": 0.1, "cls": 27.76, "cls_pw": 1.0, "obj": 20.35, "obj_pw": 1.0, "iou_t": 0.5, "ang_t": 3.1415926 / 12,
           "reg": 1.0, "fl_gamma": 0.5, "context_factor": 1.0, "lr0": 1e-4, "momentum": 0.97, "weight_decay": 0.0004569, "riou": 1}
    torch.manual_seed(0)
    SETTINGS[setting]()
    model = init_bench_weights(Darknet(make_cfg.darknet53(args.size, args.size), hyp), seed=0).to(dev).train()
    model.nc, model.arc, model.hyp = 1, "default", hyp
    model.enable_fused_loss(capacity=max(256, 8 * args.bs))
    opt = make_optimizer(model, hyp)
    from rotate_yolov3_amd.dist import GradientAllReducer
    dp = GradientAllReducer(model)            # as bench.py / train.py: param.grad lives in the flat buckets
    x = torch.rand(args.bs, 3, args.size, args.size, device=dev)
    tg = synthetic_targets(args.bs, seed=1, device=dev)

    def step():
        SETTINGS[setting]()
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            pred = model(x)
        loss, _ = compute_loss([p.float() for p in pred], tg.clone(), model, hyp)
        loss.backward()
        dp.finish()
        opt.step()
        dp.zero_grad()
    for _ in range(4):          # eager, eager, capture, replay
        step()
    torch.cuda.synchronize(dev)
    return step


def forward_ab(a, dev):
    """eval forward (bs 32, eager launches: the setting takes effect at every launch)"""
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.model.engine import HipEngine
    from rotate_yolov3_amd.model.models import Darknet
    torch.manual_seed(0)
    model = init_bench_weights(Darknet(make_cfg.darknet53(a.size, a.size), {"context_factor": 1.0}).eval(), seed=0).to(dev)
    x = torch.rand(32, 3, a.size, a.size, device=dev)
    engs = {}
    for name, setf in SETTINGS.items():      # one engine per setting: some settings change the plan (fused launches), not only the dispatch
        setf()
        engs[name] = HipEngine(model, x.shape, dev)
    times = {name: [] for name in SETTINGS}
    with torch.no_grad():
        for name, setf in SETTINGS.items():
            setf()
            for _ in range(3):
                engs[name](x)
        for _ in range(a.rounds):
            for name, setf in SETTINGS.items():
                setf()
                eng = engs[name]
                eng(x)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(10):
                    eng(x)
                torch.cuda.synchronize(dev)
                times[name].append((time.perf_counter() - t0) / 10 * 1e3)
    for name, v in times.items():
        v = sorted(v)
        print("forward bs32  %-20s median %.3f ms  min %.3f ms" % (name, v[len(v) // 2], v[0]))
    del eng, engs, model
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--ab", action="append", default=[], help="name=VAR:VALUE[,VAR:VALUE...]; replaces the built-in pair of settings")
    ap.add_argument("--forward", action="store_true", help="also the bs-32 eval forward under each setting")
    ap.add_argument("--forward-only", action="store_true", help="skip the training step")
    a = ap.parse_args()
    if a.ab:
        SETTINGS.clear()
        SETTINGS.update(settings_from_args(a.ab))
    dev = torch.device("cuda:0")
    if a.forward:
        forward_ab(a, dev)
    if a.forward_only:
        return
    steps = {name: make(a, dev, name) for name in SETTINGS}
    times = {name: [] for name in SETTINGS}
    for _ in range(a.rounds):
        for name, st in steps.items():
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(a.steps):
                st()
            torch.cuda.synchronize(dev)
            times[name].append((time.perf_counter() - t0) / a.steps * 1e3)
    for name, v in times.items():
        v = sorted(v)
        print("%-24s median %.2f ms  min %.2f ms  (%s)" % (name, v[len(v) // 2], v[0], " ".join("%.2f" % q for q in v)))


if __name__ == "__main__":
    main()
