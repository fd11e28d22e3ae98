"""Time the detection post-processing (score filter + per-class rotated NMS) on a synthetic batched prediction tensor:
bs images x N rows x (6+nc), `cand` rows per image above the confidence threshold (32 x 2000).
    python tools/post_bench.py [--bs 32] [--rows 545832] [--cand 2000] [--reps 5]"""
import argparse
import math
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.utils.nms import nms as nms_mod  # noqa: E402


def synth(bs, rows, cand, nc, dev, seed=0):
    g = torch.Generator().manual_seed(seed)
    p = torch.empty(bs, rows, 6 + nc)
    p[..., 0:2] = torch.rand(bs, rows, 2, generator=g) * 608
    p[..., 2:4] = 8 * 16 ** torch.rand(bs, rows, 2, generator=g)
    p[..., 4] = (torch.rand(bs, rows, generator=g) - 0.5) * math.pi
    p[..., 5] = torch.rand(bs, rows, generator=g) * 0.2          # below the threshold ...
    for b in range(bs):                                           # ... except `cand` rows per image
        idx = torch.randperm(rows, generator=g)[:cand]
        p[b, idx, 5] = 0.5 + 0.5 * torch.rand(cand, generator=g)
    p[..., 6:] = 1.0 if nc == 1 else torch.rand(bs, rows, nc, generator=g)
    return p.to(dev)


if __name__ == "__main__" and "--engine" not in sys.argv:
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--rows", type=int, default=545832)
    ap.add_argument("--cand", type=int, default=2000)
    ap.add_argument("--nc", type=int, default=1)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    pred = synth(a.bs, a.rows, a.cand, a.nc, dev)
    fns = [("non_max_suppression (per image / class loop)", nms_mod.non_max_suppression)]
    if hasattr(nms_mod, "non_max_suppression_batched"):
        fns.append(("non_max_suppression_batched", nms_mod.non_max_suppression_batched))
    for name, fn in fns:
        outs = None
        for r in range(a.reps + 1):
            q = pred.clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = fn(q, 0.3, 0.5)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            if r == 0:
                continue
            print("%-48s %.2f ms  kept %d" % (name, dt * 1e3, sum(len(o) for o in outs if o is not None)), flush=True)


def engine_bench(bs=32, size=608, cand=2000, reps=5):
    """End-to-end detection step on random-init Darknet-53: forward + non_max_suppression_batched(io) vs HipEngine.detect
    (fused decode + filter, no io), threshold chosen so that ~`cand` rows per image survive the filter."""
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.model.engine import HipEngine
    from rotate_yolov3_amd.model.models import Darknet
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    from bench import init_bench_weights                      # finite activations through 75 layers
    model = init_bench_weights(Darknet(make_cfg.darknet53(width=size, height=size), {"context_factor": 1.0}).eval(), seed=0).to(dev)
    x = torch.rand(bs, 3, size, size, generator=torch.Generator().manual_seed(0)).to(dev)
    eng = HipEngine(model, x.shape, dev)
    with torch.no_grad():
        io, _ = eng(x)
        sc = io[..., 5].flatten()
        thr = float(sc[::97].kthvalue(int(sc[::97].numel() * (1.0 - cand / io.shape[1]))).values)
        for name, fn in [("forward + non_max_suppression_batched", lambda: nms_mod.non_max_suppression_batched(eng(x)[0], thr, 0.5)),
                         ("HipEngine.detect (fused decode+filter)", lambda: eng.detect(x, thr, 0.5))]:
            for r in range(reps + 1):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out = fn()
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if r:
                    print("%-44s %.2f ms/batch = %.0f img/s  kept %d" % (name, dt * 1e3, bs / dt, sum(len(o) for o in out if o is not None)),
                          flush=True)


if __name__ == "__main__" and "--engine" in sys.argv:
    engine_bench()
