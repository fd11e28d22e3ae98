"""The 1x1 layers of Darknet-53 at 608^2: conv_pw.hip (auto dispatch) against the 128 x 128 implicit-GEMM tile it replaces, same
process, back-to-back launches.  python tools/pw_bench.py [--bs 32] [--reps 30] [--train]
Forward (inference epilogue), and with --train the bs-64 training forward with statistics and the 1x1 data gradients."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd import _lib  # noqa: E402
from rotate_yolov3_amd.model import hip_ops as ops  # noqa: E402
from rotate_yolov3_amd.model import hip_train_ops as tr  # noqa: E402

# (cin, cout, H, launches per forward)
FWD = [(64, 32, 304, 1), (128, 64, 152, 2), (256, 128, 76, 10), (384, 128, 76, 1), (256, 504, 76, 1), (256, 128, 38, 1), (512, 256, 38, 10),
       (768, 256, 38, 1), (512, 504, 38, 1), (512, 256, 19, 1), (1024, 512, 19, 7), (1024, 504, 19, 1)]


def timeit(fn, reps=30):
    """us per launch: `reps` launches replayed from one hipGraph (the Python / ctypes call costs ~10 us, more than the short layers
    take), 5 replays"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--train", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    tot = {"pw": 0.0, "igemm": 0.0}
    print("forward bs %d" % a.bs)
    for (cin, cout, hw, cnt) in FWD:
        x = torch.randn(a.bs, hw, hw, cin, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
        pk = ops.pack_weights(w, cin_pad=cin)
        sc, sh = torch.ones(ops.cpad(cout), device=dev), torch.zeros(ops.cpad(cout), device=dev)
        out = torch.empty(a.bs, hw, hw, cout, device=dev, dtype=torch.bfloat16)
        name = ops.conv_kernel_name(a.bs, hw, hw, cin, cout, 1)
        t_auto = timeit(lambda: ops.conv2d_bn_act(x, pk, sc, sh, cout, 1, act=1, out=out, tile=0), a.reps)
        old_tile = 3 if cout <= 32 else (2 if cout <= 64 else 1)
        t_old = timeit(lambda: ops.conv2d_bn_act(x, pk, sc, sh, cout, 1, act=1, out=out, tile=old_tile), a.reps)
        flop = 2.0 * cin * cout * hw * hw * a.bs
        byts = 2.0 * a.bs * hw * hw * (cin + cout) + 2.0 * cin * cout
        tot["pw"] += t_auto * cnt
        tot["igemm"] += t_old * cnt
        print("%5d->%4d @%3d x%2d  %-18s %7.1f us %6.0f TF/s %5.2f TB/s | igemm %7.1f us  (%.2fx)" % (
            cin, cout, hw, cnt, name, t_auto, flop / t_auto / 1e6, byts / t_auto / 1e6, t_old, t_old / t_auto), flush=True)
        del x, out
    print("sum 1x1 forward: auto %.3f ms, igemm %.3f ms" % (tot["pw"] / 1e3, tot["igemm"] / 1e3), flush=True)
    if not a.train:
        return
    bs = 64
    print("training forward with statistics, bs %d" % bs)
    for (cin, cout, hw, cnt) in FWD:
        if cout == 504:
            continue
        x = torch.randn(bs, hw, hw, cin, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
        pk = ops.pack_weights(w, cin_pad=cin)
        ones, zeros = torch.ones(ops.cpad(cout), device=dev), torch.zeros(ops.cpad(cout), device=dev)
        z = torch.empty(bs, hw, hw, cout, device=dev, dtype=torch.bfloat16)
        res = []
        for tile in (0, 3 if cout <= 32 else (2 if cout <= 64 else 1)):
            d = tr.make_desc(x, cout, 1, 1, 0, tile=tile)
            part = torch.zeros(tr.stat_rows() * 2 * ops.cpad(cout), dtype=torch.float64, device=dev)
            res.append(timeit(lambda: tr.conv_fwd_stats(d, x, pk, ones, zeros, z, part=part, clear=False), a.reps))
        print("%5d->%4d @%3d x%2d  stats  auto %7.1f us | igemm %7.1f us (%.2fx)" % (cin, cout, hw, cnt, res[0], res[1], res[1] / res[0]), flush=True)
        del x, z
    print("1x1 data gradients (accumulating), bs %d" % bs)
    for (cin, cout, hw, cnt) in FWD:
        if cout == 504 or cin in (384, 768):
            continue
        x = torch.empty(bs, hw, hw, cin, device=dev, dtype=torch.bfloat16)
        w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
        pkd = tr.pack_weights_dgrad(w, 1)
        ones, zeros = torch.ones(ops.cpad(cin), device=dev), torch.zeros(ops.cpad(cin), device=dev)
        dz = torch.randn(bs, hw, hw, cout, device=dev).to(torch.bfloat16)
        dx = torch.zeros(bs, hw, hw, cin, device=dev, dtype=torch.bfloat16)
        res = []
        for tile in (0, 1):
            d = tr.make_desc(x, cout, 1, 1, 0, tile=tile)
            res.append(timeit(lambda: tr.conv_dgrad(d, dz, pkd, ones, zeros, dx, True), a.reps))
        # with the folded BatchNorm reduce
        z = torch.randn(bs, hw, hw, cin, device=dev).to(torch.bfloat16)
        stats = tuple(torch.rand(cin, device=dev) + 0.5 for _ in range(4))
        slope = torch.tensor([0.1], device=dev)
        fold = []
        for env in ("", "igemm"):
            _lib.set_tuning("RYOLO_CONV1X1", env or None)
            d = tr.make_desc(x, cout, 1, 1, 0)
            rows = tr.dgrad_bnreduce_rows(d)
            if rows <= 0:
                fold.append(float("nan"))
                continue
            part = torch.empty(rows, 3, cin, device=dev)
            fold.append(timeit(lambda: tr.conv_dgrad_bnreduce(d, dz, pkd, ones, zeros, dx, True, z, stats, slope, part), a.reps))
        _lib.set_tuning("RYOLO_CONV1X1", None)
        print("%5d<-%4d @%3d x%2d  dgrad  auto %7.1f us | igemm %7.1f us (%.2fx) | +bn-reduce auto %7.1f us igemm %7.1f us" % (
            cin, cout, hw, cnt, res[0], res[1], res[1] / res[0], fold[0], fold[1]), flush=True)
        del x, dz, dx, z


if __name__ == "__main__":
    main()
