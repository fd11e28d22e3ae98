"""A/B timing of the conv tiles on the MFMA-bound Darknet-53 shapes (bs 32, 608^2 input): interleaved rounds in ONE process,
median and min per (shape, tile).  python tools/mp_tune.py [--tiles 1,8,11,9]   (1 = 128x128 igemm, 8 / 11 = conv_mp 256 / 192 rows, 9 = conv_mq) [--rounds 7] [--reps 10]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.model import hip_ops as ops  # noqa: E402

# (k, stride, cin, cout, Hout, count in the net)
SHAPES = [(3, 1, 128, 256, 76, 11), (3, 1, 256, 512, 38, 11), (3, 1, 512, 1024, 19, 7), (3, 2, 128, 256, 76, 1),
          (3, 2, 256, 512, 38, 1), (3, 2, 512, 1024, 19, 1), (1, 1, 512, 256, 38, 10), (1, 1, 1024, 512, 19, 7),
          (1, 1, 768, 256, 38, 1),
          # K sweep at a fixed tile count (fit T = rounds * (a + KT * b)): KT = 9, 18, 36, 72
          (3, 1, 64, 256, 76, 0), (3, 1, 256, 256, 76, 0), (3, 1, 512, 256, 76, 0)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tiles", default="1,8,11,9")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--shapes", default="")
    a = ap.parse_args()
    tiles = [int(t, 0) for t in a.tiles.split(",")]
    dev = torch.device("cuda:0")
    shapes = SHAPES if not a.shapes else [SHAPES[int(i)] for i in a.shapes.split(",")]
    for (k, s, cin, cout, ho, cnt) in shapes:
        hin = ho * s
        x = torch.randn(a.bs, hin, hin, cin, device=dev).clamp_(-3, 3).to(torch.bfloat16)
        x = torch.where(x > 0, x, x * 0.1)       # PReLU-shaped activations, like the real layers' inputs
        w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        packed = ops.pack_weights(w, cin_pad=cin)
        sc = torch.ones(ops.cpad(cout), device=dev)
        sh = torch.zeros(ops.cpad(cout), device=dev)
        out = torch.empty(a.bs, ho, ho, cout, device=dev, dtype=torch.bfloat16)
        flop = 2.0 * k * k * cin * cout * ho * ho * a.bs
        times = {t: [] for t in tiles}
        ok = {}
        for t in tiles:
            try:
                ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, out=out, tile=t)
                torch.cuda.synchronize()
                ok[t] = True
            except RuntimeError as e:
                ok[t] = False
        for r in range(a.rounds):
            for t in tiles:
                if not ok[t]:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, out=out, tile=t)
                e1.record()
                torch.cuda.synchronize()
                times[t].append(e0.elapsed_time(e1) / a.reps)
        line = "k%d s%d %4d->%4d @%2d x%2d |" % (k, s, cin, cout, ho, cnt)
        for t in tiles:
            if not ok[t]:
                line += "  t%-2d   n/a          |" % t
                continue
            v = sorted(times[t])
            med, mn = v[len(v) // 2], v[0]
            line += "  t%-2d %6.1f us %6.0f TF (min %6.1f) |" % (t, med * 1e3, flop / med / 1e9, mn * 1e3)
        print(line, flush=True)


if __name__ == "__main__":
    main()
