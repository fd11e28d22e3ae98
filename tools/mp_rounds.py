"""Per-tile cost of the persistent conv tile: 3x3 Cin->256 on N x 64 x 64 inputs, N = 16 / 32 / 48 / 64 = exactly 1 / 2 / 3 / 4
tiles per workgroup on 256 CUs.  python tools/mp_rounds.py [--cin 128] [--tiles 1,8,11,9]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.model import hip_ops as ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cin", default="128")
ap.add_argument("--tiles", default="1,8,11,9")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--res", type=int, default=0)
a = ap.parse_args()
dev = torch.device("cuda:0")
for cin in [int(c) for c in a.cin.split(",")]:
    for n in (16, 32, 48, 64):
        x = torch.randn(n, 64, 64, cin, device=dev).clamp_(-3, 3).to(torch.bfloat16)
        w = torch.randn(256, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
        packed = ops.pack_weights(w, cin_pad=cin)
        sc = torch.ones(256, device=dev)
        sh = torch.zeros(256, device=dev)
        out = torch.empty(n, 64, 64, 256, device=dev, dtype=torch.bfloat16)
        res = torch.randn(n, 64, 64, 256, device=dev).to(torch.bfloat16) if a.res else None
        line = "cin %4d KT %2d  N %2d (%d tiles/WG) |" % (cin, 9 * cin // 64, n, n // 16)
        for t in [int(v) for v in a.tiles.split(",")]:
            ts = []
            for r in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    ops.conv2d_bn_act(x, packed, sc, sh, 256, 3, act=1, out=out, tile=t, residual=res)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / a.reps)
            ts.sort()
            line += "  t%-2d %6.1f us |" % (t, ts[2] * 1e3)
        print(line, flush=True)
