"""Per-shape timing of the HIP conv block on the 23 distinct Darknet-53 conv shapes at 608x608 (SURVEY.md
Appendix A), plus the rotated-NMS pipeline at n boxes.  Run on the GPU box:
    python tools/layer_bench.py [--bs 32] [--reps 5] [--nms 50000] [--tiles]
Prints one line per shape: ms, TFLOP/s, algorithmic GB/s, and the weighted forward total."""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.model import hip_ops as ops  # noqa: E402

# (k, stride, cin, cout, Hout, count)
SHAPES = [(3, 1, 3, 32, 608, 1), (1, 1, 64, 32, 304, 1), (3, 1, 32, 64, 304, 1), (3, 2, 32, 64, 304, 1),
          (1, 1, 128, 64, 152, 2), (3, 1, 64, 128, 152, 2), (3, 2, 64, 128, 152, 1),
          (1, 1, 256, 128, 76, 10), (1, 1, 256, 504, 76, 1), (1, 1, 384, 128, 76, 1), (3, 1, 128, 256, 76, 11),
          (3, 2, 128, 256, 76, 1), (1, 1, 256, 128, 38, 1), (1, 1, 512, 256, 38, 10), (1, 1, 512, 504, 38, 1),
          (1, 1, 768, 256, 38, 1), (3, 1, 256, 512, 38, 11), (3, 2, 256, 512, 38, 1), (1, 1, 512, 256, 19, 1),
          (1, 1, 1024, 504, 19, 1), (1, 1, 1024, 512, 19, 7), (3, 1, 512, 1024, 19, 7), (3, 2, 512, 1024, 19, 1)]


def bench_conv(bs, reps, tiles, flags=0):
    dev = torch.device("cuda:0")
    total_ms, total_flop = 0.0, 0.0
    for (k, s, cin, cout, ho, cnt) in SHAPES:
        hin = ho * s
        cin_k = 8 if cin == 3 else cin
        x = torch.randn(bs, hin, hin, cin_k, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        packed = ops.pack_weights(w, cin_pad=cin_k)
        sc = torch.ones(ops.cpad(cout), device=dev)
        sh = torch.zeros(ops.cpad(cout), device=dev)
        out = torch.empty(bs, ho, ho, cout, device=dev, dtype=torch.bfloat16)
        flop = 2.0 * k * k * cin * cout * ho * ho * bs
        byts = 2.0 * bs * (hin * hin * cin_k + ho * ho * cout) + 2.0 * k * k * cin_k * cout
        best = {}
        for tile in ([1, 2, 3, 4] if tiles else [flags]):
            for _ in range(2):
                ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, out=out, tile=tile)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, out=out, tile=tile)
            e1.record()
            torch.cuda.synchronize()
            best[tile] = e0.elapsed_time(e1) / reps
        tile, ms = min(best.items(), key=lambda kv: kv[1])
        total_ms += ms * cnt
        total_flop += flop * cnt
        print("k%d s%d %4d->%4d @%3d x%2d  %8.3f ms  %7.1f TF/s  %7.1f GB/s  tile=%d %s" % (
            k, s, cin, cout, ho, cnt, ms, flop / ms / 1e9, byts / ms / 1e6, tile,
            " ".join("%d:%.3f" % kv for kv in sorted(best.items())) if tiles else ""), flush=True)
        del x, w, out
    print("conv total (bs=%d): %.3f ms  %.1f TF/s  -> %.0f img/s (convs only)" % (
        bs, total_ms, total_flop / total_ms / 1e9, bs / total_ms * 1e3), flush=True)


def bench_wgrad(bs, reps, sweep=False):
    """weight-gradient GEMM per shape (wgrad kernel + split-K reduce), same shape list."""
    from rotate_yolov3_amd.model import hip_train_ops as tops
    dev = torch.device("cuda:0")
    total_ms, total_flop = 0.0, 0.0
    for (k, s, cin, cout, ho, cnt) in SHAPES:
        hin = ho * s
        cin_k = 8 if cin == 3 else cin
        x = torch.randn(bs, hin, hin, cin_k, device=dev).to(torch.bfloat16)
        dz = torch.randn(bs, ho, ho, cout, device=dev).to(torch.bfloat16)
        g = torch.zeros(cout, cin, k, k, device=dev)
        d = tops.make_desc(x, cout, k, s, k // 2)
        ws = torch.empty(tops.wgrad_ws_bytes(d), dtype=torch.uint8, device=dev)

        def run(dd, wsb):
            for _ in range(2):
                tops.conv_wgrad(dd, x, dz, cin, g, False, wsb)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                tops.conv_wgrad(dd, x, dz, cin, g, False, wsb)
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps

        ms = run(d, ws)
        if sweep:
            T = 128 if min(cin_k, cout) >= 128 else (64 if min(cin_k, cout) >= 64 else 32)
            base = -(-cout // T) * -(-cin_k // T) * k * k
            res = []
            for blocks in (256, 384, 512, 768, 1024, 1536, 2048, 3072):
                S = max(1, blocks // base)
                if any(r[0] == S for r in res):
                    continue
                dd = tops.make_desc(x, cout, k, s, k // 2, tile=S << 16)
                nb = tops.wgrad_ws_bytes(dd)
                if nb > (3 << 30):
                    continue
                w2 = torch.empty(nb, dtype=torch.uint8, device=dev)
                res.append((S, S * base, run(dd, w2)))
                del w2
            print("   sweep base=%d: %s" % (base, "  ".join("S%d(%d)=%.3f" % r for r in res)), flush=True)
        flop = 2.0 * k * k * cin * cout * ho * ho * bs
        total_ms += ms * cnt
        total_flop += flop * cnt
        print("wgrad k%d s%d %4d->%4d @%3d x%2d  %8.3f ms  %7.1f TF/s  ws %.0f MB" % (
            k, s, cin, cout, ho, cnt, ms, flop / ms / 1e9, ws.numel() / 2**20), flush=True)
        del x, dz, ws
    print("wgrad total (bs=%d): %.3f ms  %.1f TF/s" % (bs, total_ms, total_flop / total_ms / 1e9), flush=True)


def bench_dgrad(bs, reps):
    """data-gradient conv per shape (every stride-2 parity class included), same shape list; the stem (C_in = 3) has none."""
    from rotate_yolov3_amd.model import hip_train_ops as tops
    dev = torch.device("cuda:0")
    total_ms, total_flop = 0.0, 0.0
    for (k, s, cin, cout, ho, cnt) in SHAPES:
        if cin == 3:
            continue
        hin = ho * s
        x = torch.randn(bs, hin, hin, cin, device=dev).to(torch.bfloat16)
        dz = torch.randn(bs, ho, ho, cout, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        d = tops.make_desc(x, cout, k, s, k // 2)
        pk = tops.pack_weights_dgrad(w, s)
        ones = torch.ones(ops.cpad(cin), device=dev)
        zeros = torch.zeros(ops.cpad(cin), device=dev)
        dx = torch.empty_like(x)
        for _ in range(2):
            tops.conv_dgrad(d, dz, pk, ones, zeros, dx, False)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            tops.conv_dgrad(d, dz, pk, ones, zeros, dx, False)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flop = 2.0 * k * k * cin * cout * ho * ho * bs
        byts = 2.0 * bs * (hin * hin * cin + ho * ho * cout)
        total_ms += ms * cnt
        total_flop += flop * cnt
        print("dgrad k%d s%d %4d->%4d @%3d x%2d  %8.3f ms  %7.1f TF/s  %7.1f GB/s" % (k, s, cin, cout, ho, cnt, ms, flop / ms / 1e9,
                                                                                   byts / ms / 1e6), flush=True)
        del x, dz, dx
    print("dgrad total (bs=%d): %.3f ms  %.1f TF/s" % (bs, total_ms, total_flop / total_ms / 1e9), flush=True)


def bench_nms(n, reps):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from rotate_yolov3_amd.utils.nms.r_nms import r_nms
    rng = np.random.default_rng(0)
    d = np.empty((n, 6), np.float32)
    d[:, 0] = rng.uniform(0, 608, n); d[:, 1] = rng.uniform(0, 608, n)
    d[:, 2] = 8 * 16 ** rng.uniform(0, 1, n); d[:, 3] = 8 * 16 ** rng.uniform(0, 1, n)
    d[:, 4] = rng.uniform(-np.pi / 2, np.pi / 2, n); d[:, 5] = (rng.permutation(n) + 0.5) / n
    dt = torch.from_numpy(d).cuda()
    for _ in range(2):
        k = r_nms(dt, 0.5)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        k = r_nms(dt, 0.5)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    pairs = n * (n - 1) / 2
    print("rnms n=%d kept=%d  %.3f ms  %.3e pairs/s" % (n, k.numel(), ms, pairs / ms * 1e3), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--nms", type=int, default=50000)
    ap.add_argument("--tiles", action="store_true")
    ap.add_argument("--skip-conv", action="store_true")
    ap.add_argument("--wgrad", action="store_true")
    ap.add_argument("--dgrad", action="store_true")
    ap.add_argument("--no-persist", action="store_true", help="one tile per workgroup (tile bit 0x200)")
    ap.add_argument("--sweep", action="store_true")
    a = ap.parse_args()
    if a.nms:
        bench_nms(a.nms, a.reps)
        bench_nms(2000, a.reps)
    if a.wgrad:
        bench_wgrad(a.bs, a.reps, a.sweep)
    if a.dgrad:
        bench_dgrad(a.bs, a.reps)
    if not a.skip_conv:
        bench_conv(a.bs, a.reps, a.tiles, 0x200 if a.no_persist else 0)
