"""conv_mq.hip's 128-channel tiles (round 5) against the kernels they replace, same process, graph-replayed back-to-back launches
(tools/pw_bench.py's timing): the 3x3 layers / data gradients with 128 output channels and the short-tile-list 1x1 layers of
Darknet-53 at 608^2.    python tools/mq128_bench.py [--reps 20]
Inference forward at bs 32 per forced tile; training forward with statistics, data gradient and data gradient with the folded
BatchNorm reduce at bs 64 per RYOLO_MQ128 setting (0 = round 4's kernels, 1 = 3x3 on the family, 2 = also the 1x1 layers)."""
import argparse
import ctypes
import os
import sys

import torch

# (round 6) the 128-channel family is instantiated in the measurement build only: run on the ablation library
os.environ.setdefault("RYOLO_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rotate-yolov3_amd", "libryolo_hip_ablation.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd import _lib  # noqa: E402
from rotate_yolov3_amd.model import hip_ops as ops  # noqa: E402
from rotate_yolov3_amd.model import hip_train_ops as tr  # noqa: E402
from pw_bench import timeit  # noqa: E402

TILE_NAMES = {0: "auto", 1: "igemm128x128", 7: "igemm128x128(2x2)", 0x807: "igemm(2x2) persist", 13: "conv_pw", 15: "mq 128x128", 16: "mq 64x128", 9: "mq 128x256"}
# (k, stride, cin, cout, Hout, residual, tiles to force)
FWD = [(3, 1, 64, 128, 152, True, [0x807, 1, 15, 16]), (3, 2, 64, 128, 152, False, [0x807, 1, 15, 16]),
       (3, 1, 256, 128, 76, True, [1, 15, 16]),
       (1, 1, 512, 256, 38, False, [13, 1, 7, 15, 16]), (1, 1, 768, 256, 38, False, [13, 1, 15, 16]),
       (1, 1, 1024, 512, 19, False, [1, 7, 15, 16]), (1, 1, 512, 256, 19, False, [1, 15, 16]), (1, 1, 256, 128, 38, False, [13, 1, 15, 16])]
# training shapes: (k, stride, cin, cout, Hout)
TRAIN = [(3, 1, 64, 128, 152), (3, 2, 64, 128, 152), (3, 1, 128, 256, 76), (3, 2, 128, 256, 76), (1, 1, 512, 256, 38), (1, 1, 1024, 512, 19),
         (1, 1, 256, 512, 38), (1, 1, 512, 1024, 19)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--train-bs", type=int, default=64)
    ap.add_argument("--no-train", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L = _lib.lib()
    print("# inference forward, bs %d: us per launch (TFLOP/s | algorithmic GB/s)" % a.bs)
    for (k, s, cin, cout, ho, res, tiles) in FWD:
        hin = ho * s
        x = torch.randn(a.bs, hin, hin, cin, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        packed = ops.pack_weights(w, cin_pad=cin)
        sc, sh = torch.ones(ops.cpad(cout), device=dev), torch.zeros(ops.cpad(cout), device=dev)
        out = torch.empty(a.bs, ho, ho, cout, device=dev, dtype=torch.bfloat16)
        r = torch.randn(a.bs, ho, ho, cout, device=dev).to(torch.bfloat16) if res else None
        flop = 2.0 * k * k * cin * cout * ho * ho * a.bs
        byts = 2.0 * a.bs * (hin * hin * cin + ho * ho * cout * (2 if res else 1)) + 2.0 * k * k * cin * cout
        cols = []
        for tile in [0] + tiles:
            try:
                us = timeit(lambda: ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, residual=r, out=out, tile=tile), a.reps)
                cols.append("%s %.1f (%.0f|%.0f)" % (TILE_NAMES.get(tile, hex(tile)), us, flop / us / 1e6, byts / us / 1e3))
            except Exception as e:      # noqa: BLE001
                cols.append("%s n/a" % TILE_NAMES.get(tile, hex(tile)))
        print("k%d s%d %4d->%-4d @%-3d%s  auto=%s | %s" % (k, s, cin, cout, ho, " +res" if res else "     ",
                                                          ops.conv_kernel_name(a.bs, hin, hin, cin, cout, k, s, residual=res), "  ".join(cols)), flush=True)
    print("# the same launches with 2 / 3 / 4 workgroups per CU (RYOLO_MQ128_WGPC; 163 / 112 registers and 50 / 34 KiB of LDS admit 3 / 4)")
    for (k, s, cin, cout, ho, res, tiles) in FWD[:4]:
        hin = ho * s
        x = torch.randn(a.bs, hin, hin, cin, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
        packed = ops.pack_weights(w, cin_pad=cin)
        sc, sh = torch.ones(ops.cpad(cout), device=dev), torch.zeros(ops.cpad(cout), device=dev)
        out = torch.empty(a.bs, ho, ho, cout, device=dev, dtype=torch.bfloat16)
        r = torch.randn(a.bs, ho, ho, cout, device=dev).to(torch.bfloat16) if res else None
        cols = []
        for tile in (15, 16):
            for wgpc in ("2", "3", "4"):
                os.environ["RYOLO_MQ128_WGPC"] = wgpc
                us = timeit(lambda: ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, residual=r, out=out, tile=tile), a.reps)
                cols.append("%s x%s %.1f" % (TILE_NAMES[tile], wgpc, us))
        os.environ.pop("RYOLO_MQ128_WGPC", None)
        print("k%d s%d %4d->%-4d @%-3d%s | %s" % (k, s, cin, cout, ho, " +res" if res else "     ", "  ".join(cols)), flush=True)
    if a.no_train:
        return
    bs = a.train_bs
    print("# training, bs %d: us per launch under RYOLO_MQ128 = 0 / 1 / 2 (kernel name)" % bs)
    for (k, s, cin, cout, ho) in TRAIN:
        hin = ho * s
        g = torch.Generator(device=dev).manual_seed(1)
        x = torch.randn(bs, hin, hin, cin, device=dev, generator=g).to(torch.bfloat16)
        w = torch.randn(cout, cin, k, k, device=dev, generator=g) / (cin * k * k) ** 0.5
        d = tr.make_desc(x, cout, k, s, (k - 1) // 2)
        packed = ops.pack_weights(w, cin_pad=cin)
        pk = tr.pack_weights_dgrad(w, s)
        ones_o, zeros_o = torch.ones(ops.cpad(cout), device=dev), torch.zeros(ops.cpad(cout), device=dev)
        ones_i, zeros_i = torch.ones(ops.cpad(cin), device=dev), torch.zeros(ops.cpad(cin), device=dev)
        z = torch.empty(bs, ho, ho, cout, dtype=torch.bfloat16, device=dev)
        part = torch.zeros((tr.stat_rows(), 2, ops.cpad(cout)), dtype=torch.float64, device=dev)
        dz = torch.randn(bs, ho, ho, cout, device=dev, generator=g).to(torch.bfloat16)
        dx = torch.randn(bs, hin, hin, cin, device=dev, generator=g).to(torch.bfloat16)
        zc = torch.randn(bs, hin, hin, cin, device=dev, generator=g).to(torch.bfloat16)
        stats = tuple(torch.rand(cin, device=dev) + 0.5 for _ in range(4))
        slope = torch.tensor([0.1], device=dev)
        flop = 2.0 * k * k * cin * cout * ho * ho * bs
        for what in ("fwd+stats", "dgrad", "dgrad+bnred"):
            cols = []
            for knob in ("0", "1", "2"):
                os.environ["RYOLO_MQ128"] = knob
                try:
                    if what == "fwd+stats":
                        name = ops.kernel_name_of(L.ryolo_conv_kernel_choice(ctypes.byref(d), 0, 1), k, s, cin)
                        us = timeit(lambda: tr.conv_fwd_stats(d, x, packed, ones_o, zeros_o, z, part=part, clear=False), a.reps)
                    elif what == "dgrad":
                        name = ops.kernel_name_of(L.ryolo_conv_dgrad_kernel_choice(ctypes.byref(d), 0), k, 1, cout)
                        us = timeit(lambda: tr.conv_dgrad(d, dz, pk, ones_i, zeros_i, dx, True), a.reps)
                    else:
                        rows = tr.dgrad_bnreduce_rows(d)
                        if rows <= 0:
                            cols.append("-")
                            continue
                        rp = torch.empty((rows, 3, cin), device=dev)
                        name = ops.kernel_name_of(L.ryolo_conv_dgrad_kernel_choice(ctypes.byref(d), 1), k, 1, cout)
                        us = timeit(lambda: tr.conv_dgrad_bnreduce(d, dz, pk, ones_i, zeros_i, dx, True, zc, stats, slope, rp), a.reps)
                    cols.append("%.1f us %.0f TF/s (%s)" % (us, flop / us / 1e6, name))
                except Exception as e:      # noqa: BLE001
                    cols.append("n/a (%s)" % type(e).__name__)
            print("k%d s%d %4d->%-4d @%-3d %-12s %s" % (k, s, cin, cout, ho, what, "  |  ".join(cols)), flush=True)
        os.environ.pop("RYOLO_MQ128", None)


if __name__ == "__main__":
    main()
