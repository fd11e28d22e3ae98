"""Where the time of a conv_pw.hip launch goes: the same launch with one ingredient removed at a time (ablation build of the
library, -DRYOLO_MP_ABLATION; the variants compute wrong results on purpose).  python tools/pw_ablate.py [--bs 32]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RYOLO_HIP_LIB", os.path.join(ROOT, "rotate-yolov3_amd", "libryolo_hip_ablation.so"))

import torch  # noqa: E402

import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd import _lib  # noqa: E402
from rotate_yolov3_amd.model import hip_ops as ops  # noqa: E402

L = _lib.lib()
L.ryolo_debug_convpw_set.argtypes = [C.c_int]
L.ryolo_debug_convpw_set.restype = None
L.ryolo_debug_convpw_trace.argtypes = [C.c_void_p]
L.ryolo_debug_convpw_trace.restype = None

VARIANTS = [("product", 0), ("no MFMA", 4), ("no stores", 8), ("nothing but the loop", 63), ("exit at once", 64), ("exit after the prologue", 128),
            ("first row block only", 256), ("first row block only, skeleton", 256 | 63)]


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
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    for (cin, cout, hw) in [(512, 256, 38), (1024, 512, 19)]:
        x = torch.randn(a.bs, hw, hw, cin, device=dev).to(torch.bfloat16)
        w = torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5
        pk = ops.pack_weights(w, cin_pad=cin)
        sc, sh = torch.ones(ops.cpad(cout), device=dev), torch.zeros(ops.cpad(cout), device=dev)
        out = torch.empty(a.bs, hw, hw, cout, device=dev, dtype=torch.bfloat16)
        print("%d->%d @%d bs %d" % (cin, cout, hw, a.bs))
        for cap in (0,):
            for name, flags in VARIANTS:
                L.ryolo_debug_convpw_set(flags)
                t = timeit(lambda: ops.conv2d_bn_act(x, pk, sc, sh, cout, 1, act=1, out=out, tile=13 | (cap << 16)))
                print("   grid cap %2d  %-28s %7.1f us" % (cap, name, t), flush=True)
            L.ryolo_debug_convpw_set(0)
        t = timeit(lambda: ops.conv2d_bn_act(x, pk, sc, sh, cout, 1, act=1, out=out, tile=1))
        print("   igemm 128x128 tile            %7.1f us" % t, flush=True)
        # cycle stamps of the product schedule (third of three launches)
        buf = torch.zeros(3 * 2 * 64, dtype=torch.int32, device=dev)
        L.ryolo_debug_convpw_trace(buf.data_ptr())
        L.ryolo_debug_convpw_set(512)
        for _ in range(3):
            buf.zero_()
            ops.conv2d_bn_act(x, pk, sc, sh, cout, 1, act=1, out=out, tile=13)
        torch.cuda.synchronize()
        L.ryolo_debug_convpw_set(0)
        L.ryolo_debug_convpw_trace(None)
        tr = buf.cpu().numpy().astype("int64").reshape(3, 2, 64) & 0xffffffff
        t00 = int(tr[0, 0, 0])
        kt = cin // 64
        print("   trace (cycles of s_memtime; per unit: wait | barrier | multiply), rows = workgroup, wave")
        for wg in range(3):
            for wv in range(2):
                r = [int(v) for v in tr[wg, wv]]
                n = max(i for i, v in enumerate(r) if v) + 1 if any(r) else 0
                if n < 3:
                    continue
                d = lambda a, b: (r[b] - r[a]) & 0xffffffff
                line = "   wg%d w%d start %6d prologue %5d |" % (wg, wv, (r[0] - t00) & 0xffffffff, d(0, 1))
                i = 2
                blk = 0
                while i + 3 * kt < n:
                    units = " ".join("%d|%d|%d" % (d(i + 3 * u - 1, i + 3 * u), d(i + 3 * u, i + 3 * u + 1), d(i + 3 * u + 1, i + 3 * u + 2)) for u in range(kt))
                    line += " B%d[%s] epi %d |" % (blk, units, d(i + 3 * kt - 1, i + 3 * kt))
                    i += 3 * kt + 1
                    blk += 1
                line += " drain %d total %d" % (d(n - 2, n - 1), d(0, n - 1))
                print(line, flush=True)
    # an empty kernel's launch-to-launch time on this box, for scale
    y = torch.zeros(64, device=dev)
    print("torch elementwise launch-to-launch: %.1f us" % timeit(lambda: y.add_(1.0)))


if __name__ == "__main__":
    main()
