"""Where wgrad_wide<256,128>'s time goes: timing-only ablations on the ABLATION build of the library (-DRYOLO_MP_ABLATION; the variants
compute wrong results on purpose, the shipped library does not contain them) and the same launches on zero / constant operands (same
instruction stream, only the bits toggling in the MFMA datapath and with them the chip's power and sustained clock differ).
    python tools/wgrad_ablate.py [--reps 20]
Each row: the tile kernel alone (ryolo_conv2d_wgrad_partials; the split-K reduce is not launched), graph-free, HIP events."""
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
from rotate_yolov3_amd.model import hip_train_ops as tr  # noqa: E402

L = _lib.lib()
L.ryolo_debug_wgrad_set.argtypes = [C.c_int]
L.ryolo_debug_wgrad_set.restype = None
dev = torch.device("cuda:0")
VARS = [("product", 0), ("no partial-tile stores", 1), ("fragments read once (no LDS reads in the loop)", 2), ("no direct-to-LDS fills", 4),
        ("no LDS reads, no fills (MFMA + barrier only)", 6), ("no reads, no fills, no stores", 7)]
SHAPES = [(3, 1, 128, 256, 76), (3, 1, 256, 512, 38), (3, 1, 512, 1024, 19), (1, 1, 512, 256, 38), (1, 1, 1024, 512, 19)]


def make_call(d, x, dz, cin, g, ws):
    s = torch.cuda.current_stream().cuda_stream

    def call():
        rc = L.ryolo_conv2d_wgrad_partials(C.byref(d), x.data_ptr(), dz.data_ptr(), dz.shape[-1], cin, g.data_ptr(), 1, ws.data_ptr(), ws.numel(), s)
        assert rc == 0, rc
    return call


def timed(call, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        call()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    import statistics
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--bs", type=int, default=64)
    a = ap.parse_args()
    print("# wgrad tile kernel alone (no reduce), bs %d: median us per launch over %d interleaved rounds x %d launches (every row warmed up first: "
          "the chip's clock follows the load) and TFLOP/s; library %s" % (a.bs, a.rounds, a.reps, os.path.basename(os.environ["RYOLO_HIP_LIB"])))
    for k, s, cin, cout, ho in SHAPES:
        x = torch.randn(a.bs, ho * s, ho * s, cin, device=dev).to(torch.bfloat16)
        dz = torch.randn(a.bs, ho, ho, cout, device=dev).to(torch.bfloat16)
        d = tr.make_desc(x, cout, k, s, (k - 1) // 2)
        ws = torch.empty(tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=dev)
        g = torch.zeros(cout, cin, k, k, device=dev)
        fl = 2.0 * k * k * cin * cout * ho * ho * a.bs
        print("k%d %d->%d @%d  (kernel choice %d, split-K partials %.1f MB)" % (k, cin, cout, ho, L.ryolo_conv_wgrad_kernel_choice(C.byref(d)), ws.numel() / 1e6))
        rows = [(name, v, x, dz) for name, v in VARS]
        rows += [("product on zero operands", 0, torch.zeros_like(x), torch.zeros_like(dz)),
                 ("product on constant operands (1.0)", 0, torch.ones_like(x), torch.ones_like(dz)),
                 ("product on zero x, random dz", 0, torch.zeros_like(x), dz)]
        times = [[] for _ in rows]
        for r in range(a.rounds + 1):
            for i, (name, v, xx, zz) in enumerate(rows):
                L.ryolo_debug_wgrad_set(v)
                us = timed(make_call(d, xx, zz, cin, g, ws), a.reps if r else 10)
                if r:
                    times[i].append(us)
        L.ryolo_debug_wgrad_set(0)
        for (name, v, xx, zz), t in zip(rows, times):
            us = statistics.median(t)
            print("   %-52s %8.1f us  %7.1f TF/s" % (name, us, fl / us / 1e6))


if __name__ == "__main__":
    main()
