"""A/B of wgrad_wide<256,128>'s K-step schedules on isolated launches (tile kernel alone, no reduce): tile bit 0x8000 selects the
fills-under-MFMAs schedule.  Rounds are interleaved (A B A B ...) after a warm-up, medians reported; the two schedules add in the same
order, so the partial tiles must be bit-identical (checked).      python tools/wgrad_sched_ab.py [--rounds 5] [--reps 20]"""
import argparse
import ctypes as C
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd import _lib  # noqa: E402
from rotate_yolov3_amd.model import hip_train_ops as tr  # noqa: E402

L = _lib.lib()
dev = torch.device("cuda:0")
SHAPES = [(3, 1, 128, 256, 76), (3, 1, 256, 512, 38), (3, 1, 512, 1024, 19), (3, 2, 128, 256, 76), (1, 1, 512, 256, 38), (1, 1, 1024, 512, 19),
          (1, 1, 768, 256, 38)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--bs", type=int, default=64)
    ap.add_argument("--variants", default="0,0x8000")
    a = ap.parse_args()
    variants = [int(v, 0) for v in a.variants.split(",")]
    s = torch.cuda.current_stream().cuda_stream
    print("# wgrad tile kernel alone, bs %d, median of %d interleaved rounds x %d launches; tile bits %s" % (a.bs, a.rounds, a.reps, a.variants))
    for k, st, cin, cout, ho in SHAPES:
        x = torch.randn(a.bs, ho * st, ho * st, cin, device=dev).to(torch.bfloat16)
        dz = torch.randn(a.bs, ho, ho, cout, device=dev).to(torch.bfloat16)
        g = torch.zeros(cout, cin, k, k, device=dev)
        ds, wss = [], []
        for v in variants:
            d = tr.make_desc(x, cout, k, st, (k - 1) // 2, tile=v)
            ds.append(d)
            wss.append(torch.zeros(tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=dev))

        def call(i):
            rc = L.ryolo_conv2d_wgrad_partials(C.byref(ds[i]), x.data_ptr(), dz.data_ptr(), dz.shape[-1], cin, g.data_ptr(), 1,
                                               wss[i].data_ptr(), wss[i].numel(), s)
            assert rc == 0, rc
        for i in range(len(variants)):
            for _ in range(10):
                call(i)
        torch.cuda.synchronize()
        same = all(torch.equal(wss[0], w) for w in wss[1:])
        times = [[] for _ in variants]
        for _ in range(a.rounds):
            for i in range(len(variants)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(a.reps):
                    call(i)
                e1.record()
                torch.cuda.synchronize()
                times[i].append(e0.elapsed_time(e1) / a.reps * 1e3)
        fl = 2.0 * k * k * cin * cout * ho * ho * a.bs
        row = "k%d s%d %4d->%4d @%2d  " % (k, st, cin, cout, ho)
        for i, v in enumerate(variants):
            us = statistics.median(times[i])
            row += "| tile 0x%04x (choice %d): %7.1f us %7.1f TF/s " % (v, L.ryolo_conv_wgrad_kernel_choice(C.byref(ds[i])), us, fl / us / 1e6)
        print(row + "| partial tiles %s" % ("bit-identical" if same else "DIFFER"))


if __name__ == "__main__":
    main()
