"""Cost of the BatchNorm-statistics epilogue: the training forward conv (ryolo_conv2d_bn_act_stats, linear) against the plain
conv (same kernel family without statistics) on the conv_mp shapes at bs 64.  python tools/gen_cost.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa
from rotate_yolov3_amd.model import hip_ops as ops, hip_train_ops as tr
dev = torch.device("cuda:0")
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for (cin, cout, ho, k) in [(8, 32, 608, 3), (128, 256, 76, 3), (256, 512, 38, 3), (512, 1024, 19, 3), (64, 128, 152, 3), (32, 64, 304, 3), (256, 128, 76, 1),
                           (512, 256, 38, 1), (1024, 512, 19, 1), (128, 64, 152, 1), (64, 32, 304, 1)]:
    bs = 64
    x = torch.randn(bs, ho, ho, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    pk = ops.pack_weights(w, cin_pad=cin)
    ones = torch.ones(ops.cpad(cout), device=dev); zeros = torch.zeros(ops.cpad(cout), device=dev)
    z = torch.empty(bs, ho, ho, cout, dtype=torch.bfloat16, device=dev)
    d = tr.make_desc(x, cout, k, 1, k // 2)
    part = torch.zeros(512, 2, ops.cpad(cout), dtype=torch.float64, device=dev)
    a = t(lambda: tr.conv_fwd_plain(d, x, pk, ones, zeros, z))
    b = t(lambda: tr.conv_fwd_stats(d, x, pk, ones, zeros, z, part=part, clear=False))
    fl = 2 * k * k * cin * cout * ho * ho * bs
    print("%dx%d %4d->%4d @%3d bs64: plain %.3f ms (%.0f TF/s)   with statistics %.3f ms (%.0f TF/s)  +%.1f %%" % (
        k, k, cin, cout, ho, a, fl / a / 1e9, b, fl / b / 1e9, (b / a - 1) * 100))
