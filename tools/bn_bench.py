"""Time the BatchNorm/activation passes (forward, forward + shortcut, backward = reduce + finalize + apply) on the tensor shapes\nof the bs-64 train step and print the algorithmic TB/s.  python tools/bn_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import rotate_yolov3_amd
from rotate_yolov3_amd.model import hip_train_ops as tr
dev = torch.device("cuda:0")
def t(fn, n=20):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (n, hw, c) in [(64, 76, 256), (64, 76, 128), (64, 304, 64), (64, 38, 512), (64, 19, 1024), (64, 608, 32)]:
    z = torch.randn(n, hw, hw, c, device=dev).to(torch.bfloat16)
    dy = torch.randn(n, hw, hw, c, device=dev).to(torch.bfloat16)
    res = torch.randn(n, hw, hw, c, device=dev).to(torch.bfloat16)
    y = torch.empty_like(z); dz = torch.empty_like(z)
    scale = torch.rand(c, device=dev) + 0.5; shift = torch.randn(c, device=dev) * 0.1
    mean = torch.randn(c, device=dev) * 0.1; invstd = torch.rand(c, device=dev) + 0.5
    slope = torch.tensor([0.1], device=dev)
    dg, db, dsl = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(1, device=dev)
    ws = torch.empty(tr.bn_bwd_ws_bytes(n * hw * hw, c), dtype=torch.uint8, device=dev)
    el = n * hw * hw * c
    f0 = t(lambda: tr.bn_act_fwd(z, scale, shift, 1, slope, y))
    f1 = t(lambda: tr.bn_act_fwd(z, scale, shift, 1, slope, y, residual=res))
    b = t(lambda: tr.bn_act_bwd(z, dy, (mean, invstd, scale, shift), 1, slope, dz, dg, db, dsl, ws))
    print("%2dx%3d^2x%4d: fwd %.3f ms (%.2f TB/s)  fwd+res %.3f ms (%.2f TB/s)  bwd(reduce+fin+apply) %.3f ms (%.2f TB/s)" % (
        n, hw, c, f0, el * 4 / f0 / 1e9, f1, el * 6 / f1 / 1e9, b, el * 10 / b / 1e9))
