"""Tuning sweep of the BatchNorm backward reduce pass on the ABLATION build (slab count / minimum slab / non-temporal loads):
backward = reduce + finalize + apply time per tensor shape of the bs-64 step.  python tools/bn_tune.py"""
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
L.ryolo_debug_bn_set.argtypes = [C.c_int] * 5
L.ryolo_debug_bn_set.restype = None
dev = torch.device("cuda:0")


def t(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


# (slabs, min slab, nt forward, nt reduce, nt apply); nt: 0 never, 1 always, -1 by size (the product's rule)
SETTINGS = [(1024, 256, 0, 0, 0), (1024, 256, 1, 1, 0), (1024, 256, 1, 1, 1), (512, 256, 1, 1, 1), (1024, 256, -1, -1, -1)]
for (n, hw, c) in [(64, 76, 256), (64, 76, 128), (64, 304, 64), (64, 38, 512), (64, 19, 1024), (64, 608, 32), (64, 152, 128)]:
    z = torch.randn(n, hw, hw, c, device=dev).to(torch.bfloat16)
    dy = torch.randn(n, hw, hw, c, device=dev).to(torch.bfloat16)
    dz = torch.empty_like(z)
    scale = torch.rand(c, device=dev) + 0.5
    shift = torch.randn(c, device=dev) * 0.1
    mean = torch.randn(c, device=dev) * 0.1
    invstd = torch.rand(c, device=dev) + 0.5
    slope = torch.tensor([0.1], device=dev)
    dg, db, dsl = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(1, device=dev)
    el = n * hw * hw * c
    line = "%2dx%3d^2x%4d bwd ms:" % (n, hw, c)
    res = torch.randn(n, hw, hw, c, device=dev).to(torch.bfloat16)
    y = torch.empty_like(z)
    for (slabs, smin, nf, nr, na) in SETTINGS:
        L.ryolo_debug_bn_set(slabs, smin, nf, nr, na)
        ws = torch.empty(tr.bn_bwd_ws_bytes(n * hw * hw, c), dtype=torch.uint8, device=dev)
        b = t(lambda: tr.bn_act_bwd(z, dy, (mean, invstd, scale, shift), 1, slope, dz, dg, db, dsl, ws))
        f = t(lambda: tr.bn_act_fwd(z, scale, shift, 1, slope, y, residual=res))
        line += "  [%d/%d/nt %d %d %d] bwd %.3f (%.2f TB/s) fwd+res %.3f (%.2f)" % (slabs, smin, nf, nr, na, b, el * 10 / b / 1e9, f, el * 6 / f / 1e9)
    print(line, flush=True)
