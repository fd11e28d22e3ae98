"""Layer-0 training kernels (statistics-only, forward, backward reduce, backward apply: ryolo_conv0_*) at bs 64: ms per launch.
python tools/l0_modes.py   (RYOLO_HIP_LIB selects the library build)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa
from rotate_yolov3_amd.model import hip_ops as ops, hip_train_ops as tr
dev = torch.device("cuda:0")
bs = 64
x = torch.zeros(bs, 608, 608, 8, device=dev, dtype=torch.bfloat16)
x[..., :3] = torch.randn(bs, 608, 608, 3, device=dev).to(torch.bfloat16)
w = torch.randn(32, 3, 3, 3, device=dev) / 5
packed = ops.pack_weights(w, cin_pad=8)
ones, zeros = torch.ones(128, device=dev), torch.zeros(128, device=dev)
gamma, beta = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.3
slope = torch.tensor([0.1], device=dev)
d = tr.make_desc(x, 32, 3, 1, 1)
M = bs * 608 * 608
part = tr.conv_fwd_stats(d, x, packed, ones, zeros, None)
st = tr.bn_finalize(part, 32, M, gamma, beta)
y = torch.empty(bs, 608, 608, 32, device=dev, dtype=torch.bfloat16)
dy = torch.randn(bs, 608, 608, 32, device=dev).to(torch.bfloat16)
dz = torch.empty_like(y)
dg, db, ds = torch.zeros(32, device=dev), torch.zeros(32, device=dev), torch.zeros(1, device=dev)
ws0 = tr.conv0_bn_bwd_ws(dev)
def t(fn, reps=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / reps)
    return sorted(ts)[2]
def stats():
    p = tr.conv_fwd_stats(d, x, packed, ones, zeros, None)
    tr.bn_finalize(p, 32, M, gamma, beta)
print("lib", os.environ.get("RYOLO_HIP_LIB", "default"))
print("stats-only (+finalize) %.3f ms" % t(stats))
print("forward                %.3f ms" % t(lambda: tr.conv0_bn_act_fwd(d, x, packed, st[2], st[3], 1, slope, y)))
print("backward (red+app)     %.3f ms" % t(lambda: tr.conv0_bn_bwd(d, x, packed, dy, st, 1, slope, dz, dg, db, ds, ws0)))
