"""Run one conv shape repeatedly (for rocprofv3 --pmc passes):  python tools/one_layer.py k s cin cout hout [reps] [tileflags]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.model import hip_ops as ops  # noqa: E402

if os.environ.get("ONE_LAYER_NT_MIN") is not None:        # ablation build only: the output-size threshold of the non-temporal stores
    import ctypes
    from rotate_yolov3_amd import _lib
    L = _lib.lib()
    L.ryolo_debug_conv_nt_min.argtypes = [ctypes.c_longlong]
    L.ryolo_debug_conv_nt_min.restype = None
    L.ryolo_debug_conv_nt_min(int(os.environ["ONE_LAYER_NT_MIN"]))
k, s, cin, cout, ho = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
tile = int(sys.argv[7], 0) if len(sys.argv) > 7 else 0
bs = 32
dev = torch.device("cuda:0")
x = torch.randn(bs, ho * s, ho * s, cin, device=dev).to(torch.bfloat16)
w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
packed = ops.pack_weights(w, cin_pad=cin)
sc = torch.ones(ops.cpad(cout), device=dev)
sh = torch.zeros(ops.cpad(cout), device=dev)
out = torch.empty(bs, ho, ho, cout, device=dev, dtype=torch.bfloat16)
for t in ([tile] if len(sys.argv) > 7 else [0x200, 0]):
    for _ in range(3):
        ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, out=out, tile=t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, out=out, tile=t)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("k%d s%d %d->%d @%d tile=0x%x  %.4f ms  %.1f TF/s" % (k, s, cin, cout, ho, t, ms, 2.0 * k * k * cin * cout * ho * ho * bs / ms / 1e9))
