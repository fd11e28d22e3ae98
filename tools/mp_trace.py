"""Phase timeline of the multi-phase conv tile (debug variant, tile 15): per wave, s_memtime deltas between the four stamps of
each phase (load segment start, ready for barrier 1, barrier 1 released, MFMAs issued) for K tiles 4..11 of the first tile."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd import _lib  # noqa: E402
from rotate_yolov3_amd.model import hip_ops as ops  # noqa: E402

cin = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dev = torch.device("cuda:0")
L = _lib.lib()
L.ryolo_debug_conv_trace.argtypes = [C.c_void_p]
L.ryolo_debug_conv_trace.restype = None
buf = torch.zeros(8 * 128, dtype=torch.int32, device=dev)
L.ryolo_debug_conv_trace(buf.data_ptr())
x = torch.randn(32, 76, 76, cin, device=dev).clamp_(-3, 3).to(torch.bfloat16)
w = torch.randn(256, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
packed = ops.pack_weights(w, cin_pad=cin)
sc = torch.ones(256, device=dev)
sh = torch.zeros(256, device=dev)
out = torch.empty(32, 76, 76, 256, device=dev, dtype=torch.bfloat16)
for _ in range(3):
    ops.conv2d_bn_act(x, packed, sc, sh, 256, 3, act=1, out=out, tile=15)
torch.cuda.synchronize()
tr = buf.cpu().numpy().astype(np.int64).reshape(8, 128) & 0xffffffff
t0 = tr[:, 0].min()
print("stamps relative to the first one (cycles); per phase: load-seg | wait-b1 | mfma-seg | wait-b2(next phase start)")
for wv in (0, 4, 1, 5):
    s = tr[wv] - t0
    print("wave %d first stamp %d" % (wv, s[0]))
    for ph in range(0, 31):
        a, b, c, d = s[4 * ph:4 * ph + 4]
        nxt = s[4 * ph + 4]
        print("  kt %2d ph %d: load %4d  b1wait %4d  mfma %4d  b2wait %4d   | phase total %4d" % (
            4 + ph // 4, ph % 4, b - a, c - b, d - c, nxt - d, nxt - a))
per = (tr[:, 124] - tr[:, 0]) / 31.0
print("mean cycles per phase per wave:", np.round(per, 1), " -> per K tile %.0f" % (4 * per.mean()))
