"""Ablation of the conv K loop (debug build tools/_dbg/libdbg.so): flags in tile>>8:
 2 no A loads, 4 no B loads, 8 one MFMA of 16, 16 no stage() at all (no address VALU), 32 no ds_reads."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd
from rotate_yolov3_amd import _lib
_lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools/_dbg/libdbg.so")
from rotate_yolov3_amd.model import hip_ops as ops
dev = torch.device("cuda:0")
for (k, s, cin, cout, ho) in [(3, 1, 128, 256, 76), (3, 1, 256, 512, 38)]:
  for base in (1, 4):
    bs = 32
    x = torch.randn(bs, ho, ho, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    packed = ops.pack_weights(w, cin_pad=cin)
    sc = torch.ones(ops.cpad(cout), device=dev); sh = torch.zeros(ops.cpad(cout), device=dev)
    out = torch.empty(bs, ho, ho, cout, device=dev, dtype=torch.bfloat16)
    flop = 2.0 * k * k * cin * cout * ho * ho * bs
    res = []
    for dbg in (0, 6, 16, 8, 32, 16 + 8, 16 + 32, 8 + 32, 16 + 8 + 32):
        tile = base | (dbg << 8)
        for _ in range(2):
            ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, out=out, tile=tile)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, out=out, tile=tile)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append("d%-2d %.3f" % (dbg, ms))
    print("tile%d k%d %d->%d @%d: " % (base, k, cin, cout, ho) + " | ".join(res), flush=True)
