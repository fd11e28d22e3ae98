"""First-layer kernel (conv3x3_c8_direct, 3 -> 32 channels on the 8-channel padded input at 608 x 608): time and algorithmic TB/s for
several groups-per-wave settings (upper bits of ryolo_conv_desc.tile).  python tools/l0_bench.py [bs]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd
from rotate_yolov3_amd.model import hip_ops as ops
dev = torch.device("cuda:0")
bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn(bs, 608, 608, 8, device=dev).to(torch.bfloat16)
w = torch.randn(32, 3, 3, 3, device=dev) / 5
packed = ops.pack_weights(w, cin_pad=8)
sc = torch.ones(128, device=dev); sh = torch.zeros(128, device=dev)
out = torch.empty(bs, 608, 608, 32, device=dev, dtype=torch.bfloat16)
byts = x.numel() * 2 + out.numel() * 2
for gpw in (0, 8, 16, 32, 64, 128):
    tile = gpw << 16
    for _ in range(3):
        ops.conv2d_bn_act(x, packed, sc, sh, 32, 3, act=1, out=out, tile=tile)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.conv2d_bn_act(x, packed, sc, sh, 32, 3, act=1, out=out, tile=tile)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print("bs %d gpw %3d: %.3f ms  %.2f TB/s" % (bs, gpw, ms, byts / ms / 1e9))
