"""Run one weight-gradient shape repeatedly (for rocprofv3 --pmc passes):  python tools/one_wgrad.py k s cin cout hout [reps]
(bs 64: the train step's batch; hout = output extent, the input is hout * s)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.model import hip_train_ops as tr  # noqa: E402

k, s, cin, cout, ho = [int(v) for v in sys.argv[1:6]]
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 10
bs = 64
dev = torch.device("cuda:0")
x = torch.randn(bs, ho * s, ho * s, cin, device=dev).to(torch.bfloat16)
dz = torch.randn(bs, ho, ho, cout, device=dev).to(torch.bfloat16)
d = tr.make_desc(x, cout, k, s, (k - 1) // 2)
ws = torch.empty(tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=dev)
g = torch.zeros(cout, cin, k, k, device=dev)
for _ in range(3):
    tr.conv_wgrad(d, x, dz, cin, g, True, ws)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    tr.conv_wgrad(d, x, dz, cin, g, True, ws)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print("wgrad k%d s%d %d->%d @%d bs%d  %.4f ms (tile kernel + split-K reduce)  %.1f TF/s  split-K partials %.1f MB" % (
    k, s, cin, cout, ho, bs, ms, 2.0 * k * k * cin * cout * ho * ho * bs / ms / 1e9, ws.numel() / 1e6))
