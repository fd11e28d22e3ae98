"""Time ryolo_rnms on the bench's 50 000-box workload (configs[2]); run under rocprofv3 --kernel-trace --stats for the
per-kernel split.  usage: python tools/nms_time.py [n] [reps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.utils.synthetic import random_boxes  # noqa: E402
from rotate_yolov3_amd.utils.nms import r_nms as m  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
d = torch.from_numpy(random_boxes(n, seed=0)).cuda()
for _ in range(3):
    k = m.r_nms(d, 0.5)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    k = m.r_nms(d, 0.5)
torch.cuda.synchronize()
print("n=%d kept=%d  %.3f ms per call (incl. the keep-count readback)" % (n, len(k), (time.perf_counter() - t0) / reps * 1e3))
