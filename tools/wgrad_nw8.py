"""Weight-gradient tile variants of the MEASUREMENT build against the product's dispatch, isolated launches (tile kernel alone, bs 64,
median us over interleaved rounds, one process): ryolo_debug_wgrad_set(v) with v = 0 the product, 8 = wgrad_wide<256,128> on eight waves of
64 x 64, 9 = the 256 x 256 tile on eight 64 x 128 waves (C_in % 256 == 0), 11 = one filter tap per workgroup on the 128 x 64 tile for the
C_in = 64 layers (rounds 3-5).  eq = partial tiles bit-identical to the first variant's (other split count: relative error of the reduced
gradient).  profiles/r06_wgrad_nw8.txt, r06_wgrad_three_taps.txt.      python tools/wgrad_nw8.py [variant ...]"""
import ctypes as C, os, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RYOLO_HIP_LIB", os.path.join(ROOT, "rotate-yolov3_amd", "libryolo_hip_ablation.so"))
import torch
import rotate_yolov3_amd
from rotate_yolov3_amd import _lib
from rotate_yolov3_amd.model import hip_train_ops as tr
L = _lib.lib()
L.ryolo_debug_wgrad_set.argtypes = [C.c_int]; L.ryolo_debug_wgrad_set.restype = None
dev = torch.device("cuda:0")
SHAPES = [(3, 1, 64, 128, 152), (3, 2, 64, 128, 152), (3, 1, 128, 256, 76), (3, 1, 256, 512, 38), (3, 1, 512, 1024, 19), (3, 2, 256, 512, 38), (1, 1, 512, 256, 38), (1, 1, 768, 256, 38), (1, 1, 1024, 512, 19), (1, 1, 512, 256, 19), (1, 1, 1024, 504, 19), (1, 1, 512, 504, 38), (1, 1, 256, 504, 76)]
def timed(call, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): call()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
variants = [int(v) for v in sys.argv[1:]] or [0, 8, 9, 11]
for k, s, cin, cout, ho in SHAPES:
    bs = 64
    x = torch.randn(bs, ho * s, ho * s, cin, device=dev).to(torch.bfloat16)
    dz = torch.randn(bs, ho, ho, cout, device=dev).to(torch.bfloat16)
    d = tr.make_desc(x, cout, k, s, (k - 1) // 2)
    wss = {}
    for v in variants:
        L.ryolo_debug_wgrad_set(v)
        wss[v] = torch.zeros(tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=dev)
    g = torch.zeros(cout, cin, k, k, device=dev)
    gs = {}
    for v in variants:
        L.ryolo_debug_wgrad_set(v)
        gs[v] = torch.zeros(cout, cin, k, k, device=dev)
        tr.conv_wgrad(d, x, dz, cin, gs[v], False, wss[v])
    torch.cuda.synchronize()
    err = {v: float((gs[v] - gs[variants[0]]).abs().max() / gs[variants[0]].abs().max()) for v in variants}
    fl = 2.0 * k * k * cin * cout * ho * ho * bs
    def call(v):
        def f():
            tr.conv_wgrad_partials(d, x, dz, cin, g, True, wss[v])
        return f
    for v in variants:
        L.ryolo_debug_wgrad_set(v); call(v)()
    torch.cuda.synchronize()
    eq = {v: (bool(torch.equal(wss[v], wss[variants[0]])) if wss[v].numel() == wss[variants[0]].numel() else "%.1e" % err[v]) for v in variants}
    times = {v: [] for v in variants}
    for r in range(6):
        for v in variants:
            L.ryolo_debug_wgrad_set(v)
            us = timed(call(v), 20 if r else 10)
            if r: times[v].append(us)
    L.ryolo_debug_wgrad_set(0)
    print("k%d s%d %d->%d @%d (choice %d): " % (k, s, cin, cout, ho, L.ryolo_conv_wgrad_kernel_choice(C.byref(d))) + "  ".join(
        "v%d %.1f us %.0f TF/s eq=%s" % (v, statistics.median(times[v]), fl / statistics.median(times[v]) / 1e6, eq[v]) for v in variants))
