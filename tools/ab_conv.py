"""A/B of two builds of the conv kernel in ONE process (interleaved rounds): tools/_dbg/libold.so vs the current library."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd
from rotate_yolov3_amd import _lib
from rotate_yolov3_amd.model import hip_ops as ops
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
new = _lib.lib()
old = C.CDLL(os.path.join(ROOT, "tools/_dbg/libold.so"))
old.ryolo_conv2d_bn_act.restype = C.c_int
old.ryolo_conv2d_bn_act.argtypes = [C.POINTER(ops.ConvDesc)] + [C.c_void_p] * 7
dev = torch.device("cuda:0")
for (k, s, cin, cout, ho, res) in [(3, 1, 128, 256, 76, 1), (3, 1, 256, 512, 38, 1), (3, 1, 512, 1024, 19, 0), (1, 1, 512, 256, 38, 0), (1, 1, 256, 128, 76, 0)]:
    bs = 32
    x = torch.randn(bs, ho, ho, cin, device=dev).to(torch.bfloat16)
    w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    packed = ops.pack_weights(w, cin_pad=cin)
    sc = torch.ones(ops.cpad(cout), device=dev); sh = torch.zeros(ops.cpad(cout), device=dev)
    out = torch.empty(bs, ho, ho, cout, device=dev, dtype=torch.bfloat16)
    r = torch.randn(bs, ho, ho, cout, device=dev).to(torch.bfloat16) if res else None
    d = ops.ConvDesc(bs, ho, ho, cin, cout, k, s, (k - 1) // 2, cin, cout, cout if res else 0, 1, 0.1, 1, 0)
    st = torch.cuda.current_stream().cuda_stream
    def run(lib):
        lib.ryolo_conv2d_bn_act(C.byref(d), x.data_ptr(), packed.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr() if res else None, out.data_ptr(), st)
    res_t = {"old": [], "new": []}
    for rnd in range(5):
        for name, lib in (("old", old), ("new", new)):
            run(lib); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): run(lib)
            e1.record(); torch.cuda.synchronize()
            res_t[name].append(e0.elapsed_time(e1) / 20)
    print("k%d %d->%d @%d res=%d: old %s | new %s" % (k, cin, cout, ho, res, " ".join("%.4f" % v for v in res_t["old"]), " ".join("%.4f" % v for v in res_t["new"])), flush=True)
