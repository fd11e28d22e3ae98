import copy, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd
from rotate_yolov3_amd.cfg import make_cfg
from rotate_yolov3_amd.model.models import Darknet
dev = torch.device("cuda:0")
size, bs = 160, 4
cfg = make_cfg.darknet53(size, size)
torch.manual_seed(0)
ref = Darknet(cfg, {"context_factor": 1.0})
with torch.no_grad():
    g = torch.Generator().manual_seed(5)
    for name, t in ref.state_dict().items():
        if t.dim() == 4: t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) * (6.0 / t[0].numel()) ** 0.5)
        elif name.endswith("BatchNorm2d.weight"): t.copy_(0.5 + torch.rand(t.shape, generator=g))
        elif name.endswith("BatchNorm2d.bias"): t.copy_(torch.randn(t.shape, generator=g) * 0.2)
ref = ref.to(dev).train(); ref.nc, ref.arc = 1, "default"; ref.backend = "torch"
hip = copy.deepcopy(ref); hip.backend = "hip"; hip._engines = {}
x = torch.rand(bs, 3, size, size, generator=torch.Generator().manual_seed(0)).to(dev)
outs32, outs16 = {}, {}
def hook(store, i):
    def fn(m, inp, out): store[i] = out.detach().float()
    return fn
hs = []
for i, (d, m) in enumerate(zip(ref.module_defs, ref.module_list)):
    if d["type"] == "convolutional": hs.append(m.register_forward_hook(hook(outs32, i)))
with torch.no_grad(): ref(x)
for h in hs: h.remove()
for i, (d, m) in enumerate(zip(ref.module_defs, ref.module_list)):
    if d["type"] == "convolutional": m.register_forward_hook(hook(outs16, i))
with torch.no_grad(), torch.autocast(device_type="cuda", dtype=torch.bfloat16): ref(x)
eng = hip.train_engine(x.shape, dev)
eng.forward(x)
torch.cuda.synchronize()
def rel(a, b): return (a - b).abs().mean().item() / (b.abs().mean().item() + 1e-9)
for b in eng.blocks[::6] + eng.blocks[-3:]:
    i = b["i"]
    got = b["y"].float().permute(0, 3, 1, 2)
    if b["res"] is not None: got = got - b["res"].float().permute(0, 3, 1, 2)
    print("L%-3d C=%4d hw=%3d  hip-vs-autocast %.4f   hip-vs-fp32 %.4f   autocast-vs-fp32 %.4f" % (i, b["C"], b["y"].shape[1], rel(got, outs16[i]), rel(got, outs32[i]), rel(outs16[i], outs32[i])))
