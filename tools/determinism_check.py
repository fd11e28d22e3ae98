import sys, copy, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd
from rotate_yolov3_amd.model.models import Darknet
from rotate_yolov3_amd.model.loss import compute_loss
from rotate_yolov3_amd.utils.synthetic import synthetic_targets
from tests.test_train_engine_gpu import MINI_CFG, HYP
dev = torch.device("cuda:0")
torch.manual_seed(3)
m = Darknet(MINI_CFG, dict(HYP)).to(dev).train()
m.nc, m.arc = 1, "default"
x = torch.rand(8, 3, 128, 128, generator=torch.Generator().manual_seed(0)).to(dev)
tg = synthetic_targets(8, seed=6, device=dev)
runs = []
for r in range(5):
    m.zero_grad(set_to_none=True)
    p = m(x)
    loss, items = compute_loss([q.float() for q in p], tg.clone(), m, m.hyp)
    loss.backward()
    runs.append(({k: v.grad.detach().clone() for k, v in m.named_parameters()}, [q.detach().clone() for q in p], float(loss)))
base = runs[0]
for r in range(1, 5):
    worst = []
    for k in base[0]:
        a, b = base[0][k].double().flatten(), runs[r][0][k].double().flatten()
        d = float((a - b).norm() / (a.norm() + 1e-30))
        worst.append((d, k))
    worst.sort(reverse=True)
    pd = max(float((a - b).abs().max()) for a, b in zip(base[1], runs[r][1]))
    print("run", r, "loss diff", runs[r][2] - base[2], "head max abs diff", pd, "worst grads", [(round(d, 5), k) for d, k in worst[:4]])

# ---- which layer diverges first between step 0 and step 1 of a FRESH engine?
torch.manual_seed(3)
m2 = Darknet(MINI_CFG, dict(HYP)).to(dev).train()
m2.nc, m2.arc = 1, "default"
snaps = []
for r in range(3):
    m2.zero_grad(set_to_none=True)
    p = m2(x)
    eng = [e for e in m2._engines.values() if hasattr(e, "blocks")][0]
    torch.cuda.synchronize()
    snaps.append([(b['i'], b['z'].float().clone(), [s.clone() for s in b['stats']] if b['stats'] is not None else None,
                   b['y'].float().clone()) for b in eng.blocks])
    loss, items = compute_loss([q.float() for q in p], tg.clone(), m2, m2.hyp)
    loss.backward()
for (i, z0, s0, y0), (_, z1, s1, y1), (_, z2, s2, y2) in zip(*snaps):
    dz = float((z0 - z1).abs().max())
    dy = float((y0 - y1).abs().max())
    ds = max(float((a - b).abs().max() / (a.abs().max() + 1e-30)) for a, b in zip(s0, s1)) if s0 is not None else 0.0
    d12 = float((y1 - y2).abs().max())
    print("layer %2d  step0-vs-1: max|dz| %.4g  stats rel %.3g  max|dy| %.4g   step1-vs-2 max|dy| %.4g" % (i, dz, ds, dy, d12))
