"""Golden vectors for build_targets / compute_loss FROM THE REFERENCE's own model/loss.py, imported here with the stub
recipe of SURVEY.md section 8c (plus: Tensor.cuda = identity for the hard .cuda() at loss.py:197, and pre-1.5
integer `/` semantics for the index division at loss.py:239).

    python tests/golden/gen_loss_golden.py        (needs /root/reference)

Fixture loss_d53_96.npz: the three head tensors of a procedural-weight Darknet-53 in TRAIN mode on rand(2,3,96,96),
5 targets over 2 images (the 3 labels of HRSC2016/train/100001675.txt + 2 synthetic ones, one of them shaped so that
no anchor passes the IoU/angle gate -> exercises the best-anchor fallback), hyp = cfg/HRSC+/hyp.py;
outputs: per-head indices / tbox / anchor_vec / tcls, loss, loss_items, d loss / d p per head.
Fixture loss_arcs_d53_96.npz: the same heads and targets through the reference's other arcs (Fdefault, uBCE, uCE, FuBCE, FuCE).
"""
import os
import sys
import tempfile

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

from tests.golden.gen_model_golden import install_stubs  # noqa: E402
from tests.procedural import fill_procedural  # noqa: E402


def main():
    install_stubs()
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    torch.Tensor.cuda = lambda self, *a, **k: self
    _td = torch.Tensor.__truediv__

    def legacy_div(self, other):
        if not self.is_floating_point() and not (isinstance(other, torch.Tensor) and other.is_floating_point()) \
                and not isinstance(other, float):
            return torch.div(self, other, rounding_mode="floor")
        return _td(self, other)
    torch.Tensor.__truediv__ = legacy_div

    from model import models as rmodels
    from model import loss as rloss
    from utils.utils import hyp_parse
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.cfg import make_cfg

    hyp = hyp_parse("cfg/HRSC+/hyp.py")
    tmp = tempfile.mkdtemp()
    cfg_path = os.path.join(tmp, "d53.cfg")
    open(cfg_path, "w").write(make_cfg.darknet53(width=96, height=96))
    model = rmodels.Darknet(cfg_path, hyp)
    fill_procedural(model)
    model.train()
    model.nc = 1
    model.arc = "default"
    model.hyp = hyp
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(0))
    lab = np.loadtxt("HRSC2016/train/100001675.txt").reshape(-1, 6)        # cls x y w h a
    targets = np.zeros((5, 7), np.float32)
    targets[:3, 1:] = lab
    targets[:3, 0] = [0, 0, 1]
    targets[3] = [1, 0, 0.31, 0.62, 0.93, 0.012, 0.4]      # extreme aspect: no anchor clears iou_t -> fallback branch
    targets[4] = [0, 0, 0.55, 0.25, 0.20, 0.05, -1.1]
    targets = torch.from_numpy(targets)
    with torch.no_grad():
        pred = model(x)
    p_in = [t.detach().clone().requires_grad_(True) for t in pred]
    t_in = targets.clone()
    loss, items = rloss.compute_loss(p_in, t_in, model, hyp)
    loss.backward()
    tcls, tbox, indices, av = rloss.build_targets(model, targets.clone(), hyp)
    out = dict(targets=targets.numpy(), loss=loss.detach().numpy(), loss_items=items.numpy(),
               hyp_keys=np.array(sorted(hyp)), hyp_vals=np.array([float(hyp[k]) for k in sorted(hyp)]))
    for k in range(3):
        out["p%d" % k] = pred[k].numpy()
        out["g%d" % k] = p_in[k].grad.numpy()
        out["tbox%d" % k] = tbox[k].numpy()
        out["tcls%d" % k] = tcls[k].numpy()
        out["av%d" % k] = av[k].numpy()
        out["idx%d" % k] = np.stack([t.numpy() for t in indices[k]], 0)
        out["ng%d" % k] = model.module_list[model.yolo_layers[k]].ng.numpy()
        out["anchor_vec%d" % k] = model.module_list[model.yolo_layers[k]].anchor_vec.numpy()
    np.savez_compressed(os.path.join(OUT, "loss_d53_96.npz"), **out)
    # the other arcs of compute_loss (loss.py:284-286 focal wrappers, :350-361 unified BCE / CE heads) on the same heads and
    # targets: loss, items and d loss / d p per head -> loss_arcs_d53_96.npz (inputs are those of loss_d53_96.npz)
    arcs = {}
    for arc in ("Fdefault", "uBCE", "uCE", "FuBCE", "FuCE"):
        model.arc = arc
        q_in = [t.detach().clone().requires_grad_(True) for t in pred]
        l2, it2 = rloss.compute_loss(q_in, targets.clone(), model, hyp)
        l2.backward()
        arcs["loss_" + arc] = l2.detach().numpy()
        arcs["items_" + arc] = it2.numpy()
        for k in range(3):
            arcs["g%d_%s" % (k, arc)] = q_in[k].grad.numpy()
        print(arc, float(l2), it2.numpy())
    np.savez_compressed(os.path.join(OUT, "loss_arcs_d53_96.npz"), **arcs)
    model.arc = "default"
    os.chdir(cwd)
    print("loss", float(loss), items.numpy(), [len(t) for t in tcls])


if __name__ == "__main__":
    main()
