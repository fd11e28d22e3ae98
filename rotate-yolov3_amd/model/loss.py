"""build_targets / compute_loss -- mirror of the reference's model/loss.py (build_targets :161-258,
compute_loss :266-367, h_iou_loss :119-122, FocalLoss :126-146) and utils/utils.py:wh_iou (:346-361).

Same signatures and return values:
    tcls, tbox, indices, anchor_vec = build_targets(model, targets, hyp)
    loss[1], loss_items[4] = compute_loss(p, targets, model, hyp)      items = (lobj, lcls, lreg, loss)
`targets` rows are (image, class, x, y, w, h, angle) with x, y, w, h normalised to [0, 1].

Device-agnostic (the reference hard-codes .cuda() at loss.py:197 and torch.cuda.FloatTensor at :268) and usable
under a DistributedDataParallel-style wrapper (the reference dereferences model.module_list on the wrapper and
crashes, loss.py:313).  Reference quirks that change numbers are KEPT and marked:
  Q1  `liou` is overwritten per head, then added into lreg (loss.py:322-323);
  Q2  the angle gate uses the LAST head's target/anchor angles for every head (loss.py:227) -- identical values;
  Q3  the best-anchor fallback divides indices with `/` under pre-1.5 integer semantics (loss.py:239) -> floor;
  Q6  the NoSampler mask is taken before positives are written, so objectness BCE runs over ALL cells (:298-299);
  Q7  `targets` is rescaled in place once per head by the context factor (loss.py:176-178; no-op when it is 1);
  Q8  exp(wh) is clamped at 1e3 before the anchor multiply (loss.py:315).
The samplers (loss.py:27-103) are unreachable (`sampling = False`, :289) and not reproduced.
Extension (not in the reference): hyp['riou'] = 1 replaces the axis-aligned wh_iou term of lreg by 1 - ROTATED IoU of the
decoded prediction vs the target, differentiated through the polygon overlap (HIP kernel csrc/riou_grad.h, forward and
backward; gradient checked against central differences of the fp64 polygon oracle in tests/test_riou_loss_gpu.py).
"""
import math

import torch
import torch.nn as nn


def wh_iou(box1, box2):
    """IoU of wh1 to wh2 (utils/utils.py:346-361).  box1 [2] vs box2 [n,2], or elementwise [n,2] vs [n,2]."""
    if box1.shape != box2.shape:
        box2 = box2.t()
        w1, h1 = box1[0], box1[1]
        w2, h2 = box2[0], box2[1]
    else:
        w1, h1 = box1[:, 0], box1[:, 1]
        w2, h2 = box2[:, 0], box2[:, 1]
    inter_area = torch.min(w1, w2) * torch.min(h1, h2)
    union_area = (w1 * h1 + 1e-16) + w2 * h2 - inter_area
    return inter_area / union_area


def h_iou_loss(input, target):
    return 1.0 - wh_iou(input, target)


class FocalLoss(nn.Module):
    """Wraps focal loss around an existing loss_fcn (loss.py:126-146)."""

    def __init__(self, loss_fcn, gamma=0.5, alpha=1, reduction='mean'):
        super(FocalLoss, self).__init__()
        loss_fcn.reduction = 'none'
        self.loss_fcn = loss_fcn
        self.gamma = gamma
        self.alpha = alpha
        self.reduction = reduction

    def forward(self, input, target):
        loss = self.loss_fcn(input, target)
        loss = loss * (self.alpha * (1.000001 - torch.exp(-loss)) ** self.gamma)
        if self.reduction == 'mean':
            return loss.mean()
        elif self.reduction == 'sum':
            return loss.sum()
        return loss


def _core(model):
    """The Darknet under an optional DataParallel / DDP-style wrapper."""
    return model.module if hasattr(model, 'module') and hasattr(model.module, 'module_list') else model


def build_targets(model, targets, hyp):
    core = _core(model)
    nt = len(targets)
    tcls, tbox, indices, av, square_ious = [], [], [], [], []
    dev = targets.device
    cf = float(hyp['context_factor'])
    all_ious = None
    t_gwha = None
    anchor_vec = None
    na = 0
    for i in core.yolo_layers:
        layer = core.module_list[i]
        ng, anchor_vec = layer.ng.to(dev), layer.anchor_vec.to(dev)
        # Q7: in place, once per head
        targets[:, 4] += targets[:, 5] * (cf - 1)
        targets[:, 5] *= cf
        t, a = targets, []
        gwha = t[:, 4:7].clone()
        gwha[:, :-1] *= ng
        if nt:
            # [na, nt]: every anchor's wh IoU with every target (loss.py:188 stacks one wh_iou call per anchor; one
            # broadcast expression here -- the same fp32 operations per element, so the same bits, in 8 launches not 8*na)
            aw, ah = anchor_vec[:, 0:1], anchor_vec[:, 1:2]
            gw, gh = gwha[:, 0].unsqueeze(0), gwha[:, 1].unsqueeze(0)
            inter = torch.min(aw, gw) * torch.min(ah, gh)
            all_ious = inter / ((aw * ah + 1e-16) + gw * gh - inter)
            na = len(anchor_vec)
            a = torch.arange(na, device=dev).view((-1, 1)).repeat([1, nt]).view(-1)
            t = targets.repeat([na, 1])
            gwha = gwha.repeat([na, 1])
            square_ious.append(all_ious.view(-1))
        b, c = t[:, :2].long().t()
        gxy = t[:, 2:4] * ng
        gi, gj = gxy.long().t()
        indices.append([b, a, gj, gi])
        gxy = gxy - gxy.floor()
        t_gwha = gwha.clone()
        tbox.append(torch.cat((gxy, gwha), 1))
        av.append(anchor_vec[a] if nt else anchor_vec[:0])
        tcls.append(c)
        if c.shape[0]:
            assert c.max() <= core.nc, 'Target classes exceed model classes'

    if nt:
        h = core.hyp if getattr(core, 'hyp', None) else hyp
        # Q2: last head's angles, valid for all heads
        angle_offset = (t_gwha[:, -1] - anchor_vec[:, -1].view((-1, 1)).repeat([1, nt]).view(-1)).abs()
        big = angle_offset > 0.5 * math.pi
        angle_offset[big] = math.pi - angle_offset[big]
        j_a = angle_offset < h['ang_t']
        j = [(sq > h['iou_t']) & j_a for sq in square_ious]
        gt_j = torch.stack([ju.reshape(all_ious.shape).max(0)[0] for ju in j], 0).t()   # [nt, n_heads]
        num_layers = len(core.yolo_layers)
        orphan = (~gt_j.any(1)).nonzero().flatten().tolist()        # host sync, as the reference's any() (loss.py:236)
        for gt_id in orphan:
            gt_ious = torch.cat([sq[gt_id::nt] for sq in square_ious], 0)
            best = torch.where(gt_ious == gt_ious.max(0)[0])[0]
            layer_id = int(best[0]) // na                            # Q3: floor semantics of the legacy `/`
            k = angle_offset[gt_id::nt].repeat(num_layers)[best].min(0)[1]
            best = best[k]
            j[layer_id][(best % na) * nt + gt_id] = True
        assert sum(int(m.sum()) for m in j) >= nt, 'something wrong at target building'
        for lid, m in enumerate(j):
            tbox[lid] = tbox[lid][m]
            tcls[lid] = tcls[lid][m]
            av[lid] = av[lid][m]
            indices[lid] = [indices[lid][k][m] for k in range(4)]
    return tcls, tbox, indices, av


def compute_loss(p, targets, model, hyp):
    core = _core(model)
    dev = p[0].device
    fused = getattr(core, 'fused_loss', None)          # Darknet.enable_fused_loss(): one hipGraph replay (loss_static.py)
    if fused is not None:
        out = fused.try_call(p, targets, hyp)
        if out is not None:
            return out

    def ft(v):
        return torch.tensor(v, dtype=torch.float32, device=dev)

    lcls, liou, lobj, lreg = ft([0.]), ft([0.]), ft([0.]), ft([0.])
    tcls, tbox, indices, anchor_vec = build_targets(model, targets, hyp)
    h = core.hyp if getattr(core, 'hyp', None) else hyp
    arc = core.arc

    BCEcls = nn.BCEWithLogitsLoss(pos_weight=ft([h['cls_pw']]))
    BCEobj = nn.BCEWithLogitsLoss(pos_weight=ft([h['obj_pw']]))
    BCE = nn.BCEWithLogitsLoss()
    CE = nn.CrossEntropyLoss()
    SM = nn.SmoothL1Loss(reduction='mean')
    if 'F' in arc:
        g = h['fl_gamma']
        BCEcls, BCEobj, BCE, CE, SM = FocalLoss(BCEcls, g), FocalLoss(BCEobj, g), FocalLoss(BCE, g), FocalLoss(CE, g), \
            FocalLoss(SM, g)

    for i, pi in enumerate(p):
        b, a, gj, gi = indices[i]
        tobj = torch.zeros_like(pi[..., 0])
        nb = len(b)
        if nb:
            ps = pi[b, a, gj, gi]
            tobj[b, a, gj, gi] = 1.0
            avec = core.module_list[core.yolo_layers[i]].anchor_vec.to(dev)
            pxy = torch.sigmoid(ps[:, 0:2])
            pwh = torch.exp(ps[:, 2:4]).clamp(max=1E3) * avec[a][:, :-1]          # Q8
            pa = torch.atan(ps[:, 4]) + avec[a][:, -1]
            pbox = torch.cat((pxy, pwh, pa.unsqueeze(1)), 1)
            tb = tbox[i].to(pbox.dtype)
            liou = h_iou_loss(tb[:, 2:4], pbox[:, 2:4]).mean()                     # Q1: overwritten per head
            if h.get('riou', 0):
                # the build's extension (the reference has no rotated-IoU loss, loss.py:322 is the
                # axis-aligned wh_iou): the IoU term becomes 1 - ROTATED IoU of the decoded box vs the target, with the
                # polygon-overlap gradient flowing into all five box parameters (HIP kernel csrc/riou_grad.h; no CPU path)
                from .hip_train_ops import RotatedIoU
                liou = (1.0 - RotatedIoU.apply(pbox, tb)).mean()
            lreg = lreg + SM(pbox[:, [0, 1]], tb[:, [0, 1]]) + 2 * SM(pbox[:, 4], tb[:, 4]) + liou * h['giou']
            if 'default' in arc and core.nc > 1:
                t = torch.zeros_like(ps[:, 6:])
                t[range(nb), tcls[i]] = 1.0
                lcls = lcls + BCEcls(ps[:, 6:], t)   # the reference passes ps[:, 5:] here (shape bug, loss.py:331-333;
                #                                      unreachable with the shipped nc=1 cfgs) -- fixed to the class logits
        if 'default' in arc:
            lobj = lobj + BCEobj(pi[..., 5], tobj)                                  # Q6: all cells
        elif 'BCE' in arc:
            t = torch.zeros_like(pi[..., 6:])
            if nb:
                t[b, a, gj, gi, tcls[i]] = 1.0
            lobj = lobj + BCE(pi[..., 6:], t)
        elif 'CE' in arc:
            t = torch.zeros_like(pi[..., 0], dtype=torch.long)
            if nb:
                t[b, a, gj, gi] = tcls[i] + 1
            lcls = lcls + CE(pi[..., 5:].reshape(-1, core.nc + 1), t.view(-1))

    lobj = lobj * h['obj']
    lcls = lcls * h['cls']
    lreg = lreg * h['reg']
    loss = lobj + lcls + lreg
    return loss, torch.cat((lobj, lcls, lreg, loss)).detach()
