"""Fixed-shape, host-sync-free formulation of build_targets + compute_loss.

Same numbers as the mirror in loss.py (and therefore as the reference's model/loss.py:161-367 -- every quirk Q1-Q8
listed there is kept), but written over PADDED targets with a validity mask instead of boolean compaction:

    tpad, valid = pad_targets(targets, capacity)                    # [capacity, 7], [capacity] bool
    loss, items = compute_loss_static(p, tpad, valid, model, hyp)   # loss[1], items[4] = (lobj, lcls, lreg, loss)

Every tensor has a shape that depends only on (capacity, heads), nothing is read back to the host (the reference
syncs on `any()`, `.max()`, `.sum()` asserts at loss.py:221/:236/:248 and loops over orphan targets in Python), so the
forward AND backward of the loss can be captured in a hipGraph and replayed (model/train_engine.py, FusedTrainStep).
The candidate set is the full [na, capacity] anchor x target grid per head (anchor-major, the reference's order);
a weight of 0/1 replaces row selection, means become sum(w * l) / max(sum(w), 1).

Supported by the TENSOR formulation (compute_loss_static): arc 'default' / 'defaultpw' with any nc; other arcs raise
NotImplementedError there.  The HIP implementation behind FusedLoss (csrc/loss.hip) covers every arc of the reference:
default, uBCE, uCE, each with or without the focal wrappers ('F').
The reference's two assertions (class index range, "something wrong at target building") are host syncs and are not
evaluated here; loss.compute_loss keeps them.
"""
import math

import os

import torch
import torch.nn.functional as F

from .loss import _core


def pad_targets(targets, capacity):
    """targets [nt, 7] -> ([capacity, 7] zero padded, [capacity] bool).  nt > capacity is an error (never truncates)."""
    nt = int(targets.shape[0])
    if nt > capacity:
        raise ValueError("pad_targets: %d targets exceed the capacity %d" % (nt, capacity))
    tpad = torch.zeros(capacity, 7, dtype=torch.float32, device=targets.device)
    valid = torch.zeros(capacity, dtype=torch.bool, device=targets.device)
    if nt:
        tpad[:nt] = targets
        valid[:nt] = True
    return tpad, valid


def _smooth_l1(a, b):
    d = (a - b).abs()
    return torch.where(d < 1.0, 0.5 * d * d, d - 0.5)


def build_targets_static(model, tpad, valid, hyp):
    """Per head: dict(w [na,NT] 0/1 weights, b/gj/gi [NT] long, gxy [NT,2] cell offsets, gwh [NT,2], ga [NT], cls [NT] long).
    Candidate (a, t) of head i is a positive iff w[a, t] == 1."""
    core = _core(model)
    h = core.hyp if getattr(core, 'hyp', None) else hyp
    cf = float(hyp['context_factor'])
    dev = tpad.device
    t = tpad.clone()
    heads, ious = [], []
    a_off = None
    for i in core.yolo_layers:
        layer = core.module_list[i]
        ng, av = layer.ng.to(dev), layer.anchor_vec.to(dev)
        # Q7: cumulative, once per head
        t = torch.cat((t[:, :4], (t[:, 4] + t[:, 5] * (cf - 1)).unsqueeze(1), (t[:, 5] * cf).unsqueeze(1), t[:, 6:]), 1)
        gwh = t[:, 4:6] * ng
        ga = t[:, 6]
        aw, ah = av[:, 0:1], av[:, 1:2]
        gw, gh = gwh[:, 0].unsqueeze(0), gwh[:, 1].unsqueeze(0)
        inter = torch.min(aw, gw) * torch.min(ah, gh)
        iou = inter / ((aw * ah + 1e-16) + gw * gh - inter)                     # [na, NT]
        gxy = t[:, 2:4] * ng
        gij = gxy.long()
        heads.append(dict(b=t[:, 0].long(), cls=t[:, 1].long(), gi=gij[:, 0], gj=gij[:, 1], gxy=gxy - gxy.floor(),
                          gwh=gwh, ga=ga, av=av))
        ious.append(iou)
        # Q2: the angle gate uses the last head's values (identical across heads)
        a_off = (ga.unsqueeze(0) - av[:, 2:3]).abs()
        a_off = torch.where(a_off > 0.5 * math.pi, math.pi - a_off, a_off)
    na = ious[0].shape[0]
    j_a = (a_off < h['ang_t'])
    v = valid.unsqueeze(0)
    js = [(iou > h['iou_t']) & j_a & v for iou in ious]
    covered = torch.stack([j.any(0) for j in js], 0).any(0)                      # [NT]
    orphan = valid & ~covered
    # best-anchor fallback (loss.py:233-242): over the head-major concatenation of the IoUs, the FIRST maximum picks the
    # head (Q3: floor division), the tie (the angles of one w,h anchor share an IoU) with the smallest angle offset
    # picks the anchor
    cat = torch.cat(ious, 0)                                                     # [n_heads * na, NT]
    m = cat.max(0)[0]
    tie = cat == m.unsqueeze(0)
    first = tie.int().argmax(0)
    layer_id = torch.div(first, na, rounding_mode='floor')
    ao = a_off.repeat(len(ious), 1)
    pick = torch.where(tie, ao, torch.full_like(ao, float('inf'))).argmin(0)
    anchor = pick % na
    cols = torch.arange(tpad.shape[0], device=dev)
    for lid, j in enumerate(js):
        add = torch.zeros_like(j)
        add[anchor, cols] = orphan & (layer_id == lid)
        heads[lid]['w'] = (j | add).float()
    return heads


def compute_loss_static(p, tpad, valid, model, hyp, pos_weights=None):
    """pos_weights: optional (cls_pw[1], obj_pw[1]) device tensors made ahead of time (a graph capture cannot create
    tensors from host data)."""
    core = _core(model)
    h = core.hyp if getattr(core, 'hyp', None) else hyp
    arc = core.arc
    if 'default' not in arc or 'F' in arc:
        raise NotImplementedError("compute_loss_static supports the 'default' arcs; got %r" % (arc,))
    dev = p[0].device
    heads = build_targets_static(model, tpad, valid, hyp)
    zero = torch.zeros(1, dtype=torch.float32, device=dev)
    lcls, lobj, lreg = zero.clone(), zero.clone(), zero.clone()
    if pos_weights is None:
        pos_weights = (torch.tensor([h['cls_pw']], dtype=torch.float32, device=dev),
                       torch.tensor([h['obj_pw']], dtype=torch.float32, device=dev))
    cls_pw, obj_pw = pos_weights
    for i, pi in enumerate(p):
        hd = heads[i]
        w = hd['w']                                                              # [na, NT]
        na, NT = w.shape
        n = w.sum()
        nsafe = n.clamp(min=1.0)
        has = (n > 0).float()
        ai = torch.arange(na, device=dev).unsqueeze(1).expand(na, NT)
        b, gj, gi = (hd[k].unsqueeze(0).expand(na, NT) for k in ('b', 'gj', 'gi'))
        ps = pi[b, ai, gj, gi]                                                   # [na, NT, no]
        av = hd['av']
        pxy = torch.sigmoid(ps[..., 0:2])
        pwh = torch.exp(ps[..., 2:4]).clamp(max=1E3) * av[:, :2].unsqueeze(1)    # Q8
        pa = torch.atan(ps[..., 4]) + av[:, 2:3]
        txy = hd['gxy'].unsqueeze(0)
        twh = hd['gwh'].unsqueeze(0)
        ta = hd['ga'].unsqueeze(0)
        inter = torch.min(twh[..., 0], pwh[..., 0]) * torch.min(twh[..., 1], pwh[..., 1])
        union = (twh[..., 0] * twh[..., 1] + 1e-16) + pwh[..., 0] * pwh[..., 1] - inter
        liou = (w * (1.0 - inter / union)).sum() / nsafe                         # Q1: each head adds its own
        sm_xy = (w.unsqueeze(-1) * _smooth_l1(pxy, txy)).sum() / (2.0 * nsafe)
        sm_a = (w * _smooth_l1(pa, ta)).sum() / nsafe
        lreg = lreg + has * (sm_xy + 2 * sm_a + liou * h['giou'])
        if core.nc > 1:
            onehot = (hd['cls'].unsqueeze(-1) == torch.arange(core.nc, device=dev)).float().unsqueeze(0).expand(na, NT, core.nc)
            bce = F.binary_cross_entropy_with_logits(ps[..., 6:], onehot, pos_weight=cls_pw, reduction='none')
            lcls = lcls + has * (w.unsqueeze(-1) * bce).sum() / (nsafe * core.nc)
        tobj = torch.zeros_like(pi[..., 0])
        tobj.index_put_((b.reshape(-1), ai.reshape(-1), gj.reshape(-1), gi.reshape(-1)), w.reshape(-1), accumulate=True)
        tobj = tobj.clamp(max=1.0)
        lobj = lobj + F.binary_cross_entropy_with_logits(pi[..., 5], tobj, pos_weight=obj_pw)      # Q6: all cells
    lobj = lobj * h['obj']
    lcls = lcls * h['cls']
    lreg = lreg * h['reg']
    loss = lobj + lcls + lreg
    return loss, torch.cat((lobj, lcls, lreg, loss)).detach()


# ------------------------------------------------------------------------------------------------ graph-captured loss
class _FusedLossFn(torch.autograd.Function):
    """loss = f(p) whose value and gradient were already produced by the captured graph."""

    @staticmethod
    def forward(ctx, fused, *p):
        ctx.fused = fused
        ctx.state = fused._last_state    # the engine whose head-gradient buffers hold THIS call's gradient ...
        ctx.stamp = ctx.state['stamp']   # ... until the next call ON THAT ENGINE overwrites them (a second engine -- another input shape --
                                         # has its own buffers and its own counter: ADVICE r3)
        ctx.used = False
        return fused.static_loss.clone()

    @staticmethod
    def backward(ctx, g):
        if ctx.stamp != ctx.state['stamp']:
            raise RuntimeError("fused compute_loss: backward() of a loss whose head gradients were overwritten by a later "
                               "compute_loss call on the same engine (call backward before computing the next loss)")
        if ctx.used:
            raise RuntimeError("fused compute_loss: second backward() through the same loss (the head-gradient buffers were "
                               "scaled in place by the first; retain_graph is not supported on the fused path)")
        ctx.used = True
        pg = ctx.fused.pg
        heads = ctx.fused.head_grads
        if heads is not None:
            # the NHWC loss kernel already wrote d loss / d head into the engine's head-gradient buffers (for an upstream
            # gradient of 1); the fp32 buffers returned here are only the tokens the engine recognises them by
            from . import hip_train_ops as tr
            g1 = g.detach().reshape(1).float()
            for hg in heads:
                tr.scale_bf16_if(g1, hg)
        else:
            for buf in pg:
                buf.mul_(g)
        return (None,) + tuple(pg)


class FusedLoss(object):
    """compute_loss for heads produced by the HIP TrainEngine, as ONE hipGraph replay: padded targets in, loss items and
    d loss / d p out, written straight into the engine's head-gradient buffers.  Installed by Darknet.enable_fused_loss();
    loss.compute_loss dispatches here when it can (engine heads, supported arc, nt <= capacity) and falls back to the
    eager mirror otherwise, so the caller's code (`loss, items = compute_loss(...); loss.backward()`) does not change."""

    def __init__(self, model, capacity=512, impl='hip'):
        """impl 'hip': csrc/loss.hip kernels (no autograd at all); 'torch': compute_loss_static under autograd."""
        self.model = model
        self.impl = impl
        self.capacity = int(capacity)      # per-engine capture state lives on the engine (eng._fused_state)
        self._last_state = None            # the per-engine state of the latest call (its 'stamp' counts that engine's calls: a loss's
                                           # backward must run before the next call on the SAME engine, see _FusedLossFn)

    def _engine_of(self, p):
        from .train_engine import TrainEngine
        for eng in getattr(self.model, '_engines', {}).values():
            if isinstance(eng, TrainEngine) and len(eng.p) == len(p) and all(
                    a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.dtype == b.dtype for a, b in zip(eng.p, p)):
                return eng
        return None

    def try_call(self, p, targets, hyp):
        core = self.model
        arc = core.arc
        plain = 'default' in arc and 'F' not in arc
        if (not plain and self.impl != 'hip') or targets.shape[0] > self.capacity or not p[0].is_cuda:
            return None                     # (the tensor formulation covers the default arcs only; csrc/loss.hip covers all of them)
        if not plain and p[0].shape[-1] > 32:
            return None                     # the per-cell pass of the non-default arcs keeps a cell's logits in registers
        eng = self._engine_of(p)
        # an engine that launches eagerly because a capture failed or because RYOLO_NO_GRAPH=1 asked for it keeps the HIP loss kernels (launched
        # eagerly too); an engine BUILT without graphs (use_graph=False: tests, debugging) takes the eager mirror
        eager_engine = eng is not None and not eng.use_graph and (eng.graph_fallback is not None or getattr(eng, 'no_graph_env', False))
        if eng is None or not (eng.use_graph or eager_engine):
            return None
        h = core.hyp if getattr(core, 'hyp', None) else hyp
        if h.get('riou', 0) and self.impl != 'hip':
            return None                     # the tensor formulation has no rotated IoU: eager mirror (RotatedIoU autograd function)
        key = tuple(float(h[k]) for k in ('giou', 'cls', 'cls_pw', 'obj', 'obj_pw', 'iou_t', 'ang_t', 'reg')) + (
            float(hyp['context_factor']), float(h.get('riou', 0)), str(arc), float(h.get('fl_gamma', 0.0)))
        st = getattr(eng, '_fused_state', None)
        if st is None or st['key'] != key:
            st = self._make_state(eng, hyp, key)
            eng._fused_state = st
        dev = p[0].device
        nt = int(targets.shape[0])
        with torch.no_grad():
            st['t'].zero_()
            st['valid'].zero_()
            if nt:
                st['t'][:nt].copy_(targets)
                st['valid'][:nt].fill_(True)
        self.pg, self.static_loss = st['pg'], st['loss']
        self.head_grads = st.get('head_g') if self.impl == 'hip' else None
        if st['calls'] < 2 or st.get('no_graph') or eager_engine:   # eager: lazy allocations, autograd warm-up (or a failed capture, or asked for)
            self._body(st)
        else:
            if st['graph'] is None:
                torch.cuda.synchronize(dev)
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):
                        self._body(st)
                    st['graph'] = g
                except Exception as e:      # noqa: BLE001  a runtime that refuses the capture: eager launches from here on
                    st['no_graph'] = True
                    eng.graph_fallback = "loss: %s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
                    import warnings
                    warnings.warn("fused loss: hipGraph capture failed (%s); continuing with eager launches" % eng.graph_fallback)
                    try:
                        torch.cuda.synchronize(dev)
                    except Exception:       # noqa: BLE001
                        pass
                    self._body(st)
            if st['graph'] is not None:
                st['graph'].replay()
        st['calls'] += 1
        st['stamp'] = st.get('stamp', 0) + 1
        self._last_state = st
        if st.get('head_g') is not None:
            eng.head_g_ready = True        # (python side effect: must not live in the captured body)
        loss = _FusedLossFn.apply(self, *p)
        return loss, torch.cat((st['items'][:3], loss.detach()))

    def _make_state(self, eng, hyp, key):
        dev = eng.device
        if not hasattr(eng, "static_pg"):
            eng.static_pg = [torch.zeros_like(q) for q in eng.p]
        h = self.model.hyp if getattr(self.model, 'hyp', None) else hyp
        pairs = getattr(eng, 'head_pairs', None)
        nhwc = self.impl == 'hip' and pairs is not None and all(
            (hd.shape[-1] % 8 == 0 and q.shape[1] * q.shape[4] == hd.shape[-1]) for (hd, _), q in zip(pairs, eng.p))
        if nhwc:
            eng.fused_nhwc = True          # the engine's fp32 head-gradient buffers become all-zero scratch (see backward())
            for buf in eng.static_pg:
                buf.zero_()
        return dict(key=key, hyp=dict(hyp), calls=0, graph=None, pg=eng.static_pg, eng=eng,
                    head=[a for a, _ in pairs] if nhwc else None, head_g=[b for _, b in pairs] if nhwc else None,
                    leaves=[q.detach().requires_grad_(True) for q in eng.p],
                    t=torch.zeros(self.capacity, 7, device=dev), valid=torch.zeros(self.capacity, dtype=torch.bool, device=dev),
                    loss=torch.zeros(1, device=dev), items=torch.zeros(4, device=dev),
                    pw=(torch.tensor([h['cls_pw']], dtype=torch.float32, device=dev),
                        torch.tensor([h['obj_pw']], dtype=torch.float32, device=dev)))

    def _body(self, st):
        if self.impl == 'hip':
            return self._body_hip(st)
        with torch.enable_grad():
            loss, items = compute_loss_static(st['leaves'], st['t'], st['valid'], self.model, st['hyp'], pos_weights=st['pw'])
            grads = torch.autograd.grad(loss, st['leaves'])
        with torch.no_grad():
            for buf, g in zip(st['pg'], grads):
                buf.copy_(g)
            st['loss'].copy_(loss.detach())
            st['items'].copy_(items)

    def _body_hip(self, st):
        from . import hip_train_ops as tr
        core = self.model
        h = core.hyp if getattr(core, 'hyp', None) else st['hyp']
        with torch.no_grad():
            if 'bt' not in st:
                st['bt'] = tr.BuildTargets(core, self.capacity, st['t'].device)
                st['valid_u8'] = torch.zeros(self.capacity, dtype=torch.uint8, device=st['t'].device)
            st['valid_u8'].copy_(st['valid'])
            st['bt'].run(st['t'], st['valid_u8'], h, st['hyp']['context_factor'])      # one launch for every head
            heads = st['bt'].heads()
            arc = tr.arc_flags(core.arc)
            if 'bitmaps' not in st:
                st['bitmaps'] = [tr.yolo_loss_bitmap(q, core.nc, arc) for q in st['leaves']]
            st['items'].zero_()
            for k, (q, hd, bm, dp) in enumerate(zip(st['leaves'], heads, st['bitmaps'], st['pg'])):
                bm.zero_()
                if st.get('head_g') is not None:
                    tr.yolo_loss_head_nhwc(st['head'][k], q.detach(), hd, core.nc, h, bm, dp, st['head_g'][k], st['items'], arc)
                else:
                    tr.yolo_loss_head(q.detach(), hd, core.nc, h, bm, dp, st['items'], arc)

            st['items'][3:4].copy_(st['items'][:3].sum(0, keepdim=True))
            st['loss'].copy_(st['items'][3:4])
