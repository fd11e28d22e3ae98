"""TrainEngine -- forward + backward of the Darknet cfg graph in TRAINING mode on the hand-written HIP kernels.

Replaces, for `model.train(); pred = model(imgs); loss.backward()` (train.py:268-282), the autograd walk over
nn.Conv2d / nn.BatchNorm2d (batch statistics, per replica) / nn.PReLU / shortcut / route / upsample that the reference
dispatches to cuDNN/ATen.  The engine returns the head tensors `p` as outputs of an autograd Function and receives
dL/dp back -- from the eager loss mirror (model/loss.py) or, with Darknet.enable_fused_loss(), from the graph-captured
HIP loss (model/loss_static.py, csrc/loss.hip), which writes the engine's head-gradient buffers directly.

Forward per `convolutional` block:   z = conv(x, W) (+ per-channel sums of z, z^2 in the conv epilogue)
                                     -> bn_finalize (mean, invstd, running stats, folded scale/shift)
                                     -> y = PReLU(z*scale + shift) [+ shortcut source]            (bn_act_fwd)
Backward per block:                  dy -> bn_act_bwd -> dz, dgamma, dbeta, dslope
                                     dW += wgrad(x, dz);   dx (+)= dgrad(dz, W)
Saved for backward: x (the producer's output, never overwritten), z, the four per-channel statistics.
Gradient buffers mirror the activation buffers (same concat/slice structure); the FIRST contribution to a gradient
view overwrites it and later ones accumulate (decided statically at plan time, no memset of the 6 GB of gradients).
Residual chains share ONE gradient buffer per stage (the skip branch receives dy unchanged, so no copy).
Parameter gradients accumulate in one flat fp32 buffer with stable addresses and are added to `param.grad` (possibly
views of the data-parallel buckets of rotate-yolov3_amd/dist.py) segment by segment: the backward launch list is cut
where a bucket has received its last gradient and the reducer's hooks run there, so buckets travel during backward.
Forward and each backward segment are replayed from hipGraphs after two eager steps.  Packed bf16 weight images
(forward layout and every dgrad class) are rebuilt from the fp32 parameters each step by ONE launch.
"""
import ctypes as C
import os

import torch
import torch.nn as nn

from .. import _lib
from . import hip_ops as ops
from . import hip_train_ops as tr
from .engine import HipEngine, _abs


class _Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, x, anchor):
        ctx.engine = engine
        outs = engine.forward(x)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        ctx.engine.backward(grads)
        return None, None, None


class TrainEngine(object):
    def __init__(self, model, x_shape, device, use_graph=True):
        _lib.lib()
        self.model = model
        # RYOLO_NO_GRAPH=1 (bench.py --no-graph): eager launches from the start -- the switch a launcher can fall back to when a runtime
        # refuses stream capture under a multi-rank communicator (a failed capture also falls back by itself, see _capture)
        self.no_graph_env = os.environ.get('RYOLO_NO_GRAPH', '0') == '1'
        self.use_graph = use_graph and not self.no_graph_env
        self.graph_fallback = None      # set when a hipGraph capture failed and the engine went back to eager launches
        self.reduce_fallback = None     # set when the batched split-K reduce's per-layer workspaces did not fit in memory
        self.force_eager = False        # measurement: launch eagerly although the graphs exist (bench.py's traced steps)
        self.g_fwd = self.g_bwd = None
        self._segs = None
        self._red_planned = False
        self.nbt = None
        self.packs = None
        self.steps = 0
        self.device = device
        self.bs, cin, self.H, self.W = [int(v) for v in x_shape]
        defs, mods = model.module_defs, model.module_list
        n = len(defs)
        self.defs, self.mods = defs, mods

        # ---- shapes / readers / shortcut fusion / concat homes: same planning rules as the inference engine
        shp = []
        c, h, w = cin, self.H, self.W
        for i, d in enumerate(defs):
            t = d['type']
            if t == 'convolutional':
                conv = HipEngine._conv_of(mods[i])
                k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
                c, h, w = conv.out_channels, (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
            elif t == 'upsample':
                s = int(d['stride'])
                if s != 2:
                    raise RuntimeError("training path: only x2 upsampling")
                h, w = h * s, w * s
            elif t == 'route':
                ls = [_abs(i, int(v)) for v in d['layers'].split(',')]
                c = sum(shp[l][0] for l in ls)
                h, w = shp[ls[0]][1], shp[ls[0]][2]
            elif t == 'maxpool':
                raise RuntimeError("training path: maxpool graphs (yolov3-tiny) cannot train in the reference either "
                                   "(model/loss.py:248 hard-codes three heads)")
            shp.append((c, h, w))
        self.shp = shp
        readers = [[] for _ in range(n)]
        for i, d in enumerate(defs):
            if d['type'] == 'route':
                for v in d['layers'].split(','):
                    readers[_abs(i, int(v))].append(i)
            else:
                if i > 0:
                    readers[i - 1].append(i)
                if d['type'] == 'shortcut':
                    readers[_abs(i, int(d['from']))].append(i)
        fused_into, conv_res = {}, {}
        for i, d in enumerate(defs):
            if d['type'] == 'shortcut' and i > 0 and defs[i - 1]['type'] == 'convolutional' and readers[i - 1] == [i] \
                    and _abs(i, int(d['from'])) != i - 1 and HipEngine._bn_of(mods[i - 1]) is not None:
                fused_into[i] = i - 1
                conv_res[i - 1] = _abs(i, int(d['from']))

        def new_pair(c, h, w):
            a = torch.empty((self.bs, h, w, c), dtype=torch.bfloat16, device=device)
            return a, torch.empty_like(a)

        act = [None] * n     # activation view per layer
        grd = [None] * n     # gradient view per layer (same structure)
        home = {}
        parent_of = {}
        alias = {}
        for i, d in enumerate(defs):
            if d['type'] != 'route':
                continue
            ls = [_abs(i, int(v)) for v in d['layers'].split(',')]
            if len(ls) == 1:
                alias[i] = ls[0]
                continue
            c, h, w = shp[i]
            a, g = new_pair(c, h, w)
            act[i], grd[i] = a, g
            off = 0
            for l in ls:
                src = l
                while src in alias:
                    src = alias[src]
                if src in home or defs[src]['type'] not in ('convolutional', 'shortcut', 'upsample') or shp[src][0] % 8 or off % 8:
                    raise RuntimeError("training path: route %d needs a copy (unsupported graph)" % i)
                home[src] = (a[..., off:off + shp[src][0]], g[..., off:off + shp[src][0]])
                parent_of[src] = (g.data_ptr(), tuple(g.shape), tuple(g.stride()))
                off += shp[l][0]

        def pair_for(i):
            if i in home:
                return home[i]
            return new_pair(*shp[i])

        self.x_nhwc = torch.empty((self.bs, self.H, self.W, 8), dtype=torch.bfloat16, device=device)
        cmax = max(ops.cpad(HipEngine._conv_of(m).out_channels) for d, m in zip(defs, mods) if d['type'] == 'convolutional')
        cmax = max(cmax, max(ops.cpad(HipEngine._conv_of(m).in_channels) for d, m in zip(defs, mods) if d['type'] == 'convolutional'))
        self.ones = torch.ones(cmax, device=device)
        self.zeros = torch.zeros(cmax, device=device)
        self.stat_part = torch.zeros((tr.stat_rows(), 2, cmax), dtype=torch.float64, device=device)
        self.blocks = []                     # per conv: dict of tensors / modules
        self.p, self.p_src = [], []
        self.static_grad = {}
        self.direct = False
        self.static_flat = None
        init = set()                         # gradient views that already received their first contribution
        children = {}                        # concat gradient buffer -> keys of the slices that live inside it

        def key(t):
            return (t.data_ptr(), tuple(t.shape), tuple(t.stride()))

        def first(t):
            k = key(t)
            if k in init:
                return False
            init.add(k)
            for ck in children.get(k, ()):      # writing a whole concat gradient initialises its channel slices
                init.add(ck)
            return True

        for src, (a_s, g_s) in home.items():
            children.setdefault(parent_of[src], []).append(key(g_s))
        wgrad_ws, bn_ws = 0, 0
        plan = []    # (kind, layer, payload) in forward order
        for i, d in enumerate(defs):
            t = d['type']
            if i in fused_into:
                act[i], grd[i] = act[fused_into[i]], grd[fused_into[i]]
                continue
            if t == 'convolutional':
                conv = HipEngine._conv_of(mods[i])
                bn = HipEngine._bn_of(mods[i])
                actmod = None
                is_mish = False
                for s_ in mods[i]:
                    if isinstance(s_, (nn.PReLU, nn.LeakyReLU)):
                        actmod = s_
                    elif type(s_).__name__ == 'Mish':
                        is_mish = True
                    elif not isinstance(s_, (nn.Conv2d, nn.BatchNorm2d)):
                        # Swish etc. have no kernels here: refuse instead of training them as linear
                        raise RuntimeError("activation %s has no HIP training kernels (use model.backend = 'torch')"
                                           % type(s_).__name__)
                xin = self.x_nhwc if i == 0 else act[i - 1]
                xin_g = None if i == 0 else grd[i - 1]
                k, s, pad = conv.kernel_size[0], conv.stride[0], conv.padding[0]
                final = i + 1 if i in conv_res else i
                y, dy = pair_for(final)
                # residual chain: the gradient of this block's output and of its skip source are the same tensor
                # (d(x + f(x)) passes dy to the skip branch unchanged) -- share ONE buffer instead of copying dy into the
                # source's gradient: the block reads dy before the branch's dgrad accumulates onto it (launch order)
                res_alias = False
                if i in conv_res and final not in home:
                    rg = grd[conv_res[i]]
                    if rg is not None and tuple(rg.shape) == tuple(dy.shape) and rg.is_contiguous():
                        dy, res_alias = rg, True
                act[i], grd[i] = y, dy
                c, h, w = shp[i]
                desc = tr.make_desc(xin, c, k, s, pad)
                # layer 0 trains without its conv output: z0 (4 x the input, 27 MACs per value) is recomputed in the BatchNorm
                # passes instead of being stored and re-read (include/ryolo.h: ryolo_conv0_*)
                recompute = (i == 0 and bn is not None and i not in conv_res and tr.conv0_recompute_supported(desc)
                             and os.environ.get('RYOLO_CONV0_RECOMPUTE', '1') != '0')
                if bn is not None:
                    z = None if recompute else torch.empty((self.bs, h, w, c), dtype=torch.bfloat16, device=device)
                    # (layer 0's one-pass backward never materialises dz: 1.5 GB at bs 64 / 608^2)
                    one_pass = recompute and xin_g is None and os.environ.get('RYOLO_CONV0_ONE_PASS', '1') != '0'
                    dz = None if one_pass else torch.empty((self.bs, h, w, c), dtype=torch.bfloat16, device=device)
                else:
                    z, dz = y, dy            # linear bias conv: y IS z, dz IS dy
                blk = dict(i=i, conv=conv, bn=bn, act=actmod, mish=is_mish, xin=xin, xin_g=xin_g, z=z, dz=dz, y=y, dy=dy, desc=desc,
                           res=act[conv_res[i]] if i in conv_res else None,
                           res_g=grd[conv_res[i]] if i in conv_res else None, res_alias=res_alias, cin_k=xin.shape[-1], k=k,
                           s=s, pad=pad,
                           npix=self.bs * h * w, C=c, recompute=recompute)
                wgrad_ws = max(wgrad_ws, tr.wgrad_ws_bytes(desc))
                bn_ws = max(bn_ws, tr.bn_bwd_ws_bytes(blk['npix'], c))
                self.blocks.append(blk)
                plan.append(('conv', i, blk))
            elif t == 'shortcut':
                a, b = i - 1, _abs(i, int(d['from']))
                y, dy = pair_for(i)
                act[i], grd[i] = y, dy
                plan.append(('add', i, (act[a], act[b], y, grd[a], grd[b], dy)))
            elif t == 'upsample':
                y, dy = pair_for(i)
                act[i], grd[i] = y, dy
                plan.append(('up', i, (act[i - 1], y, grd[i - 1], dy)))
            elif t == 'route':
                if i in alias:
                    act[i], grd[i] = act[alias[i]], grd[alias[i]]
            elif t == 'yolo':
                m = mods[i]
                c, h, w = shp[i]
                anchors = m.anchors.to(device=device, dtype=torch.float32).contiguous()
                pbuf = torch.empty((self.bs, m.na, h, w, model.nc + 6), dtype=torch.float32, device=device)
                # training returns the raw heads only (models.py:189-194): the decoded rows are not written -- the p-only form
                # of ryolo_yolo_decode needs the tiled kernel (no <= 96, na*no <= 1024); otherwise keep a scratch `io`
                io = None
                if model.nc + 6 > 96 or m.na * (model.nc + 6) > 1024:
                    io = torch.empty((self.bs, m.na * h * w, model.nc + 6), dtype=torch.float32, device=device)
                self.p.append(pbuf)
                plan.append(('yolo', i, (act[i - 1], grd[i - 1], m, anchors, pbuf, io, h, w)))
                act[i], grd[i] = act[i - 1], grd[i - 1]
        self.plan = plan
        # (head activation, head gradient) NHWC bf16 pairs in the order of self.p: the fused loss writes the gradients directly
        self.head_pairs = [(pl[0], pl[1]) for kind, _, pl in plan if kind == 'yolo']
        self.fused_nhwc = False
        self.head_g_ready = False
        # (weight gradients on a second stream -- a parallel branch of the backward graphs -- were measured in rounds 2 and 5 and lost both
        #  times: 63.2 vs 62.6 ms, 50.07 vs 49.57 ms; wgrad_wide's two workgroups per CU fill the register file, nothing co-resides with it.
        #  The switch and its code were removed in round 6; profiles/r05_ab_log.txt keeps the numbers.)
        # round 5: the split-K reduces of a backward segment as ONE launch -- every layer keeps its partial tiles in its own workspace (3.2 GB
        # at bs 64 / 608^2) and the segment ends with ryolo_conv_wgrad_reduce_batch; bit-identical gradients.  RYOLO_WGRAD_BATCH_REDUCE=0: one
        # reduce launch behind every weight gradient, sharing one workspace (rounds 2-4).
        self.batch_reduce = os.environ.get('RYOLO_WGRAD_BATCH_REDUCE', '1') != '0'
        self._wr_batches = {}
        # shared split-K workspace of the per-layer reduce path: allocated on first use (the batched path gives every layer its own)
        self._ws_w_bytes = max(wgrad_ws, 256)
        self._ws_w = None
        self.ws_b = torch.empty(max(bn_ws, 256), dtype=torch.uint8, device=device)

        # ---- static accumulate flags for the backward pass (reverse order; first contribution overwrites)
        self.bplan = []
        for kind, i, pl in reversed(plan):
            if kind == 'yolo':
                head, head_g = pl[0], pl[1]
                hidx = [k for k, q in enumerate(self.p) if q is pl[4]][0]
                first(head_g)
                self.bplan.append(('yolo', i, pl, hidx))               # always the first (sole) writer of the head grad
            elif kind == 'conv':
                blk = pl
                res_first = first(blk['res_g']) if (blk['res_g'] is not None and not blk['res_alias']) else None
                in_first = first(blk['xin_g']) if blk['xin_g'] is not None else None
                self.bplan.append(('conv', i, blk, (res_first, in_first)))
            elif kind == 'add':
                a_g, b_g = pl[3], pl[4]
                self.bplan.append(('add', i, pl, (first(a_g), first(b_g))))
            elif kind == 'up':
                self.bplan.append(('up', i, pl, first(pl[2])))

    # ------------------------------------------------------------------ parameter gradients
    # The backward kernels ACCUMULATE (+=) into fp32 buffers with STABLE addresses (hipGraph replays write to the same
    # pointers every step).  Two sinks:
    #   own     engine-owned flat buffer, zeroed at the start of every backward and added to param.grad after the replay
    #           (param.grad may be None, a fresh tensor, anything the user does to it);
    #   direct  a data-parallel reducer is attached (dist.GradientAllReducer: param.grad IS a view of a flat bucket the
    #           reducer owns): the kernels accumulate straight into those views -- no per-step memset of a second 250-MB
    #           buffer and no 750-MB add pass.  The views' addresses are checked before every backward; if the user
    #           replaced a param.grad (e.g. zero_grad(set_to_none=True)) it is RE-BOUND to its bucket view -- zeroed when it
    #           was None, the replacement's values copied in otherwise (_check_direct_sink, _flush_param_grads): the sink,
    #           the captured graphs and the reducer's buckets stay as they are, and the all-reduce still sends this gradient.
    def _build_grad_sink(self):
        plist = [q for q in self.model.parameters()]
        views = getattr(self.model, '_dp_grad_views', None)
        ok = views is not None and os.environ.get("RYOLO_DIRECT_GRADS", "1") != "0"
        if ok:
            for q in plist:
                v = views.get(q)
                if (v is None or v.dtype != torch.float32 or not v.is_contiguous() or v.device != q.device or v.shape != q.shape
                        or q.grad is None or q.grad.data_ptr() != v.data_ptr()):
                    ok = False
                    break
        self.direct = bool(ok)
        if ok:
            self.static_flat = None
            self.static_grad = {q: views[q] for q in plist}          # (holds the buckets' storage alive for the graphs)
        else:
            self.static_flat = torch.zeros(sum(q.numel() for q in plist), dtype=torch.float32, device=self.device)
            off = 0
            for q in plist:
                self.static_grad[q] = self.static_flat[off:off + q.numel()].view_as(q)
                off += q.numel()

    def _reset_grad_sink(self):
        """Forget the sink and every graph that holds its addresses (a reducer was attached / param.grad was replaced)."""
        self.static_grad, self.static_flat, self.direct = {}, None, False
        self._segs, self.g_bwd = None, None
        self._wr_batches = {}            # the batched reduces hold the sink's addresses

    def _check_direct_sink(self):
        """A reducer is attached and param.grad must BE the reducer's bucket view: its hooks all-reduce the buckets, so a gradient
        that lives anywhere else would be stepped on un-averaged (ranks diverge silently).  When the user replaced a param.grad --
        optimizer.zero_grad() sets it to None by default -- the view is put back: None means "zeroed" (the view is cleared), another
        tensor's values are copied in.  The sink, the graphs and the buckets stay as they are."""
        if not self.direct:
            return
        rebound = 0
        for q, v in self.static_grad.items():
            g = q.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr():
                v.copy_(g)
            else:
                continue
            q.grad = v
            rebound += 1
        if rebound and not getattr(self, '_warned_rebind', False):
            self._warned_rebind = True
            import warnings
            warnings.warn("TrainEngine: %d param.grad tensors had been replaced (zero_grad(set_to_none=True)?) while a GradientAllReducer "
                          "owns them; they were re-bound to the reducer's buckets.  Use reducer.zero_grad() to clear gradients." % rebound)

    def _grad_of(self, p):
        g = self.static_grad.get(p)
        if g is None:
            self._build_grad_sink()
            g = self.static_grad[p]
        return g

    def _flush_param_grads(self, params=None):
        """Add the engine's gradient buffers of `params` (default: all) into param.grad (own sink only: the direct sink IS
        param.grad), then run the parameters' post-accumulate-grad hooks (a data-parallel reducer launches a bucket's
        all-reduce from them)."""
        add_to, add_from = [], []
        plist = list(self.static_grad.keys()) if params is None else params
        if not self.direct:
            views = getattr(self.model, '_dp_grad_views', None) or {}
            for p in plist:
                g = self.static_grad[p]
                v = views.get(p)
                if v is not None and (p.grad is None or p.grad.data_ptr() != v.data_ptr()):
                    # a reducer owns this gradient: it must land in the bucket view the all-reduce sends (see _check_direct_sink)
                    if p.grad is None:
                        v.zero_()
                    else:
                        v.copy_(p.grad)
                    p.grad = v
                if p.grad is None:
                    p.grad = g.clone()
                else:
                    add_to.append(p.grad)
                    add_from.append(g)
            if add_to:
                torch._foreach_add_(add_to, add_from)
        for p in plist:
            hooks = getattr(p, '_post_accumulate_grad_hooks', None)
            if hooks:
                for hk in list(hooks.values()):
                    hk(p)

    @staticmethod
    def _has_wgrad(b):
        """False for layer 0 when its whole backward is the one-pass kernel (csrc/conv0_bwd.hip: no dz, no separate weight gradient)"""
        return not (b['bn'] is not None and b['recompute'] and b['xin_g'] is None and b['dz'] is None)

    @property
    def ws_w(self):
        if self._ws_w is None:
            self._ws_w = torch.empty(self._ws_w_bytes, dtype=torch.uint8, device=self.device)
        return self._ws_w

    def _ensure_reduce_batches(self, segs):
        """One job table per backward segment for ryolo_conv_wgrad_reduce_batch: built OUTSIDE any stream capture (the table is uploaded
        from the host) and once per gradient sink (it holds the sink's addresses and the layers' own partial-tile workspaces).
        The per-layer workspaces add up to 3.2 GB at bs 64 / 608^2; when they do not fit, the engine falls back to the shared workspace
        and one reduce launch per layer and says so in `reduce_fallback` (bench.py prints it in launch_mode)."""
        if not self.batch_reduce:
            return
        try:
            self._build_reduce_batches(segs)
        except torch.cuda.OutOfMemoryError as e:
            for kind, i, pl, flags in self.bplan:
                if kind == 'conv':
                    pl.pop('ws_w', None)
            self._wr_batches = {}
            torch.cuda.empty_cache()
            self.batch_reduce = False
            self.reduce_fallback = "per-layer split-K reduces on one shared workspace (the per-layer workspaces did not fit: %s)" % str(e)[:120]
            self.g_bwd = None                 # graphs captured for the batched path (none yet on the first backward) are not reusable

    def _build_reduce_batches(self, segs):
        for lo, hi, _ in segs:
            if (lo, hi) in self._wr_batches:
                continue
            blks = [pl for kind, i, pl, flags in self.bplan[lo:hi] if kind == 'conv' and self._has_wgrad(pl)]
            wb = None
            if blks:
                wb = tr.WgradReduceBatch(self.device)
                for b in blks:
                    if 'ws_w' not in b:
                        b['ws_w'] = torch.empty(max(tr.wgrad_ws_bytes(b['desc']), 256), dtype=torch.uint8, device=self.device)
                    wb.add(b['desc'], b['conv'].in_channels, b['ws_w'], self._grad_of(b['conv'].weight), True)
                wb.finalize()
            self._wr_batches[(lo, hi)] = wb

    def _segments(self):
        """Cut the backward launch list where a data-parallel bucket (model._dp_buckets, lists of parameters, set by
        dist.GradientAllReducer) has received its last gradient, so the bucket's all-reduce over xGMI runs under the
        remaining backward kernels.  Returns [(lo, hi, params)], one entry when there is no reducer."""
        if self._segs is not None:
            return self._segs
        pos = {}
        for k, (kind, i, pl, flags) in enumerate(self.bplan):
            if kind == 'conv':
                for q in list(pl['conv'].parameters()) + (list(pl['bn'].parameters()) if pl['bn'] is not None else []) + (
                        list(pl['act'].parameters()) if isinstance(pl['act'], nn.Module) else []):
                    pos[q] = k
        cuts = set()
        for bucket in getattr(self.model, '_dp_buckets', None) or []:
            ks = [pos[q] for q in bucket if q in pos]
            if ks:
                cuts.add(max(ks) + 1)
        cuts.add(len(self.bplan))
        segs, lo = [], 0
        for hi in sorted(cuts):
            if hi > lo:
                segs.append((lo, hi, [q for q, k in pos.items() if lo <= k < hi]))
                lo = hi
        self._segs = segs
        return segs


    def forward(self, x):
        dev = self.device
        x = x.float().contiguous()
        with torch.cuda.device(dev), torch.no_grad():
            # the input converter runs OUTSIDE the graph, reading the caller's tensor where it lies: the graph starts at the
            # engine's own NHWC buffer, so no 283-MB staging copy of the batch into a graph-owned tensor
            n, c, h, w = x.shape
            _lib.check(_lib.lib().ryolo_nchw_f32_to_nhwc_bf16(x.data_ptr(), n, c, h, w, 8, self.x_nhwc.data_ptr(), _lib.stream_ptr(dev)),
                       "ryolo_nchw_f32_to_nhwc_bf16")
            if not self.use_graph or self.steps < 2 or self.force_eager:       # two eager steps: lazy allocations, one-time attribute calls
                self._forward_launch()
            else:
                if self.g_fwd is None:
                    self.g_fwd = self._capture(self._forward_launch)
                if self.g_fwd is not None:
                    self.g_fwd.replay()
        return [p for p in self.p]

    def _capture(self, launch):
        """One hipGraph of `launch()`.  If the capture fails -- e.g. a runtime that refuses stream capture while a multi-rank RCCL
        communicator has work in flight -- the engine falls back to eager launches for the rest of its life (slower, never wrong),
        runs `launch()` eagerly for this step and returns None; `graph_fallback` says why (bench.py prints it)."""
        dev = self.device
        torch.cuda.synchronize(dev)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):   # RCCL's watchdog thread must not void the capture
                launch()
            return g
        except Exception as e:      # noqa: BLE001  (whatever the runtime raises: the step must still run)
            self.use_graph = False
            self.graph_fallback = "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200] if str(e) else "")
            import warnings
            warnings.warn("TrainEngine: hipGraph capture failed (%s); continuing with eager launches" % self.graph_fallback)
            try:
                torch.cuda.synchronize(dev)
            except Exception:       # noqa: BLE001  (a sticky capture error surfaces here once; the stream is usable afterwards)
                pass
            launch()
            return None

    def _forward_launch(self):
        dev = self.device
        L = _lib.lib()
        if self.nbt is None:
            self.nbt = [pl['bn'].num_batches_tracked for kind, i, pl in self.plan if kind == 'conv' and pl['bn'] is not None]
        if self.nbt:
            torch._foreach_add_(self.nbt, 1)            # one launch for the 72 counters nn.BatchNorm2d would bump
        if self.packs is None:                          # every conv's packed weights (forward + dgrad layouts): ONE launch
            self.packs = tr.WeightPackBatch(dev)
            for kind, i, pl in self.plan:
                if kind != 'conv':
                    continue
                w = pl['conv'].weight
                if w.dtype != torch.float32 or not w.is_contiguous():
                    raise RuntimeError("TrainEngine expects contiguous fp32 conv weights")
                cout, cin, k, _ = w.shape
                pl['packed'] = torch.empty(L.ryolo_conv_packed_weight_bytes(cout, pl['cin_k'], k), dtype=torch.uint8, device=dev)
                pl['packed_d'] = torch.empty(L.ryolo_conv_packed_dgrad_bytes(cout, cin, k, pl['s']), dtype=torch.uint8,
                                             device=dev) if pl['xin_g'] is not None else None
                self.packs.add(w.detach(), pl['s'], pl['cin_k'], pl['packed'], pl['packed_d'])
            self.packs.finalize()
        self.packs.run()
        for kind, i, pl in self.plan:
            if kind == 'conv':
                b = pl
                conv, bn = b['conv'], b['bn']
                if bn is not None:
                    part = tr.conv_fwd_stats(b['desc'], b['xin'], b['packed'], self.ones, self.zeros, b['z'], part=self.stat_part,
                                             clear=False)          # bn_finalize leaves the scratch zeroed (z None: sums only)
                    b['stats'] = tr.bn_finalize(part, b['C'], b['npix'], bn.weight.detach(), bn.bias.detach(), eps=bn.eps,
                                                momentum=bn.momentum, running_mean=bn.running_mean,
                                                running_var=bn.running_var, out=b.get('stats'))
                    slope = b['act'].weight.detach() if isinstance(b['act'], nn.PReLU) else None
                    if isinstance(b['act'], nn.LeakyReLU):
                        if 'leaky' not in b:
                            b['leaky'] = torch.full((1,), b['act'].negative_slope, device=dev)
                        slope = b['leaky']
                    b['slope'] = slope
                    b['actcode'] = 2 if b['mish'] else (1 if slope is not None else 0)
                    if b['recompute']:
                        tr.conv0_bn_act_fwd(b['desc'], b['xin'], b['packed'], b['stats'][2], b['stats'][3], b['actcode'], slope, b['y'])
                    else:
                        tr.bn_act_fwd(b['z'], b['stats'][2], b['stats'][3], b['actcode'], slope, b['y'], residual=b['res'])
                else:
                    if conv.bias is not None:
                        if 'bias_pad' not in b:
                            b['bias_pad'] = torch.zeros(ops.cpad(b['C']), device=dev)
                        b['bias_pad'][:b['C']].copy_(conv.bias.detach())
                        bias = b['bias_pad']
                    else:
                        bias = self.zeros
                    tr.conv_fwd_plain(b['desc'], b['xin'], b['packed'], self.ones, bias, b['z'])
                    b['stats'] = None
            elif kind == 'add':
                a, bb, y = pl[0], pl[1], pl[2]
                nn_, hh, ww, cc = y.shape
                _lib.check(L.ryolo_add_nhwc(a.data_ptr(), a.stride(2), bb.data_ptr(), bb.stride(2), y.data_ptr(), y.stride(2),
                                            nn_ * hh * ww, cc, _lib.stream_ptr(dev)), "ryolo_add_nhwc")
            elif kind == 'up':
                xin, y = pl[0], pl[1]
                nn_, hh, ww, cc = xin.shape
                _lib.check(L.ryolo_upsample_nhwc(xin.data_ptr(), xin.stride(2), y.data_ptr(), y.stride(2), nn_, hh, ww, cc, 2,
                                                 _lib.stream_ptr(dev)), "ryolo_upsample_nhwc")
            elif kind == 'yolo':
                head, _, m, anchors, pbuf, io, hh, ww = pl
                stride = float(max(self.H, self.W)) / float(max(hh, ww))
                _lib.check(L.ryolo_yolo_decode(head.data_ptr(), head.stride(2), self.bs, hh, ww, m.na, self.model.nc + 6,
                                               anchors.data_ptr(), stride, 1.0, 0, io.data_ptr() if io is not None else None, m.na * hh * ww, 0,
                                               pbuf.data_ptr(), _lib.stream_ptr(dev)), "ryolo_yolo_decode")
                # what YOLOLayer.forward would have set (model_utils.py:16-35): the loss reads ng / anchor_vec
                if (m.nx, m.ny) != (ww, hh):
                    from .models import create_grids
                    create_grids(m, (self.H, self.W), (ww, hh), dev)

    # ------------------------------------------------------------------ backward
    def backward(self, pgrads):
        dev = self.device
        with torch.cuda.device(dev), torch.no_grad():
            if not hasattr(self, "static_pg"):
                self.static_pg = [torch.zeros_like(p) for p in self.p]
            ready = self.head_g_ready
            self.head_g_ready = False
            for k, (buf, g) in enumerate(zip(self.static_pg, pgrads)):
                if g is not None and ready and g.data_ptr() == buf.data_ptr():
                    continue                                  # the NHWC loss kernel already wrote this head's gradient
                if g is None:
                    buf.zero_()
                elif g.data_ptr() != buf.data_ptr():
                    buf.copy_(g)
                tr.pgrad_to_nhwc(buf, self.head_pairs[k][1])  # fp32 [bs,na,ny,nx,no] -> the head conv's NHWC bf16 gradient
                if self.fused_nhwc and g is not None:
                    buf.zero_()                               # keep the fused loss's scratch invariant (all zero)
            self._check_direct_sink()
            self._grad_of(next(self.model.parameters()))     # make sure the gradient sink exists
            segs = self._segments()
            if self.batch_reduce:
                self._ensure_reduce_batches(segs)
            if self.g_bwd is None:
                self.g_bwd = [None] * len(segs)
            for k, (lo, hi, params) in enumerate(segs):
                if not self.use_graph or self.steps < 2 or self.force_eager:
                    self._backward_launch(lo, hi)
                else:
                    if self.g_bwd[k] is None:
                        self.g_bwd[k] = self._capture(lambda lo=lo, hi=hi: self._backward_launch(lo, hi))
                    if self.g_bwd[k] is not None:
                        self.g_bwd[k].replay()
                self._flush_param_grads(params)
            self.steps += 1

    def _backward_launch(self, lo=0, hi=None):
        dev = self.device
        L = _lib.lib()
        pgrads = self.static_pg
        if lo == 0 and self.static_flat is not None:
            self.static_flat.zero_()
        if not self._red_planned:
            self._plan_reduce_fusion()
        hook = getattr(self, 'backward_hook', None)           # tests: called before / after every entry of the launch list (eager launches only)

        def entry(kind, i, pl, flags):
            if kind == 'yolo':
                pass                                          # converted (or written by the fused loss) before the segments run
            elif kind == 'conv':
                b = pl
                conv, bn = b['conv'], b['bn']
                res_first, in_first = flags
                dy = b['dy']
                if b['res_g'] is not None and not b['res_alias']:   # fused shortcut: the skip branch receives dy unchanged
                    self._passthrough(dy, b['res_g'], res_first)
                if bn is not None:
                    dsl = self._grad_of(b['act'].weight) if isinstance(b['act'], nn.PReLU) else None
                    if b['recompute'] and b['xin_g'] is None and b['dz'] is None:
                        # layer 0: BatchNorm / activation backward AND the weight gradient in one pass over dy (csrc/conv0_bwd.hip), no dz
                        if 'ws0f' not in b:
                            b['ws0f'] = tr.conv0_bn_bwd_wgrad_ws(dev)
                        tr.conv0_bn_bwd_wgrad(b['desc'], b['xin'], b['packed'], dy, b['stats'], b['actcode'], b['slope'],
                                              self._grad_of(bn.weight), self._grad_of(bn.bias), dsl, self._grad_of(conv.weight),
                                              conv.in_channels, True, b['ws0f'])
                        return
                    elif b['recompute']:
                        if 'ws0' not in b:
                            b['ws0'] = tr.conv0_bn_bwd_ws(dev)
                        tr.conv0_bn_bwd(b['desc'], b['xin'], b['packed'], dy, b['stats'], b['actcode'], b['slope'], b['dz'],
                                        self._grad_of(bn.weight), self._grad_of(bn.bias), dsl, b['ws0'])
                    elif b.get('red_part') is not None:       # the reduce pass ran inside the data gradient that produced dy
                        tr.bn_act_bwd_reduced(b['z'], dy, b['stats'], 1, b['slope'], b['dz'],
                                              self._grad_of(bn.weight), self._grad_of(bn.bias), dsl, b['red_part'], self.ws_b)
                    else:
                        tr.bn_act_bwd(b['z'], dy, b['stats'], b['actcode'], b['slope'], b['dz'],
                                      self._grad_of(bn.weight), self._grad_of(bn.bias), dsl, self.ws_b)
                elif conv.bias is not None:
                    tr.bn_act_bwd(b['z'], dy, None, 0, None, None, None, self._grad_of(conv.bias), None, self.ws_b)
                if self.batch_reduce:
                    # partial tiles only, into the layer's own workspace; the segment's reduces run as one launch below
                    tr.conv_wgrad_partials(b['desc'], b['xin'], b['dz'], conv.in_channels, self._grad_of(conv.weight), True, b['ws_w'])
                else:
                    tr.conv_wgrad(b['desc'], b['xin'], b['dz'], conv.in_channels, self._grad_of(conv.weight), True, self.ws_w)
                if b['xin_g'] is not None:
                    y = b.get('red_for')
                    if y is not None:     # 1x1 data gradient + the reduce pass of the block whose output this conv consumed
                        tr.conv_dgrad_bnreduce(b['desc'], b['dz'], b['packed_d'], self.ones, self.zeros, b['xin_g'], not in_first,
                                               y['z'], y['stats'], y['slope'], y['red_part'])
                    else:
                        tr.conv_dgrad(b['desc'], b['dz'], b['packed_d'], self.ones, self.zeros, b['xin_g'], not in_first)
            elif kind == 'add':
                dyv = pl[5]
                self._passthrough(dyv, pl[3], flags[0])
                self._passthrough(dyv, pl[4], flags[1])
            elif kind == 'up':
                tr.upsample2x_bwd(pl[3], pl[2], not flags)

        for kind, i, pl, flags in self.bplan[lo:hi]:
            if hook is not None:
                hook('pre', kind, i, pl, flags)
            entry(kind, i, pl, flags)
            if hook is not None:
                hook('post', kind, i, pl, flags)
        if self.batch_reduce:
            wb = self._wr_batches[(lo, len(self.bplan) if hi is None else hi)]    # KeyError = partial tiles nobody would reduce
            if wb is not None:
                wb.run()                                      # the segment's split-K reduces, one launch

    def _plan_reduce_fusion(self):
        """Pairs (X, Y) of consecutive backward entries where X is a 1x1 conv whose data gradient writes the FINAL gradient of
        block Y's output (X consumed Y's output; in a residual chain X accumulates into the chain's running gradient, which
        is Y's dy) and Y is a BatchNorm + PReLU/leaky block: X's data gradient then also runs the reduce pass of Y's backward
        on the values it stores (ryolo_conv2d_dgrad_bnreduce), and Y only finalises and applies.  RYOLO_BN_REDUCE_FUSION=0: off."""
        self._red_planned = True
        if os.environ.get("RYOLO_BN_REDUCE_FUSION", "1") == "0":
            return
        pairs = []
        for (k0, i0, x, f0), (k1, i1, y, f1) in zip(self.bplan[:-1], self.bplan[1:]):
            if k0 != 'conv' or k1 != 'conv' or x['xin_g'] is None or y['bn'] is None or y.get('recompute'):
                continue
            if y.get('actcode') != 1 or y.get('slope') is None or y.get('stats') is None:
                continue
            g, dy, z = x['xin_g'], y['dy'], y['z']
            if g.data_ptr() != dy.data_ptr() or g.shape != dy.shape or g.stride() != dy.stride() or z.shape != dy.shape:
                continue
            if g.stride(2) != g.shape[3]:
                continue
            rows = tr.dgrad_bnreduce_rows(x['desc'])
            if rows > 0:
                pairs.append((x, y, rows, g.shape[3]))
        if not pairs:
            return
        # one scratch for all pairs: a pair's rows are consumed by the very next entry of the launch list
        part = torch.empty(max(rows * 3 * c for _, _, rows, c in pairs), dtype=torch.float32, device=self.device)
        for x, y, rows, c in pairs:
            x['red_for'] = y
            y['red_part'] = part[:rows * 3 * c].view(rows, 3, c)

    def _passthrough(self, src, dst, is_first):
        L = _lib.lib()
        n, h, w, c = dst.shape
        if is_first:
            _lib.check(L.ryolo_upsample_nhwc(src.data_ptr(), src.stride(2), dst.data_ptr(), dst.stride(2), n, h, w, c, 1,
                                             _lib.stream_ptr(self.device)), "ryolo_upsample_nhwc")
        else:
            _lib.check(L.ryolo_add_nhwc(src.data_ptr(), src.stride(2), dst.data_ptr(), dst.stride(2), dst.data_ptr(),
                                        dst.stride(2), n * h * w, c, _lib.stream_ptr(self.device)), "ryolo_add_nhwc")

    def __call__(self, x):
        anchor = next(self.model.parameters())        # makes the Function part of the autograd graph
        return list(_Fn.apply(self, x, anchor))
