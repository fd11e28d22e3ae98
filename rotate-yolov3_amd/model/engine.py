"""HipEngine -- static layer plan of a Darknet cfg on the MI355X hot path.

Built once per (input shape, device) from `Darknet.module_defs` / `module_list` (the reference walks the cfg
dynamically on every forward, model/models.py:244-298).  The plan:

  * every tensor is NHWC bf16 resident in HBM; a `View` = (buffer, channel offset, C, H, W);
  * `shortcut` layers are folded into the producing conv's epilogue as a residual operand (models.py:281-282);
  * `upsample` layers are folded into the producing conv's epilogue as a 2x2 replicated store (models.py:93-94);
  * multi-input `route` layers own one concat buffer and their sources are WRITTEN INTO channel slices of it by
    whoever produces them (models.py:269-278: torch.cat copies); single-input routes are aliases;
  * BatchNorm (eval) is folded to fp32 scale/shift applied on the fp32 accumulator (utils/torch_utils.py:45-69),
    PReLU(1)/LeakyReLU become the epilogue's slope;
  * the three YOLO decodes write straight into one [bs, sum(na*ny*nx), no] tensor (the torch.cat of models.py:298).
Whatever cannot be folded falls back to the small NHWC kernels of csrc/yolo.hip (add / upsample / copy / maxpool).
All launches go to torch's current stream through the C ABI; after a warm-up the sequence can be replayed from a
hipGraph (`use_graph=True`).
"""
import ctypes as C

import torch
import torch.nn as nn

from .. import _lib
from . import hip_ops as ops

_vp = C.c_void_p
_lib.declare("ryolo_yolo_decode", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_float,
                                            C.c_float, C.c_int, _vp, C.c_longlong, C.c_longlong, _vp, _vp])
_lib.declare("ryolo_yolo_decode_filter", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_float,
                                                   C.c_float, C.c_int, C.c_float, C.c_float, C.c_longlong, C.c_longlong, _vp,
                                                   _vp, _vp, C.c_int, _vp])
_lib.declare("ryolo_conv_head_decode_supported", C.c_int, [C.POINTER(ops.ConvDesc), C.c_int, C.c_int])
_lib.declare("ryolo_conv_head_decode", C.c_int, [C.POINTER(ops.ConvDesc), _vp, _vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_float, C.c_float,
                                                 C.c_int, _vp, C.c_longlong, C.c_longlong, _vp, _vp])
_lib.declare("ryolo_add_nhwc", C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, C.c_longlong, C.c_int, _vp])
_lib.declare("ryolo_upsample_nhwc", C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp])
_lib.declare("ryolo_maxpool_nhwc", C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_int, _vp])


def _abs(i, l):
    return l if l > 0 else i + l      # route/shortcut index convention of models.py:101-114 (0 is "relative")


class HipEngine(object):
    def __init__(self, model, x_shape, device, use_graph=False, want_p=True):
        _lib.lib()   # fail loudly if the HIP library is missing
        self.device = device
        self.bs, cin, self.H, self.W = [int(v) for v in x_shape]
        if self.H % 32 or self.W % 32:
            raise RuntimeError("input height/width must be multiples of 32")
        self.want_p = want_p
        self.use_graph = use_graph
        self.graph = None
        defs = model.module_defs
        mods = model.module_list
        n = len(defs)
        cf = float((model.hyp or {}).get('context_factor', 1.0))
        arc = model.arc
        self.arc_code = 0 if 'default' in arc else (1 if 'BCE' in arc else 2)

        # ---- 1. shapes
        shp = []   # (C, H, W) per layer
        c, h, w = cin, self.H, self.W
        for i, d in enumerate(defs):
            t = d['type']
            if t == 'convolutional':
                conv = self._conv_of(mods[i])
                k, s, p = conv.kernel_size[0], conv.stride[0], conv.padding[0]
                c, h, w = conv.out_channels, (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
            elif t == 'maxpool':
                k, s = int(d['size']), int(d['stride'])
                if k == 2 and s == 1:
                    pass
                else:
                    p = (k - 1) // 2
                    h, w = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
            elif t == 'upsample':
                s = int(d['stride'])
                h, w = h * s, w * s
            elif t == 'route':
                ls = [_abs(i, int(v)) for v in d['layers'].split(',')]
                c = sum(shp[l][0] for l in ls)
                h, w = shp[ls[0]][1], shp[ls[0]][2]
                for l in ls:
                    if (shp[l][1], shp[l][2]) != (h, w):
                        raise RuntimeError("route %d joins tensors of different spatial size (reorg is out of scope)" % i)
            elif t == 'shortcut':
                j = _abs(i, int(d['from']))
                if shp[j] != (c, h, w):
                    raise RuntimeError("shortcut %d adds tensors of different shape" % i)
            elif t in ('yolo', 'reorg3d'):
                pass
            shp.append((c, h, w))
        self.shapes = shp

        # ---- 2. who reads what (to decide which epilogue fusions are legal)
        readers = [[] for _ in range(n)]
        for i, d in enumerate(defs):
            t = d['type']
            if t == 'route':
                for v in d['layers'].split(','):
                    readers[_abs(i, int(v))].append(i)
            else:
                if i > 0:
                    readers[i - 1].append(i)
                if t == 'shortcut':
                    readers[_abs(i, int(d['from']))].append(i)
        fused_into = {}     # follower layer (shortcut / upsample) -> conv layer that computes it
        conv_res, conv_ups = {}, {}
        for i, d in enumerate(defs):
            if i == 0 or defs[i - 1]['type'] != 'convolutional' or readers[i - 1] != [i] or (i - 1) in fused_into.values():
                continue
            if d['type'] == 'shortcut' and _abs(i, int(d['from'])) != i - 1:
                fused_into[i] = i - 1
                conv_res[i - 1] = _abs(i, int(d['from']))
            elif d['type'] == 'upsample' and int(d['stride']) == 2:
                fused_into[i] = i - 1
                conv_ups[i - 1] = 2

        # ---- 3. homes: sources of multi-input routes live inside the route's concat buffer
        def new_buf(c, h, w):
            return torch.empty((self.bs, h, w, c), dtype=torch.bfloat16, device=device)

        views = [None] * n
        home = {}
        route_copies = {}    # route layer -> [(src layer, channel offset)] that must be copied at run time
        alias = {}
        for i, d in enumerate(defs):
            if d['type'] != 'route':
                continue
            ls = [_abs(i, int(v)) for v in d['layers'].split(',')]
            if len(ls) == 1:
                alias[i] = ls[0]
                continue
            c, h, w = shp[i]
            buf = new_buf(c, h, w)
            views[i] = buf
            off = 0
            for l in ls:
                src = l
                while src in alias:
                    src = alias[src]
                if src not in home and defs[src]['type'] in ('convolutional', 'shortcut', 'upsample', 'maxpool') \
                        and src < i and shp[src][0] % 8 == 0 and off % 8 == 0:
                    home[src] = buf[..., off:off + shp[src][0]]
                else:
                    route_copies.setdefault(i, []).append((src, off))
                off += shp[l][0]

        def view_for(i):
            c, h, w = shp[i]
            return home[i] if i in home else new_buf(c, h, w)

        # ---- 4. ops
        self.x_nhwc = torch.empty((self.bs, self.H, self.W, 8), dtype=torch.bfloat16, device=device)
        self.ops = []
        self.decodes = []   # (op index, head view, ny, nx, na, anchors, stride, cf, row offset) per yolo layer
        self.op_info = []   # per op: kind / kernel name / algorithmic flops and bytes (bench + profiling)
        self.keep = []      # tensors the closures reference
        yolo_rows = []
        for i in self.yolo_idx(defs):
            c, h, w = shp[i]
            yolo_rows.append(mods[i].na * h * w)
        self.total_rows = sum(yolo_rows)
        self.no = (model.nc + 6) if yolo_rows else 0
        self.io = torch.empty((self.bs, self.total_rows, self.no), dtype=torch.float32, device=device) if yolo_rows else None
        self.p = []
        row_off = 0
        yi = 0
        L = _lib.lib()

        def src_view(i):
            return self.x_nhwc if i < 0 else views[i]

        pending = None      # first layer of a fused stem pair, waiting for its successor
        pending_head = None  # last conv of a YOLO head, emitted together with its decode
        for i, d in enumerate(defs):
            t = d['type']
            if i in fused_into:
                views[i] = views[fused_into[i]]        # the conv already produced this layer's tensor
                continue
            if t == 'convolutional':
                conv = self._conv_of(mods[i])
                bn = self._bn_of(mods[i])
                act, slope = self._act_of(mods[i])
                xin = src_view(i - 1)                # (None behind the first layer of a fused pair: that tensor only exists in LDS)
                k, s = conv.kernel_size[0], conv.stride[0]
                pad = conv.padding[0]
                cin_k = pending['cout'] if pending is not None else xin.shape[-1]
                wt = conv.weight.detach().float()
                packed = ops.pack_weights(wt, cin_pad=cin_k)
                if bn is not None:
                    scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
                    shift = bn.bias.detach().float() - bn.running_mean.detach().float() * scale
                    if conv.bias is not None:
                        shift = shift + conv.bias.detach().float() * scale
                else:
                    scale = torch.ones(conv.out_channels, device=device)
                    shift = conv.bias.detach().float() if conv.bias is not None else torch.zeros(conv.out_channels, device=device)
                cp = ops.cpad(conv.out_channels)
                scale, shift = ops.pad_vec(scale, cp), ops.pad_vec(shift, cp)
                final = i
                res = None
                ups = 1
                if i in conv_res:
                    res = views[conv_res[i]]
                    final = i + 1
                if i in conv_ups:
                    ups = 2
                    final = i + 1
                pair_first = pending is None and self._pair_candidate(i, defs, readers, conv_res, conv_ups, home)
                head_cand = (bn is None and k == 1 and s == 1 and i + 1 < n and defs[i + 1]['type'] == 'yolo' and readers[i] == [i + 1]
                             and i not in home and i not in conv_res and i not in conv_ups)
                out = None if (pair_first or head_cand) else view_for(final)
                views[i] = out
                if conv.out_channels % 8 or cin_k % 8:
                    raise RuntimeError("conv %d: channel counts must be multiples of 8 for the HIP path" % i)
                self.keep += [packed, scale, shift]
                ho, wo = shp[i][1], shp[i][2]
                me = dict(layer=i, xin=xin, packed=packed, scale=scale, shift=shift, cout=conv.out_channels, ksize=k, stride=s, pad=pad,
                          act=act, slope=slope, cin=conv.in_channels, wnumel=conv.weight.numel(), ho=ho, wo=wo)
                if pending is not None:
                    # second layer of a fused stem pair (csrc/conv_stem.hip): the first layer's tensor is computed into LDS, never stored
                    first, pending = pending, None
                    self.ops.append(self._mk_pair(first, me, res is not None, out))
                    self.op_info.append(dict(
                        kind='conv', layer=i, name='conv_stem_pair<k%ds%d+k%ds%d%s>' % (first['ksize'], first['stride'], k, s, '+res' if res is not None else ''),
                        flops=2.0 * self.bs * (first['ksize'] ** 2 * first['cin'] * first['cout'] * first['ho'] * first['wo'] +
                                               k * k * conv.in_channels * conv.out_channels * ho * wo),
                        bytes=2.0 * self.bs * (first['xin'].shape[1] * first['xin'].shape[2] * first['cin'] + ho * wo * conv.out_channels)
                        + 2.0 * (first['wnumel'] + conv.weight.numel())))
                    continue
                if (bn is None and act == ops.ACT_LINEAR and k == 1 and s == 1 and res is None and ups == 1 and i + 1 < n and
                        defs[i + 1]['type'] == 'yolo' and readers[i] == [i + 1] and i not in home and self._head_fusable(me, mods[i + 1])):
                    # a YOLO head: conv + decode in one launch (ryolo_conv_head_decode); the head tensor only exists when detect() asks for it
                    pending_head = me
                    views[i] = None
                    continue
                if pair_first:
                    if self._pairs_with_next(i, defs, mods, conv_res, views, me):
                        pending = me                     # emitted together with layer i + 1
                        continue
                if out is None:
                    out = view_for(final)
                    views[i] = out
                self.ops.append(self._mk_conv(xin, packed, scale, shift, conv.out_channels, k, s, pad, act, slope, res,
                                              out, ups))
                # the kernel the library's dispatch takes for this launch (dry run of csrc/conv.hip dispatch())
                kname = ops.conv_kernel_name(self.bs, xin.shape[1], xin.shape[2], cin_k, conv.out_channels, k, s, pad, in_cs=xin.stride(2),
                                             out_cs=out.stride(2), res_cs=res.stride(2) if res is not None else 0, upsample=ups,
                                             residual=res is not None)
                self.op_info.append(dict(
                    kind='conv', layer=i,
                    name=kname,
                    flops=2.0 * k * k * conv.in_channels * conv.out_channels * ho * wo * self.bs,
                    bytes=2.0 * self.bs * (xin.shape[1] * xin.shape[2] * conv.in_channels + ho * wo * conv.out_channels *
                                           (ups * ups + (1 if res is not None else 0))) + 2.0 * conv.weight.numel()))
            elif t == 'shortcut':
                a, b = views[i - 1], views[_abs(i, int(d['from']))]
                out = view_for(i)
                views[i] = out
                self.ops.append(self._mk_add(a, b, out))
                self.op_info.append(dict(kind='add', layer=i, name='add_nhwc', flops=0.0, bytes=6.0 * out.numel()))
            elif t == 'upsample':
                xin = views[i - 1]
                out = view_for(i)
                views[i] = out
                self.ops.append(self._mk_upsample(xin, out, int(d['stride'])))
                self.op_info.append(dict(kind='upsample', layer=i, name='upsample_nhwc', flops=0.0, bytes=2.0 * (xin.numel() + out.numel())))
            elif t == 'maxpool':
                xin = views[i - 1]
                out = view_for(i)
                views[i] = out
                self.ops.append(self._mk_maxpool(xin, out, int(d['size']), int(d['stride'])))
                self.op_info.append(dict(kind='maxpool', layer=i, name='maxpool_nhwc', flops=0.0, bytes=2.0 * (xin.numel() + out.numel())))
            elif t == 'route':
                if i in alias:
                    views[i] = views[alias[i]]
                else:
                    for (src, off) in route_copies.get(i, []):
                        dst = views[i][..., off:off + shp[src][0]]
                        self.ops.append(self._mk_upsample(views[src], dst, 1))
                        self.op_info.append(dict(kind='copy', layer=i, name='upsample_nhwc', flops=0.0, bytes=4.0 * dst.numel()))
            elif t == 'yolo':
                m = mods[i]
                c, h, w = shp[i]
                head = views[i - 1]
                anchors = m.anchors.to(device=device, dtype=torch.float32).contiguous()
                stride = float(max(self.H, self.W)) / float(max(h, w))      # model_utils.py:19-20
                pbuf = torch.empty((self.bs, m.na, h, w, self.no), dtype=torch.float32, device=device) if want_p else None
                self.p.append(pbuf)
                self.keep.append(anchors)
                if pending_head is not None:
                    hc, pending_head = pending_head, None
                    lazy = dict(t=None, conv=hc, shape=(self.bs, h, w, hc['cout']))        # detect() materialises the head tensor on demand
                    self.ops.append(self._mk_head_decode(hc, h, w, m.na, anchors, stride, cf, row_off, pbuf))
                    self.decodes.append((len(self.ops) - 1, lazy, h, w, m.na, anchors, stride, cf, row_off))
                    self.op_info.append(dict(
                        kind='conv', layer=i, name='conv_pw<k1,K%d>+decode' % hc['cin'],
                        flops=2.0 * hc['cin'] * hc['cout'] * h * w * self.bs,
                        bytes=2.0 * self.bs * h * w * hc['cin'] + 2.0 * hc['wnumel'] + self.bs * m.na * h * w * self.no * (4.0 + (4.0 if want_p else 0.0))))
                    row_off += yolo_rows[yi]
                    yi += 1
                    views[i] = None
                    continue
                self.ops.append(self._mk_decode(head, h, w, m.na, anchors, stride, cf, row_off, pbuf))
                self.decodes.append((len(self.ops) - 1, head, h, w, m.na, anchors, stride, cf, row_off))
                self.op_info.append(dict(kind='decode', layer=i, name='yolo_decode', flops=0.0,
                                         bytes=self.bs * m.na * h * w * self.no * (2.0 + 4.0 + (4.0 if want_p else 0.0))))
                row_off += yolo_rows[yi]
                yi += 1
                views[i] = head
            elif t == 'reorg3d':
                views[i] = views[i - 1]
        self.views = views

    # ------------------------------------------------------------------ helpers
    @staticmethod
    def yolo_idx(defs):
        return [i for i, d in enumerate(defs) if d['type'] == 'yolo']

    @staticmethod
    def _conv_of(m):
        for s in m:
            if isinstance(s, nn.Conv2d):
                return s
        raise RuntimeError("convolutional block without Conv2d")

    @staticmethod
    def _bn_of(m):
        for s in m:
            if isinstance(s, nn.BatchNorm2d):
                return s
        return None

    @staticmethod
    def _act_of(m):
        for s in m:
            if isinstance(s, nn.PReLU):
                if s.weight.numel() != 1:
                    raise RuntimeError("per-channel PReLU is not on the HIP path")
                return ops.ACT_LEAKY, float(s.weight.detach().float().item())
            if isinstance(s, nn.LeakyReLU):
                return ops.ACT_LEAKY, float(s.negative_slope)
            if type(s).__name__ == 'Mish':
                return ops.ACT_MISH, 0.0
            if not isinstance(s, (nn.Conv2d, nn.BatchNorm2d)):
                # an activation (Swish, ...) the HIP conv epilogue does not implement must not silently run as linear
                raise RuntimeError("activation %s is not on the HIP path (use model.backend = 'torch')" % type(s).__name__)
        return ops.ACT_LINEAR, 0.0

    # ---- fused stem pairs (ryolo_conv2d_bn_act_pair): layers 0-1 and 2-4 of Darknet-53, where half of the HBM traffic is a tensor one
    # layer writes only for the next to read.  RYOLO_STEM_PAIR=0 keeps one launch per layer (A/B timing, tests).
    def _pair_candidate(self, i, defs, readers, conv_res, conv_ups, home):
        import os
        if os.environ.get("RYOLO_STEM_PAIR", "1") == "0" or i + 1 >= len(defs) or defs[i + 1]['type'] != 'convolutional':
            return False
        return readers[i] == [i + 1] and i not in conv_res and i not in conv_ups and i not in home and (i + 1) not in conv_ups

    def _pairs_with_next(self, i, defs, mods, conv_res, views, me):
        nxt = self._conv_of(mods[i + 1])
        if self._bn_of(mods[i + 1]) is None:
            return False
        act2, slope2 = self._act_of(mods[i + 1])
        second = dict(cout=nxt.out_channels, ksize=nxt.kernel_size[0], stride=nxt.stride[0], pad=nxt.padding[0], act=act2, slope=slope2)
        res_layer = conv_res.get(i + 1)
        shortcut = res_layer is not None
        if shortcut and views[res_layer] is not me['xin']:
            return False                        # the shortcut must come from the first layer's own input (it is taken from the LDS image)
        try:
            return ops.conv_pair_supported(me['xin'], me, second, shortcut)
        except RuntimeError:
            return False

    # ---- a YOLO head as one launch (ryolo_conv_head_decode).  RYOLO_HEAD_DECODE=0 keeps conv + decode as two launches.
    def _head_desc(self, hc):
        xin = hc['xin']
        return ops.ConvDesc(self.bs, xin.shape[1], xin.shape[2], xin.shape[3], hc['cout'], 1, 1, 0, xin.stride(2), hc['cout'], 0,
                            ops.ACT_LINEAR, 0.0, 1, 0)

    def _head_fusable(self, hc, yolo_mod):
        import os
        if os.environ.get("RYOLO_HEAD_DECODE", "1") == "0":
            return False
        d = self._head_desc(hc)
        return bool(_lib.lib().ryolo_conv_head_decode_supported(C.byref(d), int(yolo_mod.na), int(self.no)))

    def _mk_head_decode(self, hc, ny, nx, na, anchors, stride, cf, row_off, pbuf):
        d = self._head_desc(hc)

        def run():
            _lib.check(_lib.lib().ryolo_conv_head_decode(C.byref(d), hc['xin'].data_ptr(), hc['packed'].data_ptr(), hc['scale'].data_ptr(),
                                                         hc['shift'].data_ptr(), anchors.data_ptr(), na, self.no, stride, cf, self.arc_code,
                                                         self.io.data_ptr(), self.total_rows, row_off,
                                                         pbuf.data_ptr() if pbuf is not None else None, _lib.stream_ptr(self.device)),
                       "ryolo_conv_head_decode")
        return run

    def _head_tensor(self, lazy):
        """the head tensor of a fused head, for detect()'s decode + filter kernel: allocated and computed on demand"""
        hc = lazy['conv']
        if lazy['t'] is None:
            lazy['t'] = torch.empty(lazy['shape'], dtype=torch.bfloat16, device=self.device)
        ops.conv2d_bn_act(hc['xin'], hc['packed'], hc['scale'], hc['shift'], hc['cout'], 1, stride=1, pad=0, act=ops.ACT_LINEAR, out=lazy['t'])
        return lazy['t']

    def _mk_pair(self, first, second, shortcut, out):
        def run():
            ops.conv2d_bn_act_pair(first['xin'], first, second, first['packed'], first['scale'], first['shift'], second['packed'],
                                   second['scale'], second['shift'], shortcut_from_input=shortcut, out=out)
        return run

    def _mk_conv(self, xin, packed, scale, shift, cout, k, s, pad, act, slope, res, out, ups):
        def run():
            ops.conv2d_bn_act(xin, packed, scale, shift, cout, k, stride=s, pad=pad, act=act, slope=slope, residual=res,
                              out=out, upsample=ups)
        return run

    def _mk_add(self, a, b, out):
        n, h, w, c = out.shape

        def run():
            _lib.check(_lib.lib().ryolo_add_nhwc(a.data_ptr(), a.stride(2), b.data_ptr(), b.stride(2), out.data_ptr(),
                                                 out.stride(2), n * h * w, c, _lib.stream_ptr(self.device)), "ryolo_add_nhwc")
        return run

    def _mk_upsample(self, xin, out, s):
        n, h, w, c = xin.shape

        def run():
            _lib.check(_lib.lib().ryolo_upsample_nhwc(xin.data_ptr(), xin.stride(2), out.data_ptr(), out.stride(2), n, h, w,
                                                      c, s, _lib.stream_ptr(self.device)), "ryolo_upsample_nhwc")
        return run

    def _mk_maxpool(self, xin, out, k, s):
        n, h, w, c = xin.shape

        def run():
            _lib.check(_lib.lib().ryolo_maxpool_nhwc(xin.data_ptr(), xin.stride(2), out.data_ptr(), out.stride(2), n, h, w,
                                                     c, k, s, _lib.stream_ptr(self.device)), "ryolo_maxpool_nhwc")
        return run

    def _mk_decode(self, head, ny, nx, na, anchors, stride, cf, row_off, pbuf):
        def run():
            _lib.check(_lib.lib().ryolo_yolo_decode(head.data_ptr(), head.stride(2), self.bs, ny, nx, na, self.no,
                                                    anchors.data_ptr(), stride, cf, self.arc_code, self.io.data_ptr(),
                                                    self.total_rows, row_off, pbuf.data_ptr() if pbuf is not None else None,
                                                    _lib.stream_ptr(self.device)), "ryolo_yolo_decode")
        return run

    # ------------------------------------------------------------------ run
    def _launch_input(self, x):
        n, c, h, w = x.shape
        _lib.check(_lib.lib().ryolo_nchw_f32_to_nhwc_bf16(x.data_ptr(), n, c, h, w, 8, self.x_nhwc.data_ptr(),
                                                          _lib.stream_ptr(self.device)), "ryolo_nchw_f32_to_nhwc_bf16")

    def _launch_all(self, x):
        self._launch_input(x)
        for op in self.ops:
            op()

    def detect(self, x, conf_thres=0.5, nms_thres=0.5, capacity=1 << 18):
        """forward + post-processing for inference: the yolo layers run the fused decode + confidence filter +
        compaction kernel (ryolo_yolo_decode_filter; the [bs, rows, no] `io` tensor is never written), the survivors go
        through ONE segmented rotated-NMS launch.  Returns what non_max_suppression(model(x)[0], conf_thres, nms_thres)
        returns: list[bs] of [k, 8] rows (x, y, w, h, a, score, class_conf, class) by descending score, or None."""
        from ..utils.nms.nms import nms_from_candidates
        if tuple(x.shape) != (self.bs, x.shape[1], self.H, self.W) or x.shape[1] > 8:
            raise RuntimeError("engine was planned for input %s" % ((self.bs, x.shape[1], self.H, self.W),))
        x = x.float().contiguous()
        L = _lib.lib()
        dev = self.device
        skip = set(d[0] for d in self.decodes)
        with torch.cuda.device(dev):
            while True:
                if getattr(self, '_cand_cap', 0) < capacity:
                    self._cand = torch.empty((capacity, 8), dtype=torch.float32, device=dev)
                    self._cand_row = torch.empty(capacity, dtype=torch.int64, device=dev)
                    self._cand_cnt = torch.zeros(1, dtype=torch.int32, device=dev)
                    self._cand_cap = capacity
                self._cand_cnt.zero_()
                self._launch_input(x)
                for k, op in enumerate(self.ops):
                    if k not in skip:
                        op()
                for (_, head, ny, nx, na, anchors, stride, cf, row_off) in self.decodes:
                    if isinstance(head, dict):
                        head = self._head_tensor(head)
                    _lib.check(L.ryolo_yolo_decode_filter(head.data_ptr(), head.stride(2), self.bs, ny, nx, na, self.no,
                                                          anchors.data_ptr(), stride, cf, self.arc_code, float(conf_thres), 2.0,
                                                          self.total_rows, row_off, self._cand.data_ptr(),
                                                          self._cand_row.data_ptr(), self._cand_cnt.data_ptr(), self._cand_cap,
                                                          _lib.stream_ptr(dev)), "ryolo_yolo_decode_filter")
                m = int(self._cand_cnt.item())
                if m <= self._cand_cap:
                    break
                capacity = 2 * m                      # the candidate buffer overflowed: grow it and run again
            if m == 0:
                return [None] * self.bs
            rowid, o = self._cand_row[:m].sort()      # the reference's order: (image, row) ascending
            cand = self._cand[:m][o]
            return nms_from_candidates(torch.div(rowid, self.total_rows, rounding_mode='floor'), cand, self.bs, nms_thres,
                                       nc=self.no - 6)

    def __call__(self, x):
        if tuple(x.shape) != (self.bs, x.shape[1], self.H, self.W) or x.shape[1] > 8:
            raise RuntimeError("engine was planned for input %s" % ((self.bs, x.shape[1], self.H, self.W),))
        if x.dtype != torch.float32:
            x = x.float()
        x = x.contiguous()
        with torch.cuda.device(self.device):
            if not self.use_graph:
                self._launch_all(x)
            else:
                if self.graph is None:
                    self.static_x = x.clone()
                    self._launch_all(self.static_x)          # warm-up: one-time attribute setup happens here
                    torch.cuda.synchronize(self.device)
                    self.graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
                        self._launch_all(self.static_x)
                self.static_x.copy_(x)
                self.graph.replay()
        return self.io, tuple(self.p)
