"""Python handles on the training-step entry points of the C ABI (include/ryolo.h, csrc/train.hip + csrc/conv.hip).
Plumbing only: torch owns the device memory and the stream."""
import ctypes as C

import torch

from .. import _lib
from .hip_ops import ConvDesc, _check_nhwc, cpad

_vp = C.c_void_p
_P = C.POINTER(ConvDesc)
_lib.declare("ryolo_conv_stat_rows", C.c_int, [_P])
_lib.declare("ryolo_conv2d_bn_act_stats", C.c_int, [_P, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp])
_lib.declare("ryolo_bn_finalize", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_longlong, C.c_float, C.c_float, _vp, _vp, _vp,
                                            _vp, _vp, _vp, _vp, _vp, _vp])
_lib.declare("ryolo_bn_act_fwd", C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, _vp, _vp, C.c_int, _vp, C.c_int, C.c_longlong,
                                           C.c_int, _vp])
_lib.declare("ryolo_bn_act_bwd_workspace_bytes", C.c_size_t, [C.c_longlong, C.c_int])
_lib.declare("ryolo_bn_act_bwd", C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int,
                                           C.c_longlong, C.c_int, _vp, _vp, _vp, _vp, C.c_size_t, _vp])
_lib.declare("ryolo_conv_packed_dgrad_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int, C.c_int])
_lib.declare("ryolo_conv_pack_weights_dgrad", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp])
_lib.declare("ryolo_conv_dgrad_tap_table", C.c_int, [C.c_int, C.c_int, _vp])
_lib.declare("ryolo_conv2d_dgrad", C.c_int, [_P, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp])
_lib.declare("ryolo_conv2d_dgrad_bnreduce_rows", C.c_int, [_P])
_lib.declare("ryolo_conv2d_dgrad_bnreduce", C.c_int, [_P, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp,
                                                      _vp, _vp])
_lib.declare("ryolo_bn_act_bwd_reduced", C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int,
                                                   C.c_longlong, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, C.c_size_t, _vp])
_lib.declare("ryolo_conv_wgrad_workspace_bytes", C.c_size_t, [_P])
_lib.declare("ryolo_conv2d_wgrad", C.c_int, [_P, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_size_t, _vp])
_lib.declare("ryolo_conv2d_wgrad_partials", C.c_int, [_P, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_size_t, _vp])
_lib.declare("ryolo_conv2d_wgrad_reduce", C.c_int, [_P, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_size_t, _vp])
_lib.declare("ryolo_upsample2x_bwd", C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp])
_lib.declare("ryolo_pgrad_to_nhwc", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp])


_lib.declare("ryolo_yolo_loss_bitmap_bytes", C.c_size_t, [C.c_longlong])
_lib.declare("ryolo_yolo_loss", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp,
                                          _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_int, _vp, _vp, _vp, _vp])
_lib.declare("ryolo_yolo_loss_nhwc", C.c_int, [_vp, C.c_int, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp,
                                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_float, C.c_float, C.c_float, C.c_float,
                                               C.c_float, C.c_float, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp])
_lib.declare("ryolo_scale_bf16_if", C.c_int, [_vp, _vp, C.c_int, C.c_longlong, C.c_int, _vp])
_lib.declare("ryolo_yolo_loss_bitmap_bytes_arc", C.c_size_t, [C.c_longlong, C.c_int, C.c_int])
_lib.declare("ryolo_yolo_loss_arc", C.c_int, [_vp] + [C.c_int] * 6 + [_vp, C.c_int] + [_vp] * 9 + [C.c_float] * 6 +
             [C.c_int, C.c_int, C.c_float] + [_vp] * 4)
_lib.declare("ryolo_yolo_loss_nhwc_arc", C.c_int, [_vp, C.c_int, _vp] + [C.c_int] * 6 + [_vp, C.c_int] + [_vp] * 9 + [C.c_float] * 6 +
             [C.c_int, C.c_int, C.c_float] + [_vp, _vp, _vp, C.c_int, _vp, _vp])
ARC_FOCAL, ARC_UBCE, ARC_UCE = 1, 2, 4


def arc_flags(arc):
    """compute_loss's arc string (model/loss.py:268: default, uCE, uBCE, each optionally with an F for the focal wrappers) ->
    the flag word of ryolo_yolo_loss_arc"""
    f = ARC_FOCAL if 'F' in arc else 0
    if 'default' in arc:
        return f
    if 'BCE' in arc:
        return f | ARC_UBCE
    if 'CE' in arc:
        return f | ARC_UCE
    raise ValueError("unknown arc %r" % (arc,))
_lib.declare("ryolo_riou_loss_pairs", C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp])


class PackJob(C.Structure):
    """ryolo_pack_job (include/ryolo.h)."""
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("kind", C.c_int), ("Cout", C.c_int), ("Cin", C.c_int),
                ("KS", C.c_int), ("Cin_pad", C.c_int), ("ntaps", C.c_int), ("Kpad", C.c_int), ("rows", C.c_int),
                ("khs", C.c_int * 9), ("kws", C.c_int * 9), ("block_begin", C.c_int), ("block_end", C.c_int)]


_lib.declare("ryolo_conv_pack_job_fill", C.c_int, [C.POINTER(PackJob), _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp])
_lib.declare("ryolo_conv_pack_batch", C.c_int, [_vp, C.c_int, C.c_int, _vp])


class WgradReduceJob(C.Structure):
    _fields_ = [("part", C.c_void_p), ("g", C.c_void_p), ("S", C.c_int), ("Cout", C.c_int), ("Cin_real", C.c_int), ("Cin_k", C.c_int),
                ("ks", C.c_int), ("Kpad", C.c_int), ("Cout_pad", C.c_int), ("accumulate", C.c_int), ("kind", C.c_int),
                ("block_begin", C.c_int), ("block_end", C.c_int), ("wide", C.c_int)]


_lib.declare("ryolo_conv_wgrad_reduce_job_fill", C.c_int, [C.POINTER(WgradReduceJob), _P, C.c_int, _vp, _vp, C.c_int])
_lib.declare("ryolo_conv_wgrad_reduce_batch", C.c_int, [_vp, C.c_int, C.c_int, _vp])


class WgradReduceBatch(object):
    """The split-K reduces of several layers as one launch (csrc/train.hip: wgrad_reduce_batch_kernel).  Every layer keeps its partial
    tiles in its OWN workspace (conv_wgrad_partials); add() describes a layer's reduce, finalize() uploads the job table, run() reduces all
    of them -- bit-identical to the per-layer reduces."""

    def __init__(self, device):
        self.device = device
        self.jobs = []
        self.keep = []

    def add(self, d, cin_real, ws, grad, accumulate):
        j = WgradReduceJob()
        n = _lib.lib().ryolo_conv_wgrad_reduce_job_fill(C.byref(j), C.byref(d), cin_real, ws.data_ptr(), grad.data_ptr(), 1 if accumulate else 0)
        if n <= 0:
            raise RuntimeError("ryolo_conv_wgrad_reduce_job_fill failed")
        self.jobs.append(j)
        self.keep.append((ws, grad))

    def finalize(self):
        arr = (WgradReduceJob * len(self.jobs))()
        blk = 0
        for q, j in enumerate(self.jobs):
            nb = j.block_end
            j.block_begin, j.block_end = blk, blk + nb
            blk += nb
            arr[q] = j
        self.total_blocks = blk
        self.dev_jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self.n = len(self.jobs)

    def run(self):
        _lib.check(_lib.lib().ryolo_conv_wgrad_reduce_batch(self.dev_jobs.data_ptr(), self.n, self.total_blocks, _s(self.device)),
                   "ryolo_conv_wgrad_reduce_batch")


class WeightPackBatch(object):
    """Every weight pack of a training step (forward layout + dgrad classes of each conv) as one launch."""

    def __init__(self, device):
        self.device = device
        self.jobs = []

    def add(self, weight, stride, cin_pad, packed_fwd, packed_dgrad):
        cout, cin, k, _ = weight.shape
        tmp = (PackJob * 8)()      # forward + up to 4 classic + 2 x-fused dgrad images
        n = _lib.lib().ryolo_conv_pack_job_fill(tmp, weight.data_ptr(), cout, cin, k, stride, cin_pad, packed_fwd.data_ptr(),
                                                packed_dgrad.data_ptr() if packed_dgrad is not None else None)
        if n <= 0:
            raise RuntimeError("ryolo_conv_pack_job_fill failed")
        for q in range(n):
            j = PackJob()
            C.memmove(C.byref(j), C.byref(tmp[q]), C.sizeof(PackJob))
            self.jobs.append(j)

    def finalize(self):
        arr = (PackJob * len(self.jobs))()
        blk = 0
        for q, j in enumerate(self.jobs):
            nb = j.block_end
            j.block_begin, j.block_end = blk, blk + nb
            blk += nb
            arr[q] = j
        self.total_blocks = blk
        raw = bytes(arr)
        self.dev_jobs = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self.n = len(self.jobs)

    def run(self):
        _lib.check(_lib.lib().ryolo_conv_pack_batch(self.dev_jobs.data_ptr(), self.n, self.total_blocks, _s(self.device)),
                   "ryolo_conv_pack_batch")


def _s(dev):
    return _lib.stream_ptr(dev)


def make_desc(x, cout, ksize, stride, pad, out_cs=None, tile=0):
    in_cs = _check_nhwc(x, "x")
    n, h, w, cin = x.shape
    return ConvDesc(n, h, w, cin, cout, ksize, stride, pad, in_cs, out_cs or cout, 0, 0, 0.0, 1, tile)


def stat_rows():
    """Partial rows of the statistics scratch the conv kernels spread their atomics over (a library constant)."""
    d = ConvDesc(1, 8, 8, 8, 8, 1, 1, 0, 8, 8, 0, 0, 0.0, 1, 0)
    rows = _lib.lib().ryolo_conv_stat_rows(C.byref(d))
    if rows <= 0:
        raise RuntimeError("ryolo_conv_stat_rows failed")
    return rows


def conv_fwd_stats(d, x, packed_w, ones, shift, z, part=None, clear=True):
    """z = conv(x, W) (linear; `shift` must be all zeros here, see include/ryolo.h: the statistics include the padded rows of the
    last pixel tile, which are zeros only without a bias); returns the partial sums (fp64) [rows, 2, cpad] of z and z^2.
    clear=False: the caller guarantees the scratch is zero (bn_finalize zeroes what it read, so a scratch that started
    zeroed stays clean from conv to conv)."""
    L = _lib.lib()
    rows = L.ryolo_conv_stat_rows(C.byref(d))
    cp = cpad(d.Cout)
    if part is None:
        part = torch.zeros((rows, 2, cp), dtype=torch.float64, device=x.device)
    else:                       # caller-owned scratch (any shape with enough elements): carve [rows, 2, cp]
        part = part.view(-1)[:rows * 2 * cp].view(rows, 2, cp)
        if clear:
            part.zero_()
    if z is not None:
        d.out_cstride = _check_nhwc(z, "z")
    # z = None: statistics only (layer 0 of the training engine: conv0_recompute_supported(d); z is recomputed, never stored)
    _lib.check(L.ryolo_conv2d_bn_act_stats(C.byref(d), x.data_ptr(), packed_w.data_ptr(), ones.data_ptr(), shift.data_ptr(),
                                           None, z.data_ptr() if z is not None else None, part.data_ptr(), _s(x.device)),
               "ryolo_conv2d_bn_act_stats")
    return part


_lib.declare("ryolo_conv0_recompute_supported", C.c_int, [C.POINTER(ConvDesc)])
_lib.declare("ryolo_conv0_bn_act_fwd", C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp])
_lib.declare("ryolo_conv0_bn_bwd_workspace_bytes", C.c_size_t, [])
_lib.declare("ryolo_conv0_bn_bwd", C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, C.c_int,
                                             _vp, _vp, _vp, _vp, C.c_size_t, C.c_int, _vp])


def conv0_recompute_supported(d):
    """Layer 0 (3x3 / stride 1 / pad 1, 8 padded input channels -> 32): train without storing the conv output (include/ryolo.h)."""
    return bool(_lib.lib().ryolo_conv0_recompute_supported(C.byref(d)))


def conv0_bn_act_fwd(d, x, packed_w, scale, shift, act, slope, y):
    """y = act(conv(x, W) * scale + shift), batch statistics already folded into scale / shift; slope: device scalar (PReLU)."""
    d.out_cstride = _check_nhwc(y, "y")
    _lib.check(_lib.lib().ryolo_conv0_bn_act_fwd(C.byref(d), x.data_ptr(), packed_w.data_ptr(), scale.data_ptr(), shift.data_ptr(), act,
                                                 slope.data_ptr() if slope is not None else None, y.data_ptr(), _s(x.device)),
               "ryolo_conv0_bn_act_fwd")
    return y


def conv0_bn_bwd_ws(device):
    """zeroed workspace of conv0_bn_bwd (the call leaves it zeroed)"""
    return torch.zeros(_lib.lib().ryolo_conv0_bn_bwd_workspace_bytes(), dtype=torch.uint8, device=device)


def conv0_bn_bwd(d, x, packed_w, dy, stats, act, slope, dz, dgamma, dbeta, dslope, ws, ws_is_zero=True):
    """bn_act_bwd for layer 0 with z recomputed from x in both passes (no stored conv output)."""
    mean, invstd, scale, shift = stats
    _lib.check(_lib.lib().ryolo_conv0_bn_bwd(C.byref(d), x.data_ptr(), packed_w.data_ptr(), dy.data_ptr(), dy.stride(2),
                                             scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), act,
                                             slope.data_ptr() if slope is not None else None, dz.data_ptr(), dz.stride(2),
                                             dgamma.data_ptr(), dbeta.data_ptr(), dslope.data_ptr() if dslope is not None else None,
                                             ws.data_ptr(), ws.numel(), 1 if ws_is_zero else 0, _s(x.device)), "ryolo_conv0_bn_bwd")
    return dz


_lib.declare("ryolo_conv0_bn_bwd_wgrad_workspace_bytes", C.c_size_t, [])
_lib.declare("ryolo_conv0_bn_bwd_wgrad", C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp,
                                                  _vp, C.c_int, C.c_int, _vp, C.c_size_t, _vp])


def conv0_bn_bwd_wgrad_ws(device):
    return torch.empty(_lib.lib().ryolo_conv0_bn_bwd_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)


def conv0_bn_bwd_wgrad(d, x, packed_w, dy, stats, act, slope, dgamma, dbeta, dslope, grad_w, cin_real, accumulate, ws):
    """Layer 0's whole backward in one pass over dy (csrc/conv0_bwd.hip): dgamma / dbeta / dslope and the conv's weight gradient
    (OIHW fp32, the `cin_real` real input channels) without materialising dz."""
    mean, invstd, scale, shift = stats
    _lib.check(_lib.lib().ryolo_conv0_bn_bwd_wgrad(C.byref(d), x.data_ptr(), packed_w.data_ptr(), dy.data_ptr(), dy.stride(2),
                                                   scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(), act,
                                                   slope.data_ptr() if slope is not None else None, dgamma.data_ptr(), dbeta.data_ptr(),
                                                   dslope.data_ptr() if dslope is not None else None, grad_w.data_ptr(), int(cin_real),
                                                   1 if accumulate else 0, ws.data_ptr(), ws.numel(), _s(x.device)), "ryolo_conv0_bn_bwd_wgrad")


def conv_fwd_plain(d, x, packed_w, ones, shift, z):
    """z = conv(x, W) + shift (linear), no statistics (the bias convs in front of the yolo layers)."""
    d.out_cstride = _check_nhwc(z, "z")
    _lib.check(_lib.lib().ryolo_conv2d_bn_act(C.byref(d), x.data_ptr(), packed_w.data_ptr(), ones.data_ptr(), shift.data_ptr(),
                                              None, z.data_ptr(), _s(x.device)), "ryolo_conv2d_bn_act")


def bn_finalize(part, C_, count, gamma, beta, eps=1e-5, momentum=0.1, running_mean=None, running_var=None, out=None):
    dev = part.device
    if out is None:
        out = [torch.empty(C_, dtype=torch.float32, device=dev) for _ in range(4)]      # mean, invstd, scale, shift
    _lib.check(_lib.lib().ryolo_bn_finalize(part.data_ptr(), part.shape[0], part.shape[2], C_, int(count), eps, momentum,
                                            gamma.data_ptr(), beta.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                            out[2].data_ptr(), out[3].data_ptr(),
                                            running_mean.data_ptr() if running_mean is not None else None,
                                            running_var.data_ptr() if running_var is not None else None, _s(dev)),
               "ryolo_bn_finalize")
    return out


def bn_act_fwd(z, scale, shift, act, slope, y, residual=None):
    n, h, w, c = z.shape
    _lib.check(_lib.lib().ryolo_bn_act_fwd(z.data_ptr(), z.stride(2), scale.data_ptr(), shift.data_ptr(), act,
                                           slope.data_ptr() if slope is not None else None,
                                           residual.data_ptr() if residual is not None else None,
                                           residual.stride(2) if residual is not None else 0, y.data_ptr(), y.stride(2),
                                           n * h * w, c, _s(z.device)), "ryolo_bn_act_fwd")
    return y


def bn_act_bwd(z, dy, stats, act, slope, dz, dgamma, dbeta, dslope, ws):
    """stats = (mean, invstd, scale, shift) or None for a bias conv (then only dbeta (= dbias) is accumulated)."""
    n, h, w, c = z.shape
    mean, invstd, scale, shift = stats if stats is not None else (None, None, None, None)
    p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    _lib.check(_lib.lib().ryolo_bn_act_bwd(z.data_ptr(), z.stride(2), dy.data_ptr(), dy.stride(2), p(scale), p(shift), p(mean),
                                           p(invstd), act, p(slope), p(dz), dz.stride(2) if dz is not None else 0,
                                           n * h * w, c, p(dgamma), p(dbeta), p(dslope), ws.data_ptr(), ws.numel(),
                                           _s(z.device)), "ryolo_bn_act_bwd")


def bn_bwd_ws_bytes(npix, c):
    return _lib.lib().ryolo_bn_act_bwd_workspace_bytes(npix, c)


_tap_tables = {}


def dgrad_tap_table(ksize, stride, device):
    key = (ksize, stride, device.index)
    t = _tap_tables.get(key)
    if t is None:
        host = (C.c_int * 72)()
        _lib.check(_lib.lib().ryolo_conv_dgrad_tap_table(ksize, stride, host), "ryolo_conv_dgrad_tap_table")
        t = torch.tensor(list(host), dtype=torch.int32, device=device)
        _tap_tables[key] = t
    return t


def pack_weights_dgrad(weight, stride, out=None):
    cout, cin, k, _ = weight.shape
    L = _lib.lib()
    nbytes = L.ryolo_conv_packed_dgrad_bytes(cout, cin, k, stride)
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    w = weight.contiguous()
    table = dgrad_tap_table(k, stride, weight.device)
    _lib.check(L.ryolo_conv_pack_weights_dgrad(w.data_ptr(), cout, cin, k, stride, out.data_ptr(), table.data_ptr(),
                                               _s(weight.device)), "ryolo_conv_pack_weights_dgrad")
    return out


def conv_dgrad(d, dz, packed_dgrad, ones, zeros, dx, accumulate):
    _lib.check(_lib.lib().ryolo_conv2d_dgrad(C.byref(d), dz.data_ptr(), dz.stride(2), packed_dgrad.data_ptr(), ones.data_ptr(),
                                             zeros.data_ptr(), dx.data_ptr(), 1 if accumulate else 0, _s(dz.device)),
               "ryolo_conv2d_dgrad")


def dgrad_bnreduce_rows(d):
    """rows of partial sums conv_dgrad_bnreduce writes for this (forward) conv, 0 = its data gradient cannot carry the reduce"""
    return _lib.lib().ryolo_conv2d_dgrad_bnreduce_rows(C.byref(d))


def conv_dgrad_bnreduce(d, dz, packed_dgrad, ones, zeros, dx, accumulate, z, stats, slope, part):
    """conv_dgrad that also runs the reduce pass of the BatchNorm/PReLU backward of the block that produced this conv's input
    (z = that block's conv output, stats = its (mean, invstd, scale, shift)) on the final dx: part [rows, 3, C_in] fp32."""
    mean, invstd, scale, shift = stats
    _lib.check(_lib.lib().ryolo_conv2d_dgrad_bnreduce(C.byref(d), dz.data_ptr(), dz.stride(2), packed_dgrad.data_ptr(), ones.data_ptr(),
                                                      zeros.data_ptr(), dx.data_ptr(), 1 if accumulate else 0, z.data_ptr(), z.stride(2),
                                                      scale.data_ptr(), shift.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                                      slope.data_ptr(), part.data_ptr(), _s(dz.device)), "ryolo_conv2d_dgrad_bnreduce")


def bn_act_bwd_reduced(z, dy, stats, act, slope, dz, dgamma, dbeta, dslope, part, ws):
    """bn_act_bwd whose reduce pass already ran inside conv_dgrad_bnreduce (part [rows, 3, C]); ws: >= 3*C floats."""
    n, h, w, c = z.shape
    mean, invstd, scale, shift = stats
    p = lambda t: t.data_ptr() if t is not None else None   # noqa: E731
    _lib.check(_lib.lib().ryolo_bn_act_bwd_reduced(z.data_ptr(), z.stride(2), dy.data_ptr(), dy.stride(2), scale.data_ptr(), shift.data_ptr(),
                                                   mean.data_ptr(), invstd.data_ptr(), act, p(slope), dz.data_ptr(), dz.stride(2), n * h * w, c,
                                                   p(dgamma), p(dbeta), p(dslope), part.data_ptr(), part.shape[0], ws.data_ptr(), ws.numel(),
                                                   _s(z.device)), "ryolo_bn_act_bwd_reduced")


def wgrad_ws_bytes(d):
    return _lib.lib().ryolo_conv_wgrad_workspace_bytes(C.byref(d))


def conv_wgrad(d, x, dz, cin_real, grad, accumulate, ws):
    L = _lib.lib()
    args = (C.byref(d), x.data_ptr(), dz.data_ptr(), dz.stride(2), cin_real, grad.data_ptr(), 1 if accumulate else 0, ws.data_ptr(),
            ws.numel(), _s(x.device))
    if isinstance(L, _lib._CallTracer):          # bench.py's traced steps: the tile kernel and the split-K reduce as two timed calls
        _lib.check(L.ryolo_conv2d_wgrad_partials(*args), "ryolo_conv2d_wgrad_partials")
        _lib.check(L.ryolo_conv2d_wgrad_reduce(*args), "ryolo_conv2d_wgrad_reduce")
        return
    _lib.check(L.ryolo_conv2d_wgrad(*args), "ryolo_conv2d_wgrad")


def conv_wgrad_partials(d, x, dz, cin_real, grad, accumulate, ws):
    """the tile kernel of conv_wgrad only: the layer's split-K partial tiles stay in `ws` for a later (batched) reduce"""
    _lib.check(_lib.lib().ryolo_conv2d_wgrad_partials(C.byref(d), x.data_ptr(), dz.data_ptr(), dz.stride(2), cin_real, grad.data_ptr(),
                                                      1 if accumulate else 0, ws.data_ptr(), ws.numel(), _s(x.device)),
               "ryolo_conv2d_wgrad_partials")


def upsample2x_bwd(dy, dx, accumulate):
    n, h, w, c = dx.shape
    _lib.check(_lib.lib().ryolo_upsample2x_bwd(dy.data_ptr(), dy.stride(2), dx.data_ptr(), dx.stride(2), n, h, w, c,
                                               1 if accumulate else 0, _s(dy.device)), "ryolo_upsample2x_bwd")


def pgrad_to_nhwc(pgrad, out):
    bs, na, ny, nx, no = pgrad.shape
    g = pgrad.contiguous()
    _lib.check(_lib.lib().ryolo_pgrad_to_nhwc(g.data_ptr(), bs, na, ny, nx, no, out.data_ptr(), out.stride(2), _s(g.device)),
               "ryolo_pgrad_to_nhwc")


def yolo_loss_bitmap(p, nc=1, arc=0):
    """Zeroed dedup bitmap for ryolo_yolo_loss[_arc] on head tensor p [bs, na, ny, nx, no]."""
    cells = p.numel() // p.shape[-1]
    return torch.zeros(_lib.lib().ryolo_yolo_loss_bitmap_bytes_arc(cells, nc, arc) // 4, dtype=torch.int32, device=p.device)


def yolo_loss_head(p, hd, nc, h, bitmap, dp, items, arc=0):
    """One head of the 'default'-arc loss + gradient (csrc/loss.hip).  hd: a dict from loss_static.build_targets_static
    (w, b, gj, gi, cls, gxy, gwh, ga, av); h: hyper-parameters; bitmap zeroed by the caller; items[0..2] accumulated."""
    bs, na, ny, nx, no = p.shape
    w = hd['w'].contiguous()
    n = hd['npos'] if 'npos' in hd else w.sum()
    c = lambda t: t.contiguous()   # noqa: E731
    b, gj, gi, cls = c(hd['b']), c(hd['gj']), c(hd['gi']), c(hd['cls'])
    txy, twh, ta, av = c(hd['gxy']), c(hd['gwh']), c(hd['ga']), c(hd['av'].float())
    _lib.check(_lib.lib().ryolo_yolo_loss_arc(p.data_ptr(), bs, na, ny, nx, no, nc, w.data_ptr(), w.shape[1], b.data_ptr(),
                                              gj.data_ptr(), gi.data_ptr(), cls.data_ptr(), txy.data_ptr(), twh.data_ptr(),
                                              ta.data_ptr(), av.data_ptr(), n.data_ptr(), float(h['giou']), float(h['reg']),
                                              float(h['cls']), float(h['cls_pw']), float(h['obj']), float(h['obj_pw']),
                                              1 if h.get('riou', 0) else 0, int(arc), float(h.get('fl_gamma', 0.0)),
                                              bitmap.data_ptr(), dp.data_ptr(), items.data_ptr(), _s(p.device)),
               "ryolo_yolo_loss_arc")


def yolo_loss_head_nhwc(head, p, hd, nc, h, bitmap, dp_sparse, head_g, items, arc=0):
    """yolo_loss_head for a head of the training engine: `head` / `head_g` are the NHWC bf16 activation and gradient buffers of
    the head conv, dp_sparse an all-zero fp32 scratch shaped like p (left all-zero); see ryolo_yolo_loss_nhwc."""
    bs, na, ny, nx, no = p.shape
    w = hd['w'].contiguous()
    n = hd['npos'] if 'npos' in hd else w.sum()
    c = lambda t: t.contiguous()   # noqa: E731
    b, gj, gi, cls = c(hd['b']), c(hd['gj']), c(hd['gi']), c(hd['cls'])
    txy, twh, ta, av = c(hd['gxy']), c(hd['gwh']), c(hd['ga']), c(hd['av'].float())
    _lib.check(_lib.lib().ryolo_yolo_loss_nhwc_arc(head.data_ptr(), head.stride(2), p.data_ptr(), bs, na, ny, nx, no, nc, w.data_ptr(),
                                                   w.shape[1], b.data_ptr(), gj.data_ptr(), gi.data_ptr(), cls.data_ptr(),
                                                   txy.data_ptr(), twh.data_ptr(), ta.data_ptr(), av.data_ptr(), n.data_ptr(),
                                                   float(h['giou']), float(h['reg']), float(h['cls']), float(h['cls_pw']),
                                                   float(h['obj']), float(h['obj_pw']), 1 if h.get('riou', 0) else 0, int(arc),
                                                   float(h.get('fl_gamma', 0.0)), bitmap.data_ptr(), dp_sparse.data_ptr(),
                                                   head_g.data_ptr(), head_g.stride(2), items.data_ptr(), _s(p.device)),
               "ryolo_yolo_loss_nhwc_arc")


def scale_bf16_if(g, buf):
    """buf (NHWC bf16) *= g[0] unless g[0] == 1, decided on the device."""
    n, hh, ww, cc = buf.shape
    _lib.check(_lib.lib().ryolo_scale_bf16_if(g.data_ptr(), buf.data_ptr(), buf.stride(2), n * hh * ww, cc, _s(buf.device)),
               "ryolo_scale_bf16_if")


class RotatedIoU(torch.autograd.Function):
    """iou[n] = rotated IoU(pbox[n,5], tbox[n,5]) with the polygon-overlap gradient w.r.t. pbox (csrc/riou_grad.h); tbox is
    a constant.  CUDA fp32 tensors only -- there is no CPU path."""

    @staticmethod
    def forward(ctx, pbox, tbox):
        if not pbox.is_cuda:
            raise RuntimeError("RotatedIoU runs on the HIP kernel only (got a CPU tensor)")
        p = pbox.detach().float().contiguous()
        t = tbox.detach().float().contiguous()
        n = p.shape[0]
        iou = torch.empty(n, dtype=torch.float32, device=p.device)
        grad = torch.empty(n, 5, dtype=torch.float32, device=p.device)
        _lib.check(_lib.lib().ryolo_riou_loss_pairs(p.data_ptr(), t.data_ptr(), n, iou.data_ptr(), grad.data_ptr(), _s(p.device)),
                   "ryolo_riou_loss_pairs")
        ctx.save_for_backward(grad)
        return iou

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.unsqueeze(1), None


_lib.declare("ryolo_build_targets", C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_float, C.c_float, C.c_float, _vp, _vp,
                                              _vp, _vp, _vp])


class BuildTargets(object):
    """build_targets for all heads in one launch (csrc/loss.hip), writing into static buffers; `heads()` returns the dicts
    loss_static.build_targets_static returns (views of those buffers), `npos[h]` the positives' count."""

    def __init__(self, core, capacity, device):
        self.NT, self.dev = int(capacity), device
        self.layers = [core.module_list[i] for i in core.yolo_layers]
        self.nh, self.na = len(self.layers), int(self.layers[0].anchor_vec.shape[0])
        f32 = dict(dtype=torch.float32, device=device)
        self.ng = [l.ng.to(**f32).contiguous() for l in self.layers]
        self.av = [l.anchor_vec.to(**f32).contiguous() for l in self.layers]
        self.w = [torch.zeros(self.na, self.NT, **f32) for _ in self.layers]
        self.idx = [torch.zeros(4, self.NT, dtype=torch.int64, device=device) for _ in self.layers]
        self.box = [torch.zeros(5 * self.NT, **f32) for _ in self.layers]
        self.npos = torch.zeros(self.nh, **f32)
        arr = lambda ts: (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])   # noqa: E731
        self._ptrs = [arr(self.ng), arr(self.av), arr(self.w), arr(self.idx), arr(self.box),
                      arr([self.npos[h:h + 1] for h in range(self.nh)])]

    def run(self, tpad, valid, h, context_factor):
        self.npos.zero_()
        p = self._ptrs
        _lib.check(_lib.lib().ryolo_build_targets(tpad.data_ptr(), valid.data_ptr(), self.NT, self.nh, self.na, p[0], p[1],
                                                  float(h['iou_t']), float(h['ang_t']), float(context_factor), p[2], p[3], p[4],
                                                  p[5], _s(self.dev)), "ryolo_build_targets")

    def heads(self):
        NT = self.NT
        return [dict(w=self.w[k], b=self.idx[k][0], cls=self.idx[k][1], gj=self.idx[k][2], gi=self.idx[k][3],
                     gxy=self.box[k][:2 * NT].view(NT, 2), gwh=self.box[k][2 * NT:4 * NT].view(NT, 2), ga=self.box[k][4 * NT:],
                     av=self.av[k], npos=self.npos[k:k + 1]) for k in range(self.nh)]
