"""Python-side handles on the C ABI's convolution block and layout converters (include/ryolo.h).

Plumbing only: torch supplies device memory and the current stream; all arithmetic happens in
csrc/conv.hip.  Tensors here are NHWC bf16 `torch.Tensor`s of shape [N, H, W, C] whose last-dim stride is 1
and whose pixel stride (stride(2)) may exceed C (a channel slice of a wider concat buffer).
"""
import ctypes as C

import torch

from .. import _lib

ACT_LINEAR, ACT_LEAKY, ACT_MISH = 0, 1, 2


class ConvDesc(C.Structure):
    _fields_ = [("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("Cin", C.c_int), ("Cout", C.c_int),
                ("ksize", C.c_int), ("stride", C.c_int), ("pad", C.c_int),
                ("in_cstride", C.c_int), ("out_cstride", C.c_int), ("res_cstride", C.c_int),
                ("act", C.c_int), ("slope", C.c_float), ("upsample", C.c_int), ("tile", C.c_int)]


_vp = C.c_void_p
_lib.declare("ryolo_conv_packed_weight_bytes", C.c_size_t, [C.c_int, C.c_int, C.c_int])
_lib.declare("ryolo_conv_pack_weights", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp])
_lib.declare("ryolo_conv2d_bn_act", C.c_int, [C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp])
_lib.declare("ryolo_conv_pair_supported", C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), C.c_int])
_lib.declare("ryolo_conv2d_bn_act_pair", C.c_int, [C.POINTER(ConvDesc), C.POINTER(ConvDesc), _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, _vp, _vp])
_lib.declare("ryolo_conv_kernel_choice", C.c_int, [C.POINTER(ConvDesc), C.c_int, C.c_int])
_lib.declare("ryolo_conv_dgrad_kernel_choice", C.c_int, [C.POINTER(ConvDesc), C.c_int])
_lib.declare("ryolo_conv_wgrad_kernel_choice", C.c_int, [C.POINTER(ConvDesc)])
_lib.declare("ryolo_nchw_f32_to_nhwc_bf16", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp])
_lib.declare("ryolo_nhwc_bf16_to_nchw_f32", C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp])


def cpad(c, m=128):
    return (c + m - 1) // m * m


def pack_weights(weight, cin_pad=None, out=None):
    """weight: fp32 [Cout, Cin, k, k] on the GPU (nn.Conv2d.weight layout) -> packed bf16 image (uint8 tensor)."""
    assert weight.is_cuda and weight.dtype == torch.float32 and weight.dim() == 4
    cout, cin, k, k2 = weight.shape
    assert k == k2
    cin_pad = cin_pad or (cin + 7) // 8 * 8
    L = _lib.lib()
    nbytes = L.ryolo_conv_packed_weight_bytes(cout, cin_pad, k)
    if nbytes == 0:
        raise RuntimeError("unsupported conv weight shape %s" % (tuple(weight.shape),))
    w = weight.contiguous()
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    with torch.cuda.device(weight.device):
        _lib.check(L.ryolo_conv_pack_weights(w.data_ptr(), cout, cin, k, cin_pad, out.data_ptr(),
                                             _lib.stream_ptr(weight.device)), "ryolo_conv_pack_weights")
    return out


def pad_vec(v, n, fill=0.0):
    out = torch.full((n,), fill, dtype=torch.float32, device=v.device)
    out[:v.numel()] = v.float()
    return out


def _check_nhwc(t, name):
    if not (t.is_cuda and t.dtype == torch.bfloat16 and t.dim() == 4 and t.stride(3) == 1):
        raise RuntimeError("%s must be an NHWC bf16 GPU tensor with unit channel stride" % name)
    n, h, w, c = t.shape
    cs = t.stride(2)
    if t.stride(1) != w * cs or t.stride(0) != h * w * cs:
        raise RuntimeError("%s: pixels must be densely packed rows of `pixel stride` elements" % name)
    return cs


def conv2d_bn_act(x, packed_w, scale, shift, cout, ksize, stride=1, pad=None, act=ACT_LINEAR, slope=0.1,
                  residual=None, out=None, upsample=1, tile=0):
    """y = upsample(act(conv(x, W) * scale + shift) + residual); x, residual, out: NHWC bf16 (possibly slices)."""
    in_cs = _check_nhwc(x, "x")
    n, h, w, cin = x.shape
    pad = (ksize - 1) // 2 if pad is None else pad
    ho = (h + 2 * pad - ksize) // stride + 1
    wo = (w + 2 * pad - ksize) // stride + 1
    if out is None:
        out = torch.empty((n, ho * upsample, wo * upsample, cout), dtype=torch.bfloat16, device=x.device)
    out_cs = _check_nhwc(out, "out")
    assert tuple(out.shape) == (n, ho * upsample, wo * upsample, cout), (tuple(out.shape), (n, ho, wo, cout))
    res_cs = 0
    if residual is not None:
        res_cs = _check_nhwc(residual, "residual")
        assert tuple(residual.shape) == (n, ho, wo, cout)
    d = ConvDesc(n, h, w, cin, cout, ksize, stride, pad, in_cs, out_cs, res_cs, act, float(slope), upsample, tile)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ryolo_conv2d_bn_act(C.byref(d), x.data_ptr(), packed_w.data_ptr(), scale.data_ptr(),
                                            shift.data_ptr(), residual.data_ptr() if residual is not None else None,
                                            out.data_ptr(), _lib.stream_ptr(x.device))
    _lib.check(rc, "ryolo_conv2d_bn_act")
    return out


def pair_descs(x, first, second, out_cs=None):
    """ConvDesc pair for conv2d_bn_act_pair; first / second = dicts(cout, ksize, stride, pad, act, slope)"""
    in_cs = _check_nhwc(x, "x")
    n, h, w, cin = x.shape
    a = ConvDesc(n, h, w, cin, first['cout'], first['ksize'], first['stride'], first['pad'], in_cs, first['cout'], 0, first['act'],
                 float(first['slope']), 1, 0)
    h1 = (h + 2 * first['pad'] - first['ksize']) // first['stride'] + 1
    w1 = (w + 2 * first['pad'] - first['ksize']) // first['stride'] + 1
    b = ConvDesc(n, h1, w1, first['cout'], second['cout'], second['ksize'], second['stride'], second['pad'], first['cout'],
                 out_cs or second['cout'], 0, second['act'], float(second['slope']), 1, 0)
    return a, b


def conv_pair_supported(x, first, second, shortcut_from_input):
    a, b = pair_descs(x, first, second)
    return bool(_lib.lib().ryolo_conv_pair_supported(C.byref(a), C.byref(b), 1 if shortcut_from_input else 0))


def conv2d_bn_act_pair(x, first, second, packed1, scale1, shift1, packed2, scale2, shift2, shortcut_from_input=False, out=None):
    """y = block2(block1(x)) [+ x]: two conv blocks in one launch, the tensor between them never stored (csrc/conv_stem.hip)."""
    n, h, w, cin = x.shape
    h1 = (h + 2 * first['pad'] - first['ksize']) // first['stride'] + 1
    w1 = (w + 2 * first['pad'] - first['ksize']) // first['stride'] + 1
    h2 = (h1 + 2 * second['pad'] - second['ksize']) // second['stride'] + 1
    w2 = (w1 + 2 * second['pad'] - second['ksize']) // second['stride'] + 1
    if out is None:
        out = torch.empty((n, h2, w2, second['cout']), dtype=torch.bfloat16, device=x.device)
    out_cs = _check_nhwc(out, "out")
    assert tuple(out.shape) == (n, h2, w2, second['cout'])
    a, b = pair_descs(x, first, second, out_cs)
    with torch.cuda.device(x.device):
        rc = _lib.lib().ryolo_conv2d_bn_act_pair(C.byref(a), C.byref(b), x.data_ptr(), packed1.data_ptr(), scale1.data_ptr(), shift1.data_ptr(),
                                                 packed2.data_ptr(), scale2.data_ptr(), shift2.data_ptr(), 1 if shortcut_from_input else 0,
                                                 out.data_ptr(), _lib.stream_ptr(x.device))
    _lib.check(rc, "ryolo_conv2d_bn_act_pair")
    return out


_IGEMM_TILES = {1: '128x128', 2: '256x64', 3: '256x32', 4: '256x128', 6: '128x128(4x2)', 7: '128x128(2x2)'}


def conv_kernel_name(n, h, w, cin, cout, ksize, stride=1, pad=None, in_cs=None, out_cs=None, res_cs=0, upsample=1, tile=0,
                     residual=False, statistics=False):
    """Name of the kernel the library's dispatch picks for this conv on the current device (a dry run, nothing is launched):
    what the per-kernel tables of bench.py and the profiles call it."""
    pad = (ksize - 1) // 2 if pad is None else pad
    d = ConvDesc(n, h, w, cin, cout, ksize, stride, pad, in_cs or cin, out_cs or cout, res_cs or (cout if residual else 0), 1, 0.1,
                 upsample, tile)
    code = _lib.lib().ryolo_conv_kernel_choice(C.byref(d), 1 if residual else 0, 1 if statistics else 0)
    return kernel_name_of(code, ksize, stride, cin)


def kernel_name_of(code, ksize, stride, cin):
    """the name bench.py / the profiles use for a RYOLO_CONV_KERNEL_* code (include/ryolo.h)"""
    if code in (1, 2):
        return 'conv_mp<k%d,%dx256>' % (ksize, 256 if code == 1 else 192)
    if code == 3:
        return 'conv_mq<k%d,128x256>' % ksize
    if code == 4:
        return 'conv3x3_c8_direct'
    if code == 5:
        return 'conv3x3_c32_halo<s%d>' % stride
    if code == 6:
        return 'conv_pw<k1,K%d>' % cin
    if code == 7:
        return 'conv_stem_dgrad'
    if code == 8:
        return 'conv0_halo<c8>'
    if code == 11:
        return 'conv3x3_c64_halo'
    if code in (9, 10):
        return 'conv_mq<k%d,%dx128>' % (ksize, 128 if code == 9 else 64)
    if code >= 16:
        return 'conv_igemm<k%d,%s>' % (ksize, _IGEMM_TILES.get(code - 16, 'tile%d' % (code - 16)))
    return 'conv<?>'


def nchw_f32_to_nhwc_bf16(x, cpad_to=8):
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4
    x = x.contiguous()
    n, c, h, w = x.shape
    cp = (c + cpad_to - 1) // cpad_to * cpad_to
    y = torch.empty((n, h, w, cp), dtype=torch.bfloat16, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ryolo_nchw_f32_to_nhwc_bf16(x.data_ptr(), n, c, h, w, cp, y.data_ptr(),
                                                          _lib.stream_ptr(x.device)), "ryolo_nchw_f32_to_nhwc_bf16")
    return y


def nhwc_bf16_to_nchw_f32(x):
    cs = _check_nhwc(x, "x")
    n, h, w, c = x.shape
    y = torch.empty((n, c, h, w), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.check(_lib.lib().ryolo_nhwc_bf16_to_nchw_f32(x.data_ptr(), n, c, h, w, cs, y.data_ptr(),
                                                          _lib.stream_ptr(x.device)), "ryolo_nhwc_bf16_to_nchw_f32")
    return y
