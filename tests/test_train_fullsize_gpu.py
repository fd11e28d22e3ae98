"""GPU tier: the training-step convolution kernels at the FULL 608 x 608 geometry of Darknet-53 -- the
dispatch of the step BASELINE configs[3] is quoted on is size dependent (conv_mp 256 / 192 rows vs conv_mq, depth of the
persistent grids, wgrad split counts and the wide wgrad tile, the x-fused stride-2 data gradient of the stem, the 1024-split
first-layer wgrad), and none of those choices is taken by the small shapes of tests/test_train_ops_gpu.py.

For every distinct (k, stride, C_in, C_out, H_out) of the 75 convs (reference step: train.py:268-282, operators
model/models.py:49-66):
  bs 4    z and the statistics sums of the training forward, the data gradient and the weight gradient against fp32 ATen /
          autograd on the same bf16-representable inputs (CPU), bars as in test_train_ops_gpu.py;
  bs 64   (the batch of configs[3]) through a size-independent property: a convolution treats images independently, so on a batch
          that repeats the 4 checked images 16 times  z / dx must repeat the bs-4 result (2 bf16 ulp: another tile may sum in
          another order), the statistics sums must be 16 x the bs-4 sums, and dW (a sum over the batch) 16 x the bs-4 dW.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (k, stride, cin, cout, Hout) -- tools/layer_bench.py SHAPES = SURVEY.md Appendix A; C_in 3 is the 8-channel padded stem
SHAPES = [(3, 1, 3, 32, 608), (1, 1, 64, 32, 304), (3, 1, 32, 64, 304), (3, 2, 32, 64, 304),
          (1, 1, 128, 64, 152), (3, 1, 64, 128, 152), (3, 2, 64, 128, 152),
          (1, 1, 256, 128, 76), (1, 1, 256, 504, 76), (1, 1, 384, 128, 76), (3, 1, 128, 256, 76),
          (3, 2, 128, 256, 76), (1, 1, 256, 128, 38), (1, 1, 512, 256, 38), (1, 1, 512, 504, 38),
          (1, 1, 768, 256, 38), (3, 1, 256, 512, 38), (3, 2, 256, 512, 38), (1, 1, 512, 256, 19),
          (1, 1, 1024, 504, 19), (1, 1, 1024, 512, 19), (3, 1, 512, 1024, 19), (3, 2, 512, 1024, 19)]


@pytest.fixture(scope="module")
def T(cuda_dev):
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.model import hip_ops, hip_train_ops

    class NS:
        ops, tr = hip_ops, hip_train_ops
    return NS


def r16(t):
    return t.to(torch.bfloat16).float()


def nhwc(t, dev):
    return t.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)


def nchw(t):
    return t.float().cpu().permute(0, 3, 1, 2)


def _hip_pass(T, dev, xd, wt, dzd, cin_real, k, s, with_dgrad):
    """training forward (z + statistics), data gradient, weight gradient of one conv on the HIP kernels"""
    n, h, w, cin_k = xd.shape
    cout = wt.shape[0]
    pad = (k - 1) // 2
    ho = (h + 2 * pad - k) // s + 1
    d = T.tr.make_desc(xd, cout, k, s, pad)
    packed = T.ops.pack_weights(wt.to(dev), cin_pad=cin_k)
    ones = torch.ones(T.ops.cpad(max(cout, cin_k)), device=dev)
    zeros = torch.zeros(T.ops.cpad(max(cout, cin_k)), device=dev)
    z = torch.empty(n, ho, ho, cout, dtype=torch.bfloat16, device=dev)
    part = T.tr.conv_fwd_stats(d, xd, packed, ones, zeros, z)
    s1 = part[:, 0, :cout].sum(0).float().cpu()
    s2 = part[:, 1, :cout].sum(0).float().cpu()
    dx = None
    if with_dgrad:
        pk = T.tr.pack_weights_dgrad(wt.to(dev), s)
        dx = torch.full((n, h, w, cin_k), 7.0, dtype=torch.bfloat16, device=dev)
        T.tr.conv_dgrad(d, dzd, pk, ones, zeros, dx, False)
    ws = torch.empty(T.tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=dev)
    grad = torch.zeros(wt.shape, device=dev)
    T.tr.conv_wgrad(d, xd, dzd, cin_real, grad, True, ws)
    torch.cuda.synchronize()
    return z, s1, s2, dx, grad.cpu()


@pytest.mark.parametrize("k,s,cin,cout,ho", SHAPES)
def test_training_convs_at_608_geometry(T, cuda_dev, k, s, cin, cout, ho):
    g = torch.Generator().manual_seed(1000 + k * 7 + s * 3 + cin + cout + ho)
    n, hin, pad = 4, ho * s, (k - 1) // 2
    cin_k = 8 if cin == 3 else cin
    x = r16(torch.randn(n, cin, hin, hin, generator=g))
    wt = r16(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)
    xin = torch.zeros(n, cin_k, hin, hin)
    xin[:, :cin] = x
    dz = r16(torch.randn(n, cout, ho, ho, generator=g))
    with_dgrad = cin != 3                      # the stem has no data gradient
    # ---- reference: fp32 ATen / autograd on the same values
    torch.set_num_threads(max(1, torch.get_num_threads()))
    xr = x.clone().requires_grad_(with_dgrad)
    wr = wt.clone().requires_grad_(True)
    zr = F.conv2d(xr, wr, None, stride=s, padding=pad)
    zr.backward(dz)
    zr = zr.detach()
    # ---- bs 4 on the HIP kernels
    xd, dzd = nhwc(xin, cuda_dev), nhwc(dz, cuda_dev)
    z4, s1, s2, dx4, dw4 = _hip_pass(T, cuda_dev, xd, wt, dzd, cin, k, s, with_dgrad)
    zq = nchw(z4)
    assert torch.allclose(zq, r16(zr), rtol=2 ** -7, atol=2e-3), float((zq - r16(zr)).abs().max())
    assert torch.allclose(s1, zq.sum((0, 2, 3)), rtol=1e-4, atol=5e-2)              # statistics of the values as stored
    assert torch.allclose(s2, (zq * zq).sum((0, 2, 3)), rtol=1e-4, atol=5e-2)
    if with_dgrad:
        err = (nchw(dx4)[:, :cin] - r16(xr.grad)).abs()
        assert bool((err <= 2 ** -7 * xr.grad.abs() + 3e-3).all()), float(err.max())
    err = (dw4 - wr.grad).abs().max()
    assert float(err) <= 2e-3 * float(wr.grad.abs().max()) + 1e-3, (float(err), float(wr.grad.abs().max()))
    # ---- bs 64 = the same four images sixteen times: the dispatch of configs[3]
    rep = 16
    xd64, dzd64 = xd.repeat(rep, 1, 1, 1), dzd.repeat(rep, 1, 1, 1)
    z64, t1, t2, dx64, dw64 = _hip_pass(T, cuda_dev, xd64, wt, dzd64, cin, k, s, with_dgrad)
    zz = z64.view(rep, n, ho, ho, cout).float()
    ref = z4.float().unsqueeze(0)
    assert bool(((zz - ref).abs() <= 2 ** -7 * ref.abs() + 2e-3).all()), float((zz - ref).abs().max())
    zs = z64.float()
    assert torch.allclose(t1, zs.sum((0, 1, 2)).cpu(), rtol=1e-4, atol=0.5)
    assert torch.allclose(t2, (zs * zs).sum((0, 1, 2)).cpu(), rtol=1e-4, atol=0.5)
    if with_dgrad:
        dd = dx64.view(rep, n, hin, hin, cin_k).float()
        ref = dx4.float().unsqueeze(0)
        assert bool(((dd - ref).abs() <= 2 ** -7 * ref.abs() + 3e-3).all()), float((dd - ref).abs().max())
    err = (dw64 - rep * dw4).abs().max()
    assert float(err) <= 2e-3 * rep * float(dw4.abs().max()) + 1e-3 * rep, (float(err), float(dw4.abs().max()))
