"""GPU tier: the training-step kernels (conv statistics, BatchNorm+PReLU fwd/bwd, dgrad, wgrad) vs torch autograd in
fp32 on the same bf16-representable inputs.  Tolerances: bf16 outputs 2 ulp of the contributing magnitudes; fp32 weight
gradients relative to the gradient norm (sums over up to 1e5 pixels of bf16 products, fp32 accumulation)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


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


def _setup(n, cin, cout, h, w, k, seed, real_cin=None):
    g = torch.Generator().manual_seed(seed)
    real_cin = real_cin or cin
    x = r16(torch.randn(n, real_cin, h, w, generator=g))
    wt = r16(torch.randn(cout, real_cin, k, k, generator=g) / (real_cin * k * k) ** 0.5)
    return g, x, wt


@pytest.mark.parametrize("n,cin,cout,h,w,k,s", [(2, 64, 128, 20, 20, 3, 1), (3, 128, 64, 19, 19, 1, 1), (2, 64, 128, 22, 22, 3, 2),
                                                (3, 8, 32, 37, 53, 3, 1),        # the direct first-layer kernel
                                                (2, 32, 64, 21, 23, 3, 1), (2, 32, 64, 22, 26, 3, 2),    # C_in 32: two taps per K step
                                                # 256-multiple C_out: the statistics instantiation of conv_mp_kernel (stride 1 / 2,
                                                # one and two channel tiles, a partial last pixel tile)
                                                (2, 128, 256, 20, 20, 3, 1), (8, 128, 256, 32, 32, 3, 2), (1, 256, 512, 19, 19, 3, 1)])
def test_conv_stats_and_bn_forward_backward(T, cuda_dev, n, cin, cout, h, w, k, s):
    g, x, wt = _setup(n, cin, cout, h, w, k, 1)
    pad = (k - 1) // 2
    gamma = torch.rand(cout, generator=g) + 0.5
    beta = torch.randn(cout, generator=g) * 0.3
    slope = torch.tensor([0.1])
    xd = nhwc(x, cuda_dev)
    packed = T.ops.pack_weights(wt.to(cuda_dev), cin_pad=cin)
    ones = torch.ones(T.ops.cpad(cout), device=cuda_dev)
    zeros = torch.zeros(T.ops.cpad(cout), device=cuda_dev)
    d = T.tr.make_desc(xd, cout, k, s, pad)
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    z = torch.empty(n, ho, wo, cout, dtype=torch.bfloat16, device=cuda_dev)
    part = T.tr.conv_fwd_stats(d, xd, packed, ones, zeros, z)
    zr = F.conv2d(x, wt, None, stride=s, padding=pad)
    zq = nchw(z)
    assert torch.allclose(zq, r16(zr), rtol=2 ** -7, atol=2e-3)
    M = n * ho * wo
    assert part.dtype == torch.float64                      # 64-bit atomics: the arrival order of the waves does not show
    s1 = part[:, 0, :cout].sum(0).float().cpu()
    s2 = part[:, 1, :cout].sum(0).float().cpu()
    assert torch.allclose(s1, zq.sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    assert torch.allclose(s2, (zq * zq).sum((0, 2, 3)), rtol=1e-4, atol=1e-2)
    rm = torch.zeros(cout, device=cuda_dev)
    rv = torch.ones(cout, device=cuda_dev)
    mean, invstd, scale, shift = T.tr.bn_finalize(part, cout, M, gamma.to(cuda_dev), beta.to(cuda_dev), running_mean=rm,
                                                  running_var=rv)
    # reference BatchNorm (training) + PReLU + residual on the SAME stored z
    zt = zq.clone().requires_grad_(True)
    gam, bet, slp = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True), slope.clone().requires_grad_(True)
    rm_r, rv_r = torch.zeros(cout), torch.ones(cout)
    res = r16(torch.randn(n, cout, ho, wo, generator=g))
    u = F.batch_norm(zt, rm_r, rv_r, gam, bet, True, 0.1, 1e-5)
    y_ref = F.prelu(u, slp)
    assert torch.allclose(rm.cpu(), rm_r, rtol=1e-4, atol=1e-5) and torch.allclose(rv.cpu(), rv_r, rtol=1e-4, atol=1e-5)
    y = torch.empty_like(z)
    T.tr.bn_act_fwd(z, scale, shift, 1, slope.to(cuda_dev), y, residual=nhwc(res, cuda_dev))
    assert torch.allclose(nchw(y), r16(r16(y_ref.detach()) + res), rtol=2 ** -6, atol=1e-2)
    # backward
    dy = r16(torch.randn(n, cout, ho, wo, generator=g))
    y_ref.backward(dy)
    dz = torch.empty_like(z)
    dg, db, dsl = torch.zeros(cout, device=cuda_dev), torch.zeros(cout, device=cuda_dev), torch.zeros(1, device=cuda_dev)
    ws = torch.empty(T.tr.bn_bwd_ws_bytes(M, cout), dtype=torch.uint8, device=cuda_dev)
    T.tr.bn_act_bwd(z, nhwc(dy, cuda_dev), (mean, invstd, scale, shift), 1, slope.to(cuda_dev), dz, dg, db, dsl, ws)
    torch.cuda.synchronize()
    gz = zt.grad
    err = (nchw(dz) - gz).abs()
    assert float(err.max()) <= 2 ** -6 * float(gz.abs().max()) + 1e-3, float(err.max())
    assert torch.allclose(dg.cpu(), gam.grad, rtol=2e-3, atol=2e-2)
    assert torch.allclose(db.cpu(), bet.grad, rtol=2e-3, atol=2e-2)
    assert torch.allclose(dsl.cpu(), slp.grad, rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("n,c,h,w", [(2, 64, 20, 20), (3, 56, 13, 17), (1, 256, 19, 19)])
def test_bn_mish_forward_backward_vs_autograd(T, cuda_dev, n, c, h, w):
    """act code 2 (Mish, the north star's activation): y = u*tanh(softplus(u)) on the BatchNorm output u, + residual; backward
    through Mish and the batch statistics.  Reference: fp32 autograd on the SAME stored bf16 z."""
    g = torch.Generator().manual_seed(11)
    zq = r16(torch.randn(n, c, h, w, generator=g) * 1.5 + 0.2)
    z = nhwc(zq, cuda_dev)
    gamma = torch.rand(c, generator=g) + 0.5
    beta = torch.randn(c, generator=g) * 0.3
    M = n * h * w
    mean = zq.mean((0, 2, 3))
    var = zq.var((0, 2, 3), unbiased=False)
    invstd = (var + 1e-5).rsqrt()
    scale = gamma * invstd
    shift = beta - mean * scale
    dev = lambda t: t.to(cuda_dev).contiguous()      # noqa: E731
    zt = zq.clone().requires_grad_(True)
    gam, bet = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    u = F.batch_norm(zt, None, None, gam, bet, True, 0.1, 1e-5)
    y_ref = u * torch.tanh(F.softplus(u))
    res = r16(torch.randn(n, c, h, w, generator=g))
    y = torch.empty_like(z)
    T.tr.bn_act_fwd(z, dev(scale), dev(shift), 2, None, y, residual=nhwc(res, cuda_dev))
    assert torch.allclose(nchw(y), r16(r16(y_ref.detach()) + res), rtol=2 ** -6, atol=1e-2)
    dy = r16(torch.randn(n, c, h, w, generator=g))
    y_ref.backward(dy)
    dz = torch.empty_like(z)
    dg, db = torch.zeros(c, device=cuda_dev), torch.zeros(c, device=cuda_dev)
    ws = torch.empty(T.tr.bn_bwd_ws_bytes(M, c), dtype=torch.uint8, device=cuda_dev)
    T.tr.bn_act_bwd(z, nhwc(dy, cuda_dev), (dev(mean), dev(invstd), dev(scale), dev(shift)), 2, None, dz, dg, db, None, ws)
    torch.cuda.synchronize()
    gz = zt.grad
    err = (nchw(dz) - gz).abs()
    assert float(err.max()) <= 2 ** -6 * float(gz.abs().max()) + 1e-3, float(err.max())
    assert torch.allclose(dg.cpu(), gam.grad, rtol=2e-3, atol=2e-2)
    assert torch.allclose(db.cpu(), bet.grad, rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("n,cin,cout,h,w,k,s,acc", [(2, 64, 128, 20, 20, 3, 1, False), (2, 128, 64, 19, 19, 1, 1, True),
                                                    (2, 64, 128, 22, 22, 3, 2, False), (1, 32, 64, 24, 20, 3, 2, True),
                                                    (2, 32, 64, 16, 16, 3, 1, False), (2, 64, 32, 16, 16, 1, 1, False),
                                                    (1, 256, 512, 19, 19, 3, 1, True),
                                                    # stride 2 with C_in 32 / 64: the x-fused classes (even width), the classic
                                                    # four (odd width), with and without accumulation, odd height
                                                    (2, 32, 64, 26, 40, 3, 2, True), (2, 64, 128, 23, 18, 3, 2, False),
                                                    (2, 32, 64, 24, 21, 3, 2, True), (1, 64, 64, 17, 17, 3, 2, False),
                                                    # the stem's one-launch stride-2 data gradient (C_in 32, C_out 64): several 8 x 64 tiles
                                                    # per image, ragged right / bottom tiles, more workgroups than tiles and fewer
                                                    (3, 32, 64, 70, 130, 3, 2, False), (2, 32, 64, 64, 128, 3, 2, True), (1, 32, 64, 9, 67, 3, 2, False),
                                                    (6, 32, 64, 152, 152, 3, 2, False),
                                                    # ... and its stride-1 sibling (layer 3): 4 x 32 tiles, halo on every side
                                                    (3, 32, 64, 70, 130, 3, 1, True), (2, 32, 64, 9, 67, 3, 1, False), (5, 32, 64, 152, 152, 3, 1, False),
                                                    # ... and the stride-2 128 -> 64 kernel (layer 5: the waves split the output channels); stride 1 stays on the tiles
                                                    (3, 64, 128, 70, 130, 3, 2, False), (2, 64, 128, 41, 35, 3, 2, True), (3, 64, 128, 70, 50, 3, 1, True),
                                                    (2, 64, 128, 9, 67, 3, 1, False), (4, 64, 128, 152, 152, 3, 1, False),
                                                    # gradients with 128 channels (C_in 128): stride 1 with and without accumulation, the four
                                                    # stride-2 parity classes (strided placement)
                                                    (2, 128, 256, 24, 20, 3, 1, True), (4, 128, 256, 76, 76, 3, 1, False),
                                                    (2, 128, 256, 23, 18, 3, 2, False), (3, 128, 256, 40, 40, 3, 2, True)])
def test_dgrad_vs_autograd(T, cuda_dev, n, cin, cout, h, w, k, s, acc):
    g, x, wt = _setup(n, cin, cout, h, w, k, 2)
    pad = (k - 1) // 2
    xr = x.clone().requires_grad_(True)
    zr = F.conv2d(xr, wt, None, stride=s, padding=pad)
    dz = r16(torch.randn(zr.shape, generator=g))
    zr.backward(dz)
    xd = nhwc(x, cuda_dev)
    d = T.tr.make_desc(xd, cout, k, s, pad)
    pk = T.tr.pack_weights_dgrad(wt.to(cuda_dev), s)
    ones = torch.ones(T.ops.cpad(cin), device=cuda_dev)
    zeros = torch.zeros(T.ops.cpad(cin), device=cuda_dev)
    prev = r16(torch.randn(n, cin, h, w, generator=g))
    dx = nhwc(prev, cuda_dev) if acc else torch.full((n, h, w, cin), 7.0, dtype=torch.bfloat16, device=cuda_dev)
    T.tr.conv_dgrad(d, nhwc(dz, cuda_dev), pk, ones, zeros, dx, acc)
    torch.cuda.synchronize()
    want = r16(r16(xr.grad) + prev) if acc else r16(xr.grad)
    mag = xr.grad.abs() + (prev.abs() if acc else 0)
    err = (nchw(dx) - want).abs()
    assert bool((err <= 2 ** -7 * mag + 3e-3).all()), float(err.max())


@pytest.mark.parametrize("n,cin,cout,h,w,k,s,real", [(2, 128, 256, 19, 19, 3, 1, None), (3, 256, 128, 19, 19, 1, 1, None),
                                                     (2, 64, 128, 22, 22, 3, 2, None), (2, 8, 32, 40, 40, 3, 1, 3),
                                                     (2, 32, 64, 20, 20, 3, 2, None), (2, 64, 32, 20, 20, 1, 1, None),
                                                     (1, 512, 504, 10, 10, 1, 1, None), (4, 64, 64, 38, 38, 3, 1, None),
                                                     # all-taps stem kernel: rows longer than one 64-pixel K step, ragged
                                                     # last segment, odd sizes under stride 2
                                                     (2, 32, 64, 70, 70, 3, 1, None), (2, 32, 64, 141, 139, 3, 2, None),
                                                     (2, 64, 32, 66, 130, 1, 1, None), (1, 32, 64, 8, 200, 3, 1, None),
                                                     (2, 8, 32, 70, 150, 3, 1, 3),
                                                     # wide-tile kernel (c_out % 256 == 0, c_in % 128 == 0): stride 2, 1x1, two
                                                     # c_in / c_out tiles, pixel counts that are not multiples of the 32-pixel step
                                                     (2, 128, 256, 38, 38, 3, 2, None), (3, 512, 256, 13, 13, 1, 1, None),
                                                     (1, 256, 512, 21, 17, 3, 1, None),
                                                     # few splits, > 1 M weights: the transposing split-K reduce (coalesced on both sides)
                                                     (1, 256, 512, 7, 9, 3, 1, None),
                                                     # round 6 (running fill addresses in the wide kernel): rows shorter than half a K step
                                                     # (several row and image wraps inside one step), stride 2 with an odd extent, rows of 16-31
                                                     (3, 128, 256, 5, 5, 3, 1, None), (5, 128, 256, 3, 4, 3, 1, None), (2, 128, 256, 11, 40, 3, 2, None),
                                                     (2, 256, 256, 33, 16, 1, 1, None), (2, 128, 256, 24, 31, 3, 1, None),
                                                     # ragged C_out on the 256 x 128 tile (the 504-channel heads; rows 504..511 of the last tile read
                                                     # the next pixel's channels and are not stored), incl. the tensor's last pixel; the 64 x 128 tile
                                                     (2, 256, 504, 19, 19, 1, 1, None), (1, 1024, 504, 5, 7, 1, 1, None), (3, 128, 504, 9, 11, 3, 1, None),
                                                     (2, 128, 64, 38, 38, 1, 1, None), (1, 256, 64, 21, 17, 1, 1, None),
                                                     # round 6: the C_in = 64 3x3 layers on the 128 x (3 taps x 64) tile -- a workgroup owns a filter
                                                     # row, every tap image of the B operand walks its own shifted pixels: rows shorter than a
                                                     # K step (several wraps per step), stride 2 with odd extents, two c_out tiles, one image
                                                     (3, 64, 128, 5, 5, 3, 1, None), (5, 64, 128, 3, 4, 3, 1, None), (2, 64, 128, 11, 40, 3, 2, None),
                                                     (1, 64, 256, 24, 31, 3, 1, None), (2, 64, 128, 33, 17, 3, 2, None), (1, 64, 128, 76, 76, 3, 1, None)])
def test_wgrad_vs_autograd(T, cuda_dev, n, cin, cout, h, w, k, s, real):
    g, x, wt = _setup(n, cin, cout, h, w, k, 3, real_cin=real)
    pad = (k - 1) // 2
    wr = wt.clone().requires_grad_(True)
    zr = F.conv2d(x, wr, None, stride=s, padding=pad)
    dz = r16(torch.randn(zr.shape, generator=g))
    zr.backward(dz)
    xin = torch.zeros(n, cin, h, w)
    xin[:, :x.shape[1]] = x
    xd = nhwc(xin, cuda_dev)
    d = T.tr.make_desc(xd, cout, k, s, pad)
    ws = torch.empty(T.tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=cuda_dev)
    grad = torch.full(wt.shape, 0.5, device=cuda_dev)
    T.tr.conv_wgrad(d, xd, nhwc(dz, cuda_dev), x.shape[1], grad, True, ws)
    torch.cuda.synchronize()
    got = grad.cpu() - 0.5
    err = (got - wr.grad).abs().max()
    assert float(err) <= 2e-3 * float(wr.grad.abs().max()) + 1e-3, (float(err), float(wr.grad.abs().max()))


@pytest.mark.parametrize("n,cin,cout,h,w,k,s", [(2, 128, 256, 19, 19, 3, 1), (2, 128, 256, 21, 18, 3, 2), (2, 256, 512, 9, 33, 1, 1),
                                                (2, 64, 128, 22, 22, 3, 1), (2, 128, 128, 20, 20, 1, 1)])
def test_wgrad_on_channel_slices_equals_the_contiguous_call(T, cuda_dev, n, cin, cout, h, w, k, s):
    """x and dz as channel slices of wider tensors (the engine's route / concat buffers: pixel stride > channel count) give the bits of
    the contiguous call -- the sliding dz descriptor and the running x offsets of the three-stage kernels take the strides from the call."""
    g = torch.Generator().manual_seed(11 + cin + cout + k + s)
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    xw = torch.randn(n, h, w, cin + 64, generator=g).to(torch.bfloat16).to(cuda_dev)
    dw = torch.randn(n, ho, wo, cout + 128, generator=g).to(torch.bfloat16).to(cuda_dev)
    xs, dzs = xw[..., 32:32 + cin], dw[..., 64:64 + cout]
    out = []
    for xv, dv in ((xs.contiguous(), dzs.contiguous()), (xs, dzs)):
        d = T.tr.make_desc(xv, cout, k, s, pad)
        ws = torch.empty(T.tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=cuda_dev)
        grad = torch.zeros(cout, cin, k, k, device=cuda_dev)
        T.tr.conv_wgrad(d, xv, dv, cin, grad, False, ws)
        torch.cuda.synchronize()
        out.append(grad.clone())
    assert torch.equal(out[0], out[1])


@pytest.mark.parametrize("n,cin,cout,h,w,k,s", [(3, 256, 128, 19, 19, 1, 1), (2, 384, 128, 21, 17, 1, 1), (2, 64, 128, 22, 22, 3, 1),
                                                (2, 64, 128, 23, 21, 3, 2), (2, 64, 256, 20, 20, 3, 1), (2, 512, 504, 19, 19, 1, 1),
                                                (2, 128, 64, 40, 40, 1, 1),
                                                # round 6: 1x1 launches of the 256 x 128 tile against the square kernel
                                                (2, 512, 256, 19, 19, 1, 1), (3, 256, 512, 13, 11, 1, 1), (1, 1024, 512, 19, 19, 1, 1)])
def test_wgrad_three_stage_tiles_equal_the_square_kernel(T, cuda_dev, n, cin, cout, h, w, k, s):
    """The three-stage (counted-wait) weight-gradient kernel on its 128 x 128 and 128 x 64 tiles against the two-stage square
    kernel it replaced (tile bit 0x2000 forces that one): the square tile keeps fragments and summation order (bit-identical),
    the 128 x 64 tile sums the same products per split in the same order over pixels (equal to fp32 rounding of the reduce)."""
    g = torch.Generator().manual_seed(5 + cin + cout + k)
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // s + 1, (w + 2 * pad - k) // s + 1
    xd = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).to(cuda_dev)
    dz = torch.randn(n, ho, wo, cout, generator=g).to(torch.bfloat16).to(cuda_dev)
    out = []
    for tile in (0, 0x2000):
        d = T.tr.make_desc(xd, cout, k, s, pad, tile=tile)
        ws = torch.empty(T.tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=cuda_dev)
        grad = torch.zeros(cout, cin, k, k, device=cuda_dev)
        T.tr.conv_wgrad(d, xd, dz, cin, grad, False, ws)
        torch.cuda.synchronize()
        out.append(grad.clone())
    if cin % 128 == 0 and cout == 128:       # the 128 x 128 three-stage tile: the square kernel's fragments, splits and summation order
        assert torch.equal(out[0], out[1])
    else:
        assert float((out[0] - out[1]).abs().max()) <= 1e-5 * float(out[1].abs().max())


def test_batched_wgrad_reduce_equals_the_per_layer_reduces(T, cuda_dev):
    """ryolo_conv_wgrad_reduce_batch (round 5: the split-K reduces of a backward segment as ONE launch over a job table, every layer's
    partial tiles in its own workspace) against ryolo_conv2d_wgrad per layer: the same bits, for every reduce kind -- four split quarters
    per workgroup (many splits), one element per thread (few splits), the transposing 3x3 variant (few splits, > 1 M weights) -- and the
    stem's per-tap kernels (unpadded partial rows); with and without accumulation into the gradient."""
    tr, dev = T.tr, cuda_dev
    g = torch.Generator().manual_seed(5)
    layers = [(2, 128, 256, 38, 38, 3, 1, None), (1, 256, 512, 7, 9, 3, 1, None), (1, 512, 504, 10, 10, 1, 1, None), (2, 32, 64, 70, 70, 3, 1, None),
              (3, 256, 128, 19, 19, 1, 1, None), (2, 8, 32, 40, 40, 3, 1, 3), (2, 64, 128, 22, 22, 3, 2, None)]
    for accumulate in (True, False):
        batch = tr.WgradReduceBatch(dev)
        want, got, kinds = [], [], set()
        for (n, cin, cout, h, w, k, s_, real) in layers:
            real = real or cin
            x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).to(dev)
            if real < cin:
                x[..., real:] = 0
            pad = (k - 1) // 2
            ho, wo = (h + 2 * pad - k) // s_ + 1, (w + 2 * pad - k) // s_ + 1
            dz = torch.randn(n, ho, wo, cout, generator=g).to(torch.bfloat16).to(dev)
            d = tr.make_desc(x, cout, k, s_, pad)
            g0 = torch.randn(cout, real, k, k, generator=g).to(dev)
            ws_a = torch.empty(tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=dev)
            ga = g0.clone()
            tr.conv_wgrad(d, x, dz, real, ga, accumulate, ws_a)
            want.append(ga)
            ws_b = torch.full((tr.wgrad_ws_bytes(d),), 0x7f, dtype=torch.uint8, device=dev)       # (its own workspace, garbage on entry)
            gb = g0.clone()
            tr.conv_wgrad_partials(d, x, dz, real, gb, accumulate, ws_b)
            assert torch.equal(gb, g0)                    # the tile kernel alone does not touch the gradient
            batch.add(d, real, ws_b, gb, accumulate)
            kinds.add(batch.jobs[-1].kind)
            got.append(gb)
        assert kinds == {0, 1, 3, 4}, kinds       # (3, 4: the batched launch's own forms of 1 and 2; 1 stays for C_in = 3)
        batch.finalize()
        batch.run()
        torch.cuda.synchronize()
        for a, b in zip(want, got):
            assert torch.equal(a, b)
        batch.run()                                        # the partials are still there: a second run adds them again (or rewrites)
        torch.cuda.synchronize()
        if not accumulate:
            for a, b in zip(want, got):
                assert torch.equal(a, b)


def test_batched_reduce_four_channel_form_gives_the_bits_of_the_per_layer_form(T, cuda_dev):
    """Job kind 3 of ryolo_conv_wgrad_reduce_batch (four input channels per thread, 16-B buffer loads, all loads of a split quarter in
    flight; what the planner picks for the many-split layers) against kind 1 (the per-layer body) on the SAME synthetic partial tiles, for
    every split count from 8 to 140 -- quarters of 2 .. 35 splits: the one-pass forms of 4 and 8, every remainder of the 16-pass form --
    3x3 and 1x1, accumulating and overwriting, one block per 256 elements and a grid-stride walk; and kind 4 (every load of a workgroup in
    flight) against kind 2 (the transposing 3x3 body) for S = 1 .. 7.  Each in its one-group and its wide form.  Bit for bit; 819 jobs in one launch (the job search's > 256 branch)."""
    import ctypes as C
    tr, dev = T.tr, cuda_dev
    gen = torch.Generator().manual_seed(17)
    L = tr._lib.lib()
    jobs, keep, pairs = [], [], []
    for ks, cout, cin, cin_k in ((3, 8, 12, 16), (1, 40, 72, 72)):
        taps = ks * ks
        kpad, cout_pad = (taps * cin_k + 63) // 64 * 64, 128
        total = cout * cin * taps
        for S in range(8, 141):
            # values of very different magnitude: a changed summation order shows in the last bits
            part = (torch.randn(S, cout_pad, kpad, generator=gen) * torch.exp(3 * torch.randn(S, 1, 1, generator=gen))).to(dev)
            g0 = torch.randn(cout, cin, ks, ks, generator=gen).to(dev)
            ga, gb, gc = g0.clone(), g0.clone(), g0.clone()
            acc = S % 2
            for kind, wide, gt in ((1, 0, ga), (3, 0, gb), (3, 1, gc)):     # (wide: 4 / 2 groups of channels per thread for quarters of <= 4 / 8)
                j = tr.WgradReduceJob()
                j.part, j.g, j.S, j.Cout, j.Cin_real, j.Cin_k, j.ks, j.Kpad, j.Cout_pad = part.data_ptr(), gt.data_ptr(), S, cout, cin, cin_k, ks, kpad, cout_pad
                j.accumulate, j.kind, j.wide = acc, kind, wide
                quarter = (S + 3) // 4
                per_block = 64 if kind == 1 else 256 * ((4 if quarter <= 4 else (2 if quarter <= 8 else 1)) if wide else 1)
                nb = (total + per_block - 1) // per_block
                j.block_begin, j.block_end = 0, nb if S % 3 else max(1, nb // 3)         # every third: fewer blocks, grid-stride walk
                jobs.append(j)
            keep.append(part)
            pairs.append((S, ks, ga, (gb, gc), g0))
    # the transposing 3x3 form (kind 2) and its every-load-in-flight form (kind 4): S = 1 .. 7, one workgroup per (c_out, 64 input channels)
    cout, cin, kpad, cout_pad = 5, 128, 1152, 128
    for S in range(1, 8):
        part = (torch.randn(S, cout_pad, kpad, generator=gen) * torch.exp(3 * torch.randn(S, 1, 1, generator=gen))).to(dev)
        g0 = torch.randn(cout, cin, 3, 3, generator=gen).to(dev)
        ga, gb, gc = g0.clone(), g0.clone(), g0.clone()
        for kind, wide, gt in ((2, 0, ga), (4, 0, gb), (4, 1, gc)):         # (wide: 4 / 2 units per workgroup for S <= 3 / 7; 10 units: a ragged last one)
            j = tr.WgradReduceJob()
            j.part, j.g, j.S, j.Cout, j.Cin_real, j.Cin_k, j.ks, j.Kpad, j.Cout_pad = part.data_ptr(), gt.data_ptr(), S, cout, cin, cin, 3, kpad, cout_pad
            units = cout * (cin // 64)
            per_wg = (4 if S <= 3 else 2) if wide else 1
            j.accumulate, j.kind, j.wide, j.block_begin, j.block_end = S % 2, kind, wide, 0, (units + per_wg - 1) // per_wg
            jobs.append(j)
        keep.append(part)
        pairs.append((S, 3, ga, (gb, gc), g0))
    arr = (tr.WgradReduceJob * len(jobs))()
    blk = 0
    for q, j in enumerate(jobs):
        nb = j.block_end
        j.block_begin, j.block_end = blk, blk + nb
        blk += nb
        arr[q] = j
    table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    assert len(jobs) > 256
    tr._lib.check(L.ryolo_conv_wgrad_reduce_batch(table.data_ptr(), len(jobs), blk, tr._s(dev)),
                  "ryolo_conv_wgrad_reduce_batch")
    torch.cuda.synchronize()
    for S, ks, ga, others, g0 in pairs:
        assert not torch.equal(ga, g0)
        for w, gb in enumerate(others):
            assert torch.equal(ga, gb), (S, ks, w, (ga - gb).abs().max().item())


def test_batched_weight_pack_equals_the_single_layout_packs(T, cuda_dev):
    """ryolo_conv_pack_batch (every packed weight image of a step in one launch, LDS-tiled transposes) byte for byte against
    the element-wise single-layout kernels (ryolo_conv_pack_weights / _dgrad): 3x3 stride 1 and 2, 1x1, the padded first
    layer (C_in 3 -> 8), channel counts that are not multiples of the tile (504, 56, 72)."""
    g = torch.Generator().manual_seed(12)
    shapes = [(32, 3, 3, 1, 8), (64, 32, 3, 2, 32), (128, 64, 3, 1, 64), (64, 128, 1, 1, 128), (504, 1024, 1, 1, 1024),
              (256, 128, 3, 2, 128), (56, 72, 3, 1, 72), (72, 56, 1, 1, 56)]
    batch = T.tr.WeightPackBatch(cuda_dev)
    keep = []
    L = T.tr._lib.lib()
    for cout, cin, k, s, cin_pad in shapes:
        w = torch.randn(cout, cin, k, k, generator=g).to(cuda_dev)
        pf = torch.full((L.ryolo_conv_packed_weight_bytes(cout, cin_pad, k),), 0x5a, dtype=torch.uint8, device=cuda_dev)
        pd = None
        if cin != 3:
            pd = torch.full((L.ryolo_conv_packed_dgrad_bytes(cout, cin, k, s),), 0x5a, dtype=torch.uint8, device=cuda_dev)
        batch.add(w, s, cin_pad, pf, pd)
        keep.append((w, s, cin_pad, pf, pd))
    batch.finalize()
    batch.run()
    torch.cuda.synchronize()
    for w, s, cin_pad, pf, pd in keep:
        assert torch.equal(pf, T.ops.pack_weights(w, cin_pad=cin_pad)), tuple(w.shape)
        if pd is not None:
            assert torch.equal(pd, T.tr.pack_weights_dgrad(w, s)), (tuple(w.shape), s)


def test_upsample_bwd_and_pgrad_layout(T, cuda_dev):
    g = torch.Generator().manual_seed(4)
    dy = r16(torch.randn(2, 16, 8, 12, generator=g))
    dx = torch.zeros(2, 4, 6, 16, dtype=torch.bfloat16, device=cuda_dev)
    T.tr.upsample2x_bwd(nhwc(dy, cuda_dev), dx, False)
    want = r16(F.avg_pool2d(dy, 2) * 4)
    assert torch.allclose(nchw(dx), want, rtol=2 ** -7, atol=1e-2)
    pg = torch.randn(2, 3, 4, 5, 7, generator=g)
    out = torch.zeros(2, 4, 5, 24, dtype=torch.bfloat16, device=cuda_dev)
    T.tr.pgrad_to_nhwc(pg.to(cuda_dev), out)
    ref = pg.permute(0, 2, 3, 1, 4).reshape(2, 4, 5, 21)
    assert torch.equal(out[..., :21].float().cpu(), r16(ref)) and bool((out[..., 21:] == 0).all())
    # tiled variant (na*no a multiple of 8): 72 anchors x 7, a pixel count that is not a multiple of the 32-pixel tile and
    # a tile that straddles two images
    pg = torch.randn(3, 72, 5, 7, 7, generator=g)
    out = torch.full((3, 5, 7, 512), 3.0, dtype=torch.bfloat16, device=cuda_dev)
    T.tr.pgrad_to_nhwc(pg.to(cuda_dev), out)
    ref = pg.permute(0, 2, 3, 1, 4).reshape(3, 5, 7, 504)
    assert torch.equal(out[..., :504].float().cpu(), r16(ref)) and bool((out[..., 504:] == 3.0).all())


@pytest.mark.parametrize("nc", [1, 3])
def test_yolo_loss_kernel_vs_autograd(cuda_dev, nc):
    """csrc/loss.hip (objectness over all cells + the positives' regression / class terms, with gradient) against
    autograd through loss_static.compute_loss_static on the same head: several targets per image, two of them in the
    same cell (shared objectness target, accumulated gradients), padded rows, a clamped exp(wh)."""
    import math
    from rotate_yolov3_amd.model import hip_train_ops as tr
    from rotate_yolov3_amd.model.loss_static import build_targets_static, compute_loss_static, pad_targets

    class Obj(object):
        pass
    bs, na, ny, nx = 3, 6, 5, 7
    no = 6 + nc
    g = torch.Generator().manual_seed(5 + nc)
    layer, core = Obj(), Obj()
    layer.ng = torch.tensor([float(nx), float(ny)], device=cuda_dev)
    wh = torch.tensor([[1.0, 0.5], [2.0, 0.6], [3.0, 1.0]]).repeat_interleave(2, 0)
    ang = torch.tensor([-0.6, 0.6]).repeat(3)
    layer.anchor_vec = torch.cat((wh, ang[:, None]), 1).to(cuda_dev)
    hyp = {"giou": 0.7, "cls": 1.3, "cls_pw": 1.5, "obj": 2.1, "obj_pw": 0.8, "iou_t": 0.3, "ang_t": math.pi / 4,
           "reg": 1.1, "context_factor": 1.0}
    core.yolo_layers, core.module_list, core.nc, core.arc, core.hyp = [0], [layer], nc, "default", hyp
    rows = []
    for i in range(bs):
        for _ in range(3):
            cx, cy = (0.1 + 0.8 * torch.rand(2, generator=g)).tolist()
            w = float(0.15 + 0.3 * torch.rand(1, generator=g))
            rows.append([i, int(torch.randint(0, nc, (1,), generator=g)), cx, cy, w, w / 3,
                         float((torch.rand(1, generator=g) - 0.5) * 2.5)])
    rows.append(list(rows[0]))            # duplicate target: same cell, same anchors
    rows[-1][4] *= 1.05
    targets = torch.tensor(rows, dtype=torch.float32, device=cuda_dev)
    tpad, valid = pad_targets(targets, 16)
    p = (torch.randn(bs, na, ny, nx, no, generator=g) * 1.5).to(cuda_dev)
    p[0, :, 0, 0, 2] = 8.0                # exp(8) > 1e3: the clamp cuts the gradient
    leaf = p.clone().requires_grad_(True)
    loss, items = compute_loss_static([leaf], tpad, valid, core, hyp)
    loss.backward()
    heads = build_targets_static(core, tpad, valid, hyp)
    assert float(heads[0]['w'].sum()) >= len(rows)          # every target got an anchor (fallback included)
    dp = torch.full_like(p, 7.0)
    it = torch.zeros(4, device=cuda_dev)
    tr.yolo_loss_head(p, heads[0], nc, hyp, tr.yolo_loss_bitmap(p), dp, it)
    torch.cuda.synchronize()
    assert torch.allclose(it[:3], items[:3], rtol=2e-5, atol=1e-6), (it, items)
    err = (dp - leaf.grad).abs().max().item()
    assert torch.allclose(dp, leaf.grad, rtol=1e-4, atol=1e-8), err


@pytest.mark.parametrize("nc", [1, 3])
def test_yolo_loss_nhwc_kernel_equals_the_fp32_layout_path(cuda_dev, nc):
    """ryolo_yolo_loss_nhwc (what the training engine runs: bf16 NHWC head in, bf16 NHWC head gradient out, fp32 scratch left
    all-zero) against ryolo_yolo_loss + the layout pass on the same head: same loss items, the same gradient at bf16
    resolution, scratch zero afterwards -- twice in a row (the second call relies on the re-zeroed scratch)."""
    import math
    from rotate_yolov3_amd.model import hip_train_ops as tr
    from rotate_yolov3_amd.model.loss_static import build_targets_static, pad_targets

    class Obj(object):
        pass
    bs, na, ny, nx = 3, 8, 5, 7
    no = 6 + nc
    g = torch.Generator().manual_seed(9 + nc)
    layer, core = Obj(), Obj()
    layer.ng = torch.tensor([float(nx), float(ny)], device=cuda_dev)
    wh = torch.tensor([[1.0, 0.5], [2.0, 0.6], [3.0, 1.0], [1.5, 1.5]]).repeat_interleave(2, 0)
    ang = torch.tensor([-0.6, 0.6]).repeat(4)
    layer.anchor_vec = torch.cat((wh, ang[:, None]), 1).to(cuda_dev)
    hyp = {"giou": 0.7, "cls": 1.3, "cls_pw": 1.5, "obj": 2.1, "obj_pw": 0.8, "iou_t": 0.3, "ang_t": math.pi / 4,
           "reg": 1.1, "context_factor": 1.0}
    core.yolo_layers, core.module_list, core.nc, core.arc, core.hyp = [0], [layer], nc, "default", hyp
    rows = []
    for i in range(bs):
        for _ in range(3):
            cx, cy = (0.1 + 0.8 * torch.rand(2, generator=g)).tolist()
            w = float(0.15 + 0.3 * torch.rand(1, generator=g))
            rows.append([i, int(torch.randint(0, nc, (1,), generator=g)), cx, cy, w, w / 3,
                         float((torch.rand(1, generator=g) - 0.5) * 2.5)])
    rows.append(list(rows[0]))
    targets = torch.tensor(rows, dtype=torch.float32, device=cuda_dev)
    tpad, valid = pad_targets(targets, 16)
    heads = build_targets_static(core, tpad, valid, hyp)
    C = na * no
    head = (torch.randn(bs, ny, nx, C, generator=g) * 1.5).to(torch.bfloat16).to(cuda_dev)
    p = head.float().reshape(bs, ny, nx, na, no).permute(0, 3, 1, 2, 4).contiguous()      # what ryolo_yolo_decode hands out
    dp = torch.full_like(p, 7.0)
    it0 = torch.zeros(4, device=cuda_dev)
    tr.yolo_loss_head(p, heads[0], nc, hyp, tr.yolo_loss_bitmap(p), dp, it0)
    want = torch.empty(bs, ny, nx, C, dtype=torch.bfloat16, device=cuda_dev)
    tr.pgrad_to_nhwc(dp, want)
    scratch = torch.zeros_like(p)
    for rep in range(2):
        hg = torch.full((bs, ny, nx, C), 3.0, dtype=torch.bfloat16, device=cuda_dev)
        it1 = torch.zeros(4, device=cuda_dev)
        tr.yolo_loss_head_nhwc(head, p, heads[0], nc, hyp, tr.yolo_loss_bitmap(p), scratch, hg, it1)
        torch.cuda.synchronize()
        assert torch.allclose(it1[:3], it0[:3], rtol=2e-5, atol=1e-6), (it1, it0)
        assert torch.allclose(hg.float(), want.float(), rtol=2 ** -7, atol=1e-9), (hg.float() - want.float()).abs().max()
        assert float(scratch.abs().max()) == 0.0
        assert float((want.float() != 0).float().mean()) > 1.0 / no - 1e-6     # the objectness column is dense
    # the upstream-gradient scale: a no-op for 1, a multiply otherwise
    keep = hg.clone()
    tr.scale_bf16_if(torch.ones(1, device=cuda_dev), hg)
    assert torch.equal(hg, keep)
    tr.scale_bf16_if(torch.full((1,), 0.5, device=cuda_dev), hg)
    assert torch.equal(hg.float(), keep.float() * 0.5)


@pytest.mark.parametrize("seed,cf", [(0, 1.0), (1, 1.0), (2, 1.25)])
def test_build_targets_kernel_equals_tensor_formulation(cuda_dev, seed, cf):
    """ryolo_build_targets (one thread per target over the head-major candidate table) against
    loss_static.build_targets_static: identical candidate weights, indices and target boxes -- incl. targets that need
    the best-anchor fallback (seed 1: thin boxes no anchor accepts) and a context factor != 1."""
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.model import hip_train_ops as tr
    from rotate_yolov3_amd.model.loss_static import build_targets_static, pad_targets
    from rotate_yolov3_amd.model.models import Darknet, create_grids
    from rotate_yolov3_amd.utils.synthetic import synthetic_targets
    hyp = {"giou": 0.1, "cls": 27.76, "cls_pw": 1.0, "obj": 20.35, "obj_pw": 1.0, "iou_t": 0.5, "ang_t": 3.1415926 / 12,
           "reg": 1.0, "context_factor": cf}
    model = Darknet(make_cfg.darknet53(width=96, height=96), hyp).to(cuda_dev)
    model.nc, model.arc, model.hyp = 1, "default", hyp
    for k, i in enumerate(model.yolo_layers):
        n = 96 // (32 >> k)
        create_grids(model.module_list[i], (96, 96), (n, n), cuda_dev)
    targets = synthetic_targets(6, seed=seed + 30, device=cuda_dev)
    if seed == 1:
        targets[:, 4] = 0.9
        targets[:, 5] = 0.02
    tpad, valid = pad_targets(targets, 40)
    want = build_targets_static(model, tpad, valid, hyp)
    bt = tr.BuildTargets(model, 40, cuda_dev)
    bt.run(tpad, valid.to(torch.uint8), hyp, cf)
    got = bt.heads()
    torch.cuda.synchronize()
    nt = len(targets)
    assert sum(float(w['w'].sum()) for w in want) >= nt
    for a, b in zip(want, got):
        assert torch.equal(a['w'], b['w'])
        assert float(b['npos']) == float(a['w'].sum())
        for k in ('b', 'cls', 'gj', 'gi'):
            assert torch.equal(a[k][:nt], b[k][:nt]), k
        for k in ('gxy', 'gwh', 'ga'):
            assert torch.equal(a[k][:nt], b[k][:nt]), (k, (a[k][:nt] - b[k][:nt]).abs().max().item(), a[k][:3], b[k][:3])


def test_fused_sgd_equals_torch_sgd(cuda_dev):
    """utils/fused_sgd.FusedSGD (one launch) against torch.optim.SGD over 4 steps: momentum + nesterov, weight decay on one
    param group only, a learning-rate change between steps, a parameter without gradient, state_dict layout."""
    from rotate_yolov3_amd.utils.fused_sgd import FusedSGD
    g = torch.Generator().manual_seed(9)
    shapes = [(64, 32, 3, 3), (64,), (1,), (1000, 17), (5,)]
    pa = [torch.randn(s, generator=g).to(cuda_dev).requires_grad_(True) for s in shapes]
    pb = [p.detach().clone().requires_grad_(True) for p in pa]

    def make(cls, ps):
        o = cls(ps[1:], lr=0.01, momentum=0.97, nesterov=True)
        o.add_param_group({'params': ps[:1], 'weight_decay': 0.05})
        return o
    oa, ob = make(torch.optim.SGD, pa), make(FusedSGD, pb)
    for step in range(4):
        for k, (x, y) in enumerate(zip(pa, pb)):
            if k == 4:
                continue                                        # never gets a gradient
            gr = torch.randn(x.shape, generator=g).to(cuda_dev)
            x.grad, y.grad = gr.clone(), gr.clone()
        if step == 2:
            for o in (oa, ob):
                for grp in o.param_groups:
                    grp['lr'] *= 0.1
        oa.step()
        ob.step()
        for x, y in zip(pa, pb):
            assert torch.allclose(x, y, rtol=1e-5, atol=1e-6), (step, (x - y).abs().max())     # fma vs mul+add: ulps
    sa, sb = oa.state_dict(), ob.state_dict()
    assert sa['state'].keys() == sb['state'].keys()
    for k in sa['state']:
        assert torch.allclose(sa['state'][k]['momentum_buffer'], sb['state'][k]['momentum_buffer'], rtol=1e-5, atol=1e-6)


def test_hip_build_targets_and_loss_against_the_reference_golden(cuda_dev):
    """the fixture captured from the reference's own model/loss.py (tests/golden/loss_d53_96.npz: three
    HRSC labels + two hand-made targets incl. the best-anchor fallback) fed DIRECTLY through the HIP kernels
    (ryolo_build_targets + ryolo_yolo_loss), no torch formulation in between: positives (image, anchor, gj, gi), target boxes,
    anchor vectors, loss items and d loss / d p per head against the reference's numbers (loss.py:161-367)."""
    from rotate_yolov3_amd.model import hip_train_ops as tr
    from rotate_yolov3_amd.model.loss_static import pad_targets
    from tests.test_loss import load_case
    z, hyp, model = load_case(cuda_dev)
    model.nc, model.arc, model.hyp = 1, "default", hyp
    targets = torch.from_numpy(z["targets"]).to(cuda_dev)
    cap = 8
    tpad, valid = pad_targets(targets, cap)
    bt = tr.BuildTargets(model, cap, cuda_dev)
    bt.run(tpad, valid.to(torch.uint8), hyp, hyp["context_factor"])
    heads = bt.heads()
    torch.cuda.synchronize()
    items = torch.zeros(4, device=cuda_dev)
    for k in range(3):
        hd = heads[k]
        w = hd["w"].cpu()                                   # [na, cap] 0/1
        a_idx, t_idx = torch.nonzero(w, as_tuple=True)
        got = sorted(zip(hd["b"].cpu()[t_idx].tolist(), a_idx.tolist(), hd["gj"].cpu()[t_idx].tolist(), hd["gi"].cpu()[t_idx].tolist(),
                         t_idx.tolist()))
        ref_idx = z["idx%d" % k]                            # rows: b, a, gj, gi (loss.py:252 indices.append((b, a, gj, gi)))
        want = sorted(zip(*[ref_idx[r].tolist() for r in range(4)])) if ref_idx.shape[1] else []
        assert [g[:4] for g in got] == want, (k, got, want)
        # target boxes (gxy - gij, gwh, ga) and anchor vectors of the positives, matched through (b, a, gj, gi)
        ref_rows = {}
        for j in range(ref_idx.shape[1]):
            ref_rows.setdefault(tuple(int(ref_idx[r][j]) for r in range(4)), []).append((z["tbox%d" % k][j], z["av%d" % k][j]))
        av = model.module_list[model.yolo_layers[k]].anchor_vec.cpu().numpy()
        for (b, a, gj, gi, t) in got:
            tb = np.concatenate([hd["gxy"].cpu().numpy()[t], hd["gwh"].cpu().numpy()[t], hd["ga"].cpu().numpy()[t:t + 1]])
            cands = ref_rows[(b, a, gj, gi)]
            assert any(np.allclose(tb, c[0], rtol=1e-6, atol=1e-7) and np.allclose(av[a], c[1], rtol=1e-6) for c in cands), (k, tb, cands)
        p = torch.from_numpy(z["p%d" % k]).to(cuda_dev)
        dp = torch.full_like(p, 7.0)
        tr.yolo_loss_head(p, hd, model.nc, hyp, tr.yolo_loss_bitmap(p), dp, items)
        torch.cuda.synchronize()
        g = z["g%d" % k]
        assert np.allclose(dp.cpu().numpy(), g, rtol=1e-4, atol=1e-8), (k, np.abs(dp.cpu().numpy() - g).max())
    it = items.cpu().numpy()
    assert np.allclose(it[:3], z["loss_items"][:3], rtol=2e-5, atol=1e-6), (it, z["loss_items"])
    assert abs(it[:3].sum() - float(z["loss"][0])) <= 2e-5 * abs(float(z["loss"][0]))


@pytest.mark.parametrize("arc", ["Fdefault", "uBCE", "uCE", "FuBCE", "FuCE"])
@pytest.mark.parametrize("layout", ["fp32", "nhwc"])
def test_hip_loss_other_arcs_against_the_reference_golden(cuda_dev, arc, layout):
    """The reference's other arcs (model/loss.py:284-286 focal wrappers around every criterion but the IoU term, :350-360 unified
    BCE / CE heads; train.py:380 recommends --arc Fdefault) on the HIP kernels, against loss items and d loss / d p the
    reference's own compute_loss produced for the heads and targets of loss_d53_96.npz (tests/golden/loss_arcs_d53_96.npz).
    Both entry points: the fp32-layout one and the NHWC bf16 one the training engine uses (the head gradient then carries
    bf16 rounding: 2^-8 relative)."""
    import os
    from rotate_yolov3_amd.model import hip_train_ops as tr
    from rotate_yolov3_amd.model.loss_static import pad_targets
    from tests.test_loss import load_case
    z, hyp, model = load_case(cuda_dev)
    za = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss_arcs_d53_96.npz"))
    model.nc, model.arc, model.hyp = 1, arc, hyp
    flags = tr.arc_flags(arc)
    targets = torch.from_numpy(z["targets"]).to(cuda_dev)
    cap = 8
    tpad, valid = pad_targets(targets, cap)
    bt = tr.BuildTargets(model, cap, cuda_dev)
    bt.run(tpad, valid.to(torch.uint8), hyp, hyp["context_factor"])
    heads = bt.heads()
    items = torch.zeros(4, device=cuda_dev)
    items_ref = torch.zeros(4, device=cuda_dev)
    for k in range(3):
        p = torch.from_numpy(z["p%d" % k]).to(cuda_dev)
        g = za["g%d_%s" % (k, arc)]
        bm = tr.yolo_loss_bitmap(p, model.nc, flags)
        if layout == "fp32":
            dp = torch.full_like(p, 7.0)
            tr.yolo_loss_head(p, heads[k], model.nc, hyp, bm, dp, items, flags)
            torch.cuda.synchronize()
            got = dp.cpu().numpy()
            assert np.allclose(got, g, rtol=2e-4, atol=1e-8), (k, np.abs(got - g).max(), np.abs(g).max())
        else:
            # the engine's entry point reads the head as the conv wrote it (NHWC bf16, channel = a*no + k) and writes the head
            # gradient in the same layout: compared with the fp32-layout kernels (checked against the reference above) on the
            # bf16-rounded head
            bs, na, ny, nx, no = p.shape
            head = p.permute(0, 2, 3, 1, 4).reshape(bs, ny, nx, na * no).to(torch.bfloat16).contiguous()
            pq = head.float().reshape(bs, ny, nx, na, no).permute(0, 3, 1, 2, 4).contiguous()
            dp = torch.full_like(pq, 7.0)
            tr.yolo_loss_head(pq, heads[k], model.nc, hyp, tr.yolo_loss_bitmap(pq, model.nc, flags), dp, items_ref, flags)
            hg = torch.full_like(head, 3.0)
            dps = torch.zeros_like(pq)
            tr.yolo_loss_head_nhwc(head, pq, heads[k], model.nc, hyp, bm, dps, hg, items, flags)
            torch.cuda.synchronize()
            assert not bool(dps.any()), "the sparse scratch must be left all-zero"
            got = hg.float().reshape(bs, ny, nx, na, no).permute(0, 3, 1, 2, 4)
            want = dp.to(torch.bfloat16).float()
            assert torch.allclose(got, want, rtol=2 ** -7, atol=1e-9), (k, float((got - want).abs().max()))
    it = items.cpu().numpy()
    ref = za["items_" + arc] if layout == "fp32" else items_ref.cpu().numpy()
    assert np.allclose(it[:3], ref[:3], rtol=5e-5, atol=1e-6), (it, ref)


@pytest.mark.parametrize("n,cin,cout,hw,tile", [(2, 64, 128, 192, 0x800),      # 576 tiles of 128x128 on the persistent grid
                                                (2, 128, 256, 192, 0x800),     # two channel tiles per pixel tile
                                                (5, 64, 32, 176, 0x800),       # 256x32 tiles (605 of them)
                                                (2, 64, 128, 192, 0x200)])     # same layer on the one-tile-per-workgroup kernel
def test_conv_1x1_statistics_on_the_persistent_grid(T, cuda_dev, n, cin, cout, hw, tile):
    """Training forward of a 1x1 layer on conv_igemm_persist_kernel<..., STATS>: the statistics are accumulated in registers
    over all the tiles a workgroup walks and flushed once; output and sums must equal the fp32 convolution of the same
    bf16 operands / the sums of the stored z."""
    g, x, wt = _setup(n, cin, cout, hw, hw, 1, 5)
    xd = nhwc(x, cuda_dev)
    packed = T.ops.pack_weights(wt.to(cuda_dev), cin_pad=cin)
    ones = torch.ones(T.ops.cpad(cout), device=cuda_dev)
    zeros = torch.zeros(T.ops.cpad(cout), device=cuda_dev)
    d = T.tr.make_desc(xd, cout, 1, 1, 0, tile=tile)
    z = torch.empty(n, hw, hw, cout, dtype=torch.bfloat16, device=cuda_dev)
    part = T.tr.conv_fwd_stats(d, xd, packed, ones, zeros, z)
    torch.cuda.synchronize()
    zq = nchw(z)
    assert torch.allclose(zq, r16(F.conv2d(x, wt)), rtol=2 ** -7, atol=2e-3)
    s1 = part[:, 0, :cout].sum(0).cpu()
    s2 = part[:, 1, :cout].sum(0).cpu()
    zd = zq.double()
    assert torch.allclose(s1, zd.sum((0, 2, 3)), rtol=1e-5, atol=2e-2)
    assert torch.allclose(s2, (zd * zd).sum((0, 2, 3)), rtol=1e-5, atol=2e-2)
    assert float(part[:, :, cout:].abs().max()) == 0.0 if part.shape[2] > cout else True
    # the partial rows are reproducible run to run (fixed-order fp32 sums inside a workgroup, 64-bit atomics across)
    part2 = T.tr.conv_fwd_stats(d, xd, packed, ones, zeros, torch.empty_like(z))
    assert torch.equal(part.sum(0), part2.sum(0))


@pytest.mark.parametrize("act,n,h,w", [(1, 2, 40, 48), (2, 1, 37, 53), (0, 3, 16, 16)])
def test_layer0_recompute_path_equals_the_stored_z_path(T, cuda_dev, act, n, h, w):
    """Layer 0 trains without its conv output (ryolo_conv0_*: statistics-only pass, BatchNorm + activation on the recomputed z,
    backward reduce / apply on the recomputed z).  Same results as the generic path that stores z: y and dz bit for bit (same
    MFMAs, same bf16 rounding of z), statistics and parameter gradients to fp32 summation order."""
    tr = T.tr
    g = torch.Generator().manual_seed(31 + act)
    x = torch.zeros(n, 8, h, w)
    x[:, :3] = r16(torch.randn(n, 3, h, w, generator=g))
    wt = r16(torch.randn(32, 3, 3, 3, generator=g) / 5.0)
    xd = nhwc(x, cuda_dev)
    packed = T.ops.pack_weights(wt.to(cuda_dev), cin_pad=8)
    ones, zeros = torch.ones(128, device=cuda_dev), torch.zeros(128, device=cuda_dev)
    gamma = (torch.rand(32, generator=g) + 0.5).to(cuda_dev)
    beta = (torch.randn(32, generator=g) * 0.3).to(cuda_dev)
    slope = torch.tensor([0.1], device=cuda_dev) if act == 1 else None
    dy = nhwc(r16(torch.randn(n, 32, h, w, generator=g)), cuda_dev)
    d = tr.make_desc(xd, 32, 3, 1, 1)
    assert tr.conv0_recompute_supported(d)
    M = n * h * w
    # generic path
    z = torch.empty(n, h, w, 32, dtype=torch.bfloat16, device=cuda_dev)
    part = tr.conv_fwd_stats(tr.make_desc(xd, 32, 3, 1, 1), xd, packed, ones, zeros, z)
    st_a = tr.bn_finalize(part.clone(), 32, M, gamma, beta)
    y_a = torch.empty_like(z)
    tr.bn_act_fwd(z, st_a[2], st_a[3], act, slope, y_a)
    dz_a = torch.empty_like(z)
    dg_a, db_a, ds_a = torch.zeros(32, device=cuda_dev), torch.zeros(32, device=cuda_dev), torch.zeros(1, device=cuda_dev)
    ws = torch.empty(tr.bn_bwd_ws_bytes(M, 32), dtype=torch.uint8, device=cuda_dev)
    tr.bn_act_bwd(z, dy, st_a, act, slope, dz_a, dg_a, db_a, ds_a if act == 1 else None, ws)
    # recompute path
    part_b = tr.conv_fwd_stats(d, xd, packed, ones, zeros, None)
    assert torch.allclose(part_b.sum(0), part.sum(0), rtol=1e-12, atol=1e-9)
    st_b = tr.bn_finalize(part_b, 32, M, gamma, beta)
    for a, b in zip(st_a, st_b):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)
    y_b = torch.full_like(z, 9.0)
    tr.conv0_bn_act_fwd(d, xd, packed, st_a[2], st_a[3], act, slope, y_b)
    assert torch.equal(y_b, y_a)
    dz_b = torch.full_like(z, 9.0)
    dg_b, db_b, ds_b = torch.zeros(32, device=cuda_dev), torch.zeros(32, device=cuda_dev), torch.zeros(1, device=cuda_dev)
    ws0 = tr.conv0_bn_bwd_ws(cuda_dev)
    for rep in range(2):       # twice: the workspace must be left zeroed
        dg_b.zero_(); db_b.zero_(); ds_b.zero_()
        tr.conv0_bn_bwd(d, xd, packed, dy, st_a, act, slope, dz_b, dg_b, db_b, ds_b if act == 1 else None, ws0)
        torch.cuda.synchronize()
        assert torch.allclose(dg_b, dg_a, rtol=2e-4, atol=2e-3) and torch.allclose(db_b, db_a, rtol=2e-4, atol=2e-3)
        if act == 1:
            assert torch.allclose(ds_b, ds_a, rtol=2e-4, atol=2e-3)
        err = (dz_b.float() - dz_a.float()).abs()
        assert float(err.max()) <= 2 ** -7 * float(dz_a.float().abs().max()), float(err.max())     # (s1, s2 differ in the last bits)
    assert not bool(ws0[:-64 * 4].any())          # the partial rows are left zeroed (the tail holds the apply pass's constants)
    # ---- the one-pass form (csrc/conv0_bwd.hip): dgamma / dbeta / dslope and the weight gradient from G, Z, Sx, no dz.
    # Against the two-pass path + ryolo_conv2d_wgrad on its dz (which rounds dz to bf16 first: agreement to that rounding), and
    # against fp64 sums of the definition on the same bf16 z and dy.
    gw_a = torch.zeros(32, 3, 3, 3, device=cuda_dev)
    wsw = torch.empty(tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=cuda_dev)
    tr.conv_wgrad(d, xd, dz_b, 3, gw_a, True, wsw)
    wsf = torch.full((tr.conv0_bn_bwd_wgrad_ws(cuda_dev).numel(),), 0x7f, dtype=torch.uint8, device=cuda_dev)     # any contents
    outs = []
    for rep in range(2):
        gw_f = torch.full((32, 3, 3, 3), 0.25, device=cuda_dev)
        dg_f, db_f, ds_f = torch.zeros(32, device=cuda_dev), torch.zeros(32, device=cuda_dev), torch.zeros(1, device=cuda_dev)
        tr.conv0_bn_bwd_wgrad(d, xd, packed, dy, st_a, act, slope, dg_f, db_f, ds_f if act == 1 else None, gw_f, 3, True, wsf)
        torch.cuda.synchronize()
        outs.append((gw_f.clone(), dg_f.clone(), db_f.clone(), ds_f.clone()))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))                # bit-reproducible (fixed-order partial rows)
    gw_f = gw_f - 0.25
    assert torch.allclose(dg_f, dg_a, rtol=2e-4, atol=2e-3) and torch.allclose(db_f, db_a, rtol=2e-4, atol=2e-3)
    if act == 1:
        assert torch.allclose(ds_f, ds_a, rtol=2e-4, atol=2e-3)
    scale_w = float(gw_a.abs().max())
    assert float((gw_f - gw_a).abs().max()) <= 2 ** -7 * scale_w + 1e-3, (float((gw_f - gw_a).abs().max()), scale_w)
    z64, dy64 = z.double().view(-1, 32), dy.double().view(-1, 32)
    mean, invstd, scl, shf = [t.double() for t in st_a]
    u = z64 * scl + shf
    if act == 1:
        g64 = torch.where(u > 0, dy64, dy64 * 0.1)
    elif act == 2:
        sp = torch.nn.functional.softplus(u)
        tsp = torch.tanh(sp)
        g64 = dy64 * (tsp + u * (1 - tsp * tsp) * torch.sigmoid(u))
    else:
        g64 = dy64
    xhat = (z64 - mean) * invstd
    dz64 = scl * (g64 - g64.mean(0) - xhat * (g64 * xhat).mean(0))
    xp = F.unfold(x[:, :3].double().to(cuda_dev), 3, padding=1)                   # [n, 27 (ci, kh, kw), h*w]
    dw64 = torch.einsum("npc,nkp->ck", dz64.view(n, h * w, 32), xp).view(32, 3, 3, 3)
    assert float((gw_f.double() - dw64).abs().max()) <= 2e-3 * float(dw64.abs().max()) + 1e-3, (
        float((gw_f.double() - dw64).abs().max()), float(dw64.abs().max()))


@pytest.mark.parametrize("n,hw,cin,cout,acc,k", [(16, 76, 256, 128, True, 1), (16, 76, 256, 128, False, 1), (40, 38, 512, 256, True, 1),
                                                 (8, 152, 128, 64, True, 1),
                                                 # one row per pixel tile (the one-tile-per-workgroup kernels): 3x3 on the 128 x 128 tile, 3x3
                                                 # and 1x1 on the narrow tiles, and a launch with > 2048 rows (folded before the finalise)
                                                 (8, 76, 128, 256, True, 3), (4, 152, 64, 128, True, 3), (8, 152, 64, 128, False, 3),
                                                 (3, 75, 128, 256, True, 3),
                                                 (16, 76, 128, 256, True, 3), (2, 40, 128, 256, True, 3), (4, 76, 128, 256, False, 3)])
def test_dgrad_with_folded_bn_reduce_equals_the_two_pass_path(T, cuda_dev, n, hw, cin, cout, acc, k):
    """ryolo_conv2d_dgrad_bnreduce + ryolo_bn_act_bwd_reduced (the reduce pass of the producing block's BatchNorm / PReLU backward
    folded into the 1x1 data gradient that writes its dy) against ryolo_conv2d_dgrad followed by ryolo_bn_act_bwd: dx bit for
    bit (same kernel body), dz / dgamma / dbeta / dslope to fp32 summation order (rows per workgroup instead of pixel slabs)."""
    tr = T.tr
    g = torch.Generator().manual_seed(77 + hw + cin)
    dev = cuda_dev
    wt = r16(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5)
    xd = torch.randn(n, hw, hw, cin, generator=g).to(torch.bfloat16).to(dev)          # (only its shape matters to the descriptor)
    d = tr.make_desc(xd, cout, k, 1, (k - 1) // 2)
    rows = tr.dgrad_bnreduce_rows(d)
    assert rows > 0
    pk = tr.pack_weights_dgrad(wt.to(dev), 1)
    ones, zeros = torch.ones(T.ops.cpad(cin), device=dev), torch.zeros(T.ops.cpad(cin), device=dev)
    dz_x = torch.randn(n, hw, hw, cout, generator=g).to(torch.bfloat16).to(dev)       # gradient of the 1x1 conv's output
    prev = torch.randn(n, hw, hw, cin, generator=g).to(torch.bfloat16).to(dev)        # running gradient of the shortcut chain
    z = (torch.randn(n, hw, hw, cin, generator=g) * 1.5).to(torch.bfloat16).to(dev)   # conv output of the producing block
    gamma = (torch.rand(cin, generator=g) + 0.5).to(dev)
    beta = (torch.randn(cin, generator=g) * 0.3).to(dev)
    M = n * hw * hw
    zf = z.float().view(-1, cin)
    mean = zf.mean(0)
    invstd = (zf.var(0, unbiased=False) + 1e-5).rsqrt()
    scale = gamma * invstd
    shift = beta - mean * scale
    stats = (mean.contiguous(), invstd.contiguous(), scale.contiguous(), shift.contiguous())
    slope = torch.tensor([0.1], device=dev)
    # two-pass path
    dx_a = prev.clone() if acc else torch.full_like(prev, 7.0)
    tr.conv_dgrad(d, dz_x, pk, ones, zeros, dx_a, acc)
    dz_a = torch.empty_like(z)
    dg_a, db_a, ds_a = torch.zeros(cin, device=dev), torch.zeros(cin, device=dev), torch.zeros(1, device=dev)
    ws = torch.empty(tr.bn_bwd_ws_bytes(M, cin), dtype=torch.uint8, device=dev)
    tr.bn_act_bwd(z, dx_a, stats, 1, slope, dz_a, dg_a, db_a, ds_a, ws)
    # folded path (twice: `part` may hold anything on entry)
    part = torch.full((rows, 3, cin), 123.0, device=dev)
    for rep in range(2):
        dx_b = prev.clone() if acc else torch.full_like(prev, 7.0)
        tr.conv_dgrad_bnreduce(d, dz_x, pk, ones, zeros, dx_b, acc, z, stats, slope, part)
        dz_b = torch.full_like(z, 9.0)
        dg_b, db_b, ds_b = torch.zeros(cin, device=dev), torch.zeros(cin, device=dev), torch.zeros(1, device=dev)
        tr.bn_act_bwd_reduced(z, dx_b, stats, 1, slope, dz_b, dg_b, db_b, ds_b, part, ws)
        torch.cuda.synchronize()
        assert torch.equal(dx_b, dx_a)
        assert torch.allclose(dg_b, dg_a, rtol=2e-4, atol=2e-2) and torch.allclose(db_b, db_a, rtol=2e-4, atol=2e-2), (
            float((dg_b - dg_a).abs().max()), float((db_b - db_a).abs().max()))
        assert torch.allclose(ds_b, ds_a, rtol=2e-4, atol=2e-2), (float(ds_b), float(ds_a))
        err = (dz_b.float() - dz_a.float()).abs()
        assert float(err.max()) <= 2 ** -7 * float(dz_a.float().abs().max()), float(err.max())
    # and against fp64 sums of the definition on the stored dx
    d64, z64 = dx_a.double().view(-1, cin), z.double().view(-1, cin)
    u = z64 * scale.double() + shift.double()
    gg = torch.where(u > 0, d64, d64 * 0.1)
    assert torch.allclose(db_b.double(), gg.sum(0), rtol=1e-4, atol=1e-2)
    assert torch.allclose(dg_b.double(), (gg * (z64 - mean.double()) * invstd.double()).sum(0), rtol=1e-4, atol=1e-2)


@pytest.mark.parametrize("n,cin,cout,h,w", [(2, 256, 512, 38, 38), (3, 256, 1024, 21, 18)])
def test_stride2_dgrad_conv_mp_equals_conv_mq_bitwise(T, cuda_dev, n, cin, cout, h, w):
    """the stride-2 parity-class data gradients with >= 512 channels in K took the channel-major K order in conv_mp.hip's launcher
    but not in conv_mq.hip's, so the fp32 sums depended on which kernel the dispatch picked.  Both launchers now apply one rule: the two
    kernels (tile 8 / tile 9) give the same bits, with the rule's own choice and with either order forced."""
    from rotate_yolov3_amd import _lib
    tr, dev = T.tr, cuda_dev
    g = torch.Generator().manual_seed(77 + cout)
    wt = r16(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5).to(dev)
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    dz = torch.randn(n, ho, wo, cout, generator=g).to(torch.bfloat16).to(dev)
    prev = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).to(dev)
    pk = tr.pack_weights_dgrad(wt, 2)
    ones, zeros = torch.ones(T.ops.cpad(cin), device=dev), torch.zeros(T.ops.cpad(cin), device=dev)
    try:
        for order in (None, "0", "1"):
            _lib.set_tuning("RYOLO_MQ_KORDER", order)
            out = []
            for tile in (8, 9):
                d = tr.make_desc(prev, cout, 3, 2, 1, tile=tile)
                dx = prev.clone()
                tr.conv_dgrad(d, dz, pk, ones, zeros, dx, True)
                torch.cuda.synchronize()
                out.append(dx)
            assert torch.equal(out[0], out[1]), order
    finally:
        _lib.set_tuning("RYOLO_MQ_KORDER", None)


def test_dgrad_bnreduce_is_refused_where_it_does_not_apply(T, cuda_dev):
    tr = T.tr
    for (n, hw, cin, cout, k, s) in [(16, 38, 256, 512, 3, 1), (2, 304, 32, 64, 3, 1), (2, 304, 64, 32, 1, 1), (16, 76, 192, 128, 1, 1), (2, 19, 1024, 504, 1, 1), (8, 76, 256, 128, 1, 2)]:
        xd = torch.empty(n, hw, hw, cin, dtype=torch.bfloat16, device=cuda_dev)
        assert tr.dgrad_bnreduce_rows(tr.make_desc(xd, cout, k, s, (k - 1) // 2)) == 0, (n, hw, cin, cout, k, s)
