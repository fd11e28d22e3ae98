"""GPU tier: the weight-stationary 1x1 kernel (csrc/conv_pw.hip, tile 13 / auto) -- against the fp32 oracle like every conv tile
(2 bf16 ulp), and BIT FOR BIT against the 128 x 128 implicit-GEMM tile it replaces (same K order, same epilogue arithmetic):
inference, upsample, channel slices, the training forward with statistics, the data gradient with accumulation."""
import pytest
import torch
import torch.nn.functional as F

from tests.test_conv_gpu import _case, ops  # noqa: F401  (the oracle-checked case runner and its fixture)
from tests.test_train_ops_gpu import T, nhwc, nchw, r16, _setup  # noqa: F401

pytestmark = pytest.mark.gpu

# (C_in, C_out) of every configuration conv_pw serves in the forward: K 256/384 -> 128 (4 waves), K 128/256/512 -> 256 per
# workgroup (8 waves x 32), K 768/1024 -> 128 per workgroup (8 waves x 16); 504 = the YOLO heads (a partial last channel block)
def pw_cap(k, n):
    """tile bits 16+ for tile 13: workgroups per XCD = the number of channel blocks, so that ONE workgroup per XCD and channel block
    walks all the row blocks of its XCD -- a small tensor then reaches the steady state of the ring (the counted waits that
    account for earlier epilogues' stores)"""
    ncb = 128 if (k in (256, 384) and n <= 128) or k >= 768 else 256
    return 13 | (-(-n // ncb) << 16)


FWD = [(256, 128), (384, 128), (128, 256), (256, 512), (512, 256), (768, 256), (1024, 512), (1024, 504), (512, 504), (256, 504)]


@pytest.mark.parametrize("cin,cout", FWD)
@pytest.mark.parametrize("cap", [0, 1])
def test_conv_pw_vs_oracle_and_bitwise_vs_igemm(ops, cuda_dev, cin, cout, cap):
    """cap = 1: eight workgroups (one per XCD) walk the whole tensor -- several row blocks each, i.e. the steady state of the ring
    and its counted waits; cap = 0: the full grid (most workgroups own at most one row block)."""
    tile = pw_cap(cin, cout) if cap else 13
    act = 1 if cout != 504 else 0
    a = _case(ops, cuda_dev, 3, 41, 47, cin, cout, 1, 1, act, tile=tile, seed=cin + cout, ret_out=True)     # M = 5781: a partial last row block
    b = _case(ops, cuda_dev, 3, 41, 47, cin, cout, 1, 1, act, tile=1, seed=cin + cout, ret_out=True)
    assert torch.equal(a, b)


def test_conv_pw_mish_slices_and_upsample(ops, cuda_dev):
    _case(ops, cuda_dev, 2, 41, 37, 256, 128, 1, 1, 2, tile=pw_cap(256, 128), seed=3)                                    # mish
    _case(ops, cuda_dev, 2, 10, 10, 512, 256, 1, 1, 1, upsample=2, out_slice=(768, 0), tile=13, seed=26)               # layers 84-86
    _case(ops, cuda_dev, 2, 33, 31, 256, 128, 1, 1, 1, upsample=2, out_slice=(384, 0), tile=pw_cap(256, 128), seed=27)   # layers 96-98
    _case(ops, cuda_dev, 2, 40, 40, 768, 256, 1, 1, 1, in_slice=(1024, 128), tile=pw_cap(768, 256), seed=28)             # reads a concat slice
    _case(ops, cuda_dev, 2, 36, 36, 512, 256, 1, 1, 1, residual=True, tile=pw_cap(512, 256), seed=29)                    # a 1x1 with a shortcut


def test_conv_pw_is_the_auto_choice_where_it_wins(ops, cuda_dev):
    """the measured rule of conv_pw_preferred(): at least two row blocks per workgroup, K <= 768, C_out <= 1.5 C_in"""
    for (cin, cout, hw) in [(256, 128, 76), (512, 256, 38), (768, 256, 38), (384, 128, 76), (512, 504, 38)]:
        assert ops.conv_kernel_name(32, hw, hw, cin, cout, 1).startswith("conv_pw"), (cin, cout, hw)
    for (cin, cout, hw) in [(1024, 512, 19), (1024, 504, 19), (512, 256, 19), (256, 128, 38), (256, 504, 76), (128, 64, 152)]:
        assert not ops.conv_kernel_name(32, hw, hw, cin, cout, 1).startswith("conv_pw"), (cin, cout, hw)
    assert not ops.conv_kernel_name(32, 76, 76, 128, 256, 3).startswith("conv_pw")


@pytest.mark.parametrize("cin,cout,n,hw", [(256, 128, 32, 76), (512, 256, 32, 38), (1024, 512, 32, 19), (768, 256, 8, 38)])
def test_conv_pw_full_size_bitwise_vs_igemm(ops, cuda_dev, cin, cout, n, hw):
    """the bs-32 / 608^2 geometry (every workgroup walks 2-6 row blocks on the full grid): outputs equal the implicit-GEMM tile's"""
    g = torch.Generator(device=cuda_dev).manual_seed(cin)
    x = torch.randn(n, hw, hw, cin, device=cuda_dev, generator=g).to(torch.bfloat16)
    w = torch.randn(cout, cin, 1, 1, device=cuda_dev, generator=g) / cin ** 0.5
    pk = ops.pack_weights(w, cin_pad=cin)
    sc = torch.rand(ops.cpad(cout), device=cuda_dev, generator=g) + 0.5
    sh = torch.randn(ops.cpad(cout), device=cuda_dev, generator=g) * 0.3
    a = ops.conv2d_bn_act(x, pk, sc, sh, cout, 1, act=1, tile=13)
    b = ops.conv2d_bn_act(x, pk, sc, sh, cout, 1, act=1, tile=1)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    a2 = ops.conv2d_bn_act(x, pk, sc, sh, cout, 1, act=1, tile=13)
    assert torch.equal(a, a2)


@pytest.mark.parametrize("cin,cout,n,hw,cap", [(256, 128, 2, 60, 1), (512, 256, 3, 41, 1), (1024, 512, 2, 39, 1), (256, 128, 16, 76, 0),
                                               (128, 256, 2, 63, 1), (384, 128, 2, 47, 1)])
def test_conv_pw_training_forward_statistics(T, cuda_dev, cin, cout, n, hw, cap):
    """z bit for bit the implicit-GEMM tile's; the per-channel sums of z and z^2 against fp64 sums of the stored z; reproducible"""
    g, x, wt = _setup(n, cin, cout, hw, hw, 1, 5)
    xd = nhwc(x, cuda_dev)
    packed = T.ops.pack_weights(wt.to(cuda_dev), cin_pad=cin)
    ones = torch.ones(T.ops.cpad(cout), device=cuda_dev)
    zeros = torch.zeros(T.ops.cpad(cout), device=cuda_dev)
    d = T.tr.make_desc(xd, cout, 1, 1, 0, tile=pw_cap(cin, cout) if cap else 13)
    z = torch.empty(n, hw, hw, cout, dtype=torch.bfloat16, device=cuda_dev)
    part = T.tr.conv_fwd_stats(d, xd, packed, ones, zeros, z)
    d1 = T.tr.make_desc(xd, cout, 1, 1, 0, tile=1)
    z1 = torch.empty_like(z)
    T.tr.conv_fwd_stats(d1, xd, packed, ones, zeros, z1)
    torch.cuda.synchronize()
    assert torch.equal(z, z1)
    zd = nchw(z).double()
    s1 = part[:, 0, :cout].sum(0).cpu()
    s2 = part[:, 1, :cout].sum(0).cpu()
    assert torch.allclose(s1, zd.sum((0, 2, 3)), rtol=1e-5, atol=2e-2)
    assert torch.allclose(s2, (zd * zd).sum((0, 2, 3)), rtol=1e-5, atol=2e-2)
    part2 = T.tr.conv_fwd_stats(d, xd, packed, ones, zeros, torch.empty_like(z))
    assert torch.equal(part.sum(0), part2.sum(0))


@pytest.mark.parametrize("cin,cout,n,hw,acc,cap", [(256, 128, 2, 63, True, 1), (256, 128, 2, 63, False, 1), (512, 256, 3, 41, True, 1),
                                                   (1024, 512, 2, 39, True, 1), (256, 128, 16, 76, True, 0)])
def test_conv_pw_data_gradient(T, cuda_dev, cin, cout, n, hw, acc, cap):
    """the stride-1 data gradient of a 1x1 layer (K = the forward's C_out, channels = its C_in) on conv_pw: against autograd, and
    bit for bit against the implicit-GEMM tile"""
    g, x, wt = _setup(n, cin, cout, hw, hw, 1, 2)
    dz = r16(torch.randn(n, cout, hw, hw, generator=g))
    prev = r16(torch.randn(n, cin, hw, hw, generator=g))
    xd = nhwc(x, cuda_dev)
    pk = T.tr.pack_weights_dgrad(wt.to(cuda_dev), 1)
    ones = torch.ones(T.ops.cpad(cin), device=cuda_dev)
    zeros = torch.zeros(T.ops.cpad(cin), device=cuda_dev)
    outs = []
    for tile in (pw_cap(cout, cin) if cap else 13, 1):
        d = T.tr.make_desc(xd, cout, 1, 1, 0, tile=tile)
        dx = nhwc(prev, cuda_dev) if acc else torch.full((n, hw, hw, cin), 7.0, dtype=torch.bfloat16, device=cuda_dev)
        T.tr.conv_dgrad(d, nhwc(dz, cuda_dev), pk, ones, zeros, dx, acc)
        outs.append(dx)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    if n * hw * hw <= 8192:
        gx = F.conv_transpose2d(dz, wt)
        want = r16(r16(gx) + prev) if acc else r16(gx)
        mag = gx.abs() + (prev.abs() if acc else 0)
        err = (nchw(outs[0]) - want).abs()
        # accumulation rounds twice (the bf16 product sum, then the sum with the running gradient): one ulp each
        assert bool((err <= (2.0 if acc else 1.0) * 2 ** -7 * mag + 3e-3).all()), float(err.max())
