"""GPU tier: the MFMA implicit-GEMM conv block (through the C ABI) vs oracle/conv_oracle.py.
Tolerance (include/ryolo.h): 2 bf16 ulp relative + a small absolute floor from fp32 summation order."""
import numpy as np
import pytest
import torch

from oracle import conv_oracle as co

pytestmark = pytest.mark.gpu

REL = 2.0 ** -7      # 2 bf16 ulp
ABS = 2e-3


@pytest.fixture
def tune():
    """set one of the library's tuning switches for this test (rotate-yolov3_amd/_lib.py: set_tuning); cleared afterwards"""
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd import _lib
    touched = set()

    def setter(name, value):
        touched.add(name)
        _lib.set_tuning(name, value)
    yield setter
    for name in touched:
        _lib.set_tuning(name, None)


@pytest.fixture(scope="module")
def ops(cuda_dev):
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.model import hip_ops
    return hip_ops


def _case(ops, dev, n, h, w, cin, cout, k, stride, act, *, real_cin=None, residual=False, upsample=1, tile=0,
          in_slice=None, out_slice=None, seed=0, ret_out=False, slope=0.1):
    g = torch.Generator().manual_seed(seed)
    real_cin = real_cin or cin
    x = co.bf16_round(torch.randn(n, real_cin, h, w, generator=g))
    wt = co.bf16_round(torch.randn(cout, real_cin, k, k, generator=g) / (real_cin * k * k) ** 0.5)
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.5
    pad = (k - 1) // 2
    ho, wo = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    res = co.bf16_round(torch.randn(n, cout, ho, wo, generator=g)) if residual else None
    actname = {0: "linear", 1: "leaky", 2: "mish"}[act]
    want = co.conv_block(x, wt, scale, shift, stride, pad, act=actname, slope=slope, residual=res, upsample=upsample)
    # tolerance scale: 2 bf16 ulp of the magnitudes that get rounded (the pre-add value and the residual)
    mag = co.conv_block(x, wt, scale, shift, stride, pad, act=actname, slope=slope, upsample=upsample).abs()
    if residual:
        mag = mag + torch.nn.functional.interpolate(res, scale_factor=upsample, mode="nearest").abs() if upsample != 1 else mag + res.abs()

    # device tensors: NHWC bf16, optionally channel slices of wider buffers
    xin = torch.zeros(n, h, w, cin, dtype=torch.bfloat16)
    xin[..., :real_cin] = x.permute(0, 2, 3, 1).to(torch.bfloat16)
    if in_slice:
        tot, off = in_slice
        buf = torch.randn(n, h, w, tot).to(torch.bfloat16)
        buf[..., off:off + cin] = xin
        xd = buf.to(dev)[..., off:off + cin]
    else:
        xd = xin.to(dev)
    packed = ops.pack_weights(wt.to(dev), cin_pad=cin)
    sc = ops.pad_vec(scale.to(dev), ops.cpad(cout))
    sh = ops.pad_vec(shift.to(dev), ops.cpad(cout))
    resd = res.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev) if residual else None
    if out_slice:
        tot, off = out_slice
        obuf = torch.full((n, ho * upsample, wo * upsample, tot), 7.0, dtype=torch.bfloat16, device=dev)
        out = obuf[..., off:off + cout]
    else:
        obuf, out = None, None
    y = ops.conv2d_bn_act(xd, packed, sc, sh, cout, k, stride=stride, act=act, slope=slope, residual=resd, out=out,
                          upsample=upsample, tile=tile)
    torch.cuda.synchronize()
    got = y.float().cpu().permute(0, 3, 1, 2)
    err = (got - want).abs()
    # with a residual the result is rounded twice (pre-add value, then the sum): one ulp flip of each can add up
    tol = (2.0 if residual else 1.0) * REL * mag + ABS
    assert bool((err <= tol).all()), "max err %.4g (tol %.4g) at %s" % (
        err.max().item(), tol.flatten()[err.argmax()].item(), np.unravel_index(err.argmax().item(), err.shape))
    if obuf is not None:   # bytes outside the slice untouched
        mask = torch.ones(obuf.shape[-1], dtype=torch.bool)
        mask[out_slice[1]:out_slice[1] + cout] = False
        assert bool((obuf[..., mask.to(dev)] == 7.0).all())
    if ret_out:
        return y.clone()
    return err.max().item()


@pytest.mark.parametrize("tile", [1, 2, 3, 4])
def test_conv3x3_s1_tiles(ops, cuda_dev, tile):
    _case(ops, cuda_dev, 2, 24, 20, 32, 64, 3, 1, 1, tile=tile, seed=tile)


@pytest.mark.parametrize("tile", [1, 2, 3, 4])
def test_conv1x1_tiles(ops, cuda_dev, tile):
    _case(ops, cuda_dev, 3, 19, 19, 128, 64, 1, 1, 1, tile=tile, seed=10 + tile)


def test_first_layer_cin3_padded_to_8(ops, cuda_dev):
    # auto dispatch: the direct kernel (fragments straight from global memory); tile 3 / 0x103: the implicit-GEMM paths
    _case(ops, cuda_dev, 2, 64, 64, 8, 32, 3, 1, 1, real_cin=3, seed=20)
    _case(ops, cuda_dev, 2, 64, 64, 8, 32, 3, 1, 1, real_cin=3, tile=3, seed=20)
    _case(ops, cuda_dev, 3, 37, 53, 8, 32, 3, 1, 1, real_cin=3, seed=21)          # odd sizes, pixel count not a multiple of 16
    _case(ops, cuda_dev, 1, 5, 3, 8, 32, 3, 1, 2, real_cin=3, seed=22)            # tiny map, mish
    a = _case(ops, cuda_dev, 2, 96, 80, 8, 32, 3, 1, 1, real_cin=3, seed=23, ret_out=True)
    b = _case(ops, cuda_dev, 2, 96, 80, 8, 32, 3, 1, 1, real_cin=3, seed=23, ret_out=True, tile=0x103)
    assert (a.float() - b.float()).abs().max().item() <= 2.0 ** -7 * b.float().abs().max().item()


def test_conv3x3_stride2(ops, cuda_dev):
    _case(ops, cuda_dev, 2, 38, 38, 64, 128, 3, 2, 1, seed=21)
    _case(ops, cuda_dev, 1, 33, 31, 32, 64, 3, 2, 1, seed=22)      # odd sizes


def test_residual_shortcut_fused(ops, cuda_dev):
    _case(ops, cuda_dev, 2, 19, 19, 128, 256, 3, 1, 1, residual=True, seed=23)


def test_head_conv_504_linear_bias(ops, cuda_dev):
    _case(ops, cuda_dev, 2, 19, 19, 1024, 504, 1, 1, 0, seed=24)


def test_deep_k_512_to_1024(ops, cuda_dev):
    _case(ops, cuda_dev, 1, 19, 19, 512, 1024, 3, 1, 1, seed=25)


def test_route_concat_slices_and_upsample(ops, cuda_dev):
    # layer 84/85/86 pattern: 1x1 512->256, nearest x2, written into channels [0,256) of a 768-wide buffer
    _case(ops, cuda_dev, 2, 10, 10, 512, 256, 1, 1, 1, upsample=2, out_slice=(768, 0), seed=26)
    # layer 61 pattern: output lands in channels [256,768) of the concat buffer; layer 62 reads that slice
    _case(ops, cuda_dev, 1, 20, 20, 256, 512, 3, 1, 1, residual=True, out_slice=(768, 256), seed=27)
    _case(ops, cuda_dev, 1, 20, 20, 512, 64, 3, 2, 1, in_slice=(768, 256), seed=28)
    # non power-of-two Cin on a 1x1 (route output 384 / 768 channels)
    _case(ops, cuda_dev, 1, 16, 16, 384, 128, 1, 1, 1, seed=29)
    # ... and on a 3x3 (yolov3-tiny layer 21: 3x3 on the 384-channel concat)
    _case(ops, cuda_dev, 1, 12, 12, 384, 256, 3, 1, 1, seed=31)


def test_pipelined_256x128_variant_deep_and_shallow_k(ops, cuda_dev):
    # tile 4 = 256x128, 3-stage ring with counted vmcnt: K steps 1, 2, 3 (ring shorter than / equal to the depth) and 72
    _case(ops, cuda_dev, 2, 19, 19, 64, 128, 1, 1, 1, tile=4, seed=40)       # KT = 1
    _case(ops, cuda_dev, 2, 19, 19, 128, 128, 1, 1, 1, tile=4, seed=41)      # KT = 2
    _case(ops, cuda_dev, 2, 19, 19, 192, 136, 1, 1, 1, tile=4, seed=42)      # KT = 3, ragged channels
    _case(ops, cuda_dev, 1, 19, 19, 512, 1024, 3, 1, 1, tile=4, residual=True, seed=43)   # KT = 72
    _case(ops, cuda_dev, 2, 38, 38, 64, 128, 3, 2, 1, tile=4, seed=44)
    _case(ops, cuda_dev, 2, 10, 10, 512, 256, 1, 1, 1, upsample=2, out_slice=(768, 0), tile=4, seed=45)


@pytest.mark.parametrize("case", [
    dict(n=4, h=160, w=160, cin=64, cout=128, k=3, stride=1, act=1, residual=True),            # 800 tiles
    dict(n=3, h=152, w=152, cin=128, cout=256, k=1, stride=1, act=1, upsample=2, out_slice=(384, 128)),   # 542 x 2
    dict(n=8, h=203, w=197, cin=64, cout=128, k=3, stride=2, act=1),                           # odd sizes, M tail
    dict(n=4, h=200, w=200, cin=64, cout=32, k=1, stride=1, act=1),                            # 256x32 tile
    dict(n=2, h=192, w=192, cin=384, cout=128, k=1, stride=1, act=0),                          # Cin not a power of two
    dict(n=2, h=100, w=100, cin=256, cout=504, k=1, stride=1, act=0, residual=True),           # ragged Cout, 4 n-tiles
    dict(n=1, h=300, w=300, cin=128, cout=128, k=3, stride=1, act=2),                          # mish
])
def test_persistent_grid_many_tiles(ops, cuda_dev, case):
    """More tiles than 2 x CU count -> the persistent kernel (next-tile prefetch under the epilogue).  Same cases with
    tile bit 0x200 (one tile per workgroup): both meet the oracle tolerance and agree bit for bit (same arithmetic)."""
    kw = dict(case)
    args = [kw.pop(k) for k in ("n", "h", "w", "cin", "cout", "k", "stride", "act")]
    a = _case(ops, cuda_dev, *args, seed=60, ret_out=True, tile=0x800, **kw)    # persistent also for the 3x3 cases
    b = _case(ops, cuda_dev, *args, seed=60, ret_out=True, tile=0x200, **kw)
    assert torch.equal(a, b)


def test_general_address_path_forced(ops, cuda_dev):
    # bit 8 of `tile` forces the general per-lane address path on shapes that normally take the FAST (scalar tap,
    # buffer-load) path, so both code paths see the same cases
    _case(ops, cuda_dev, 2, 19, 19, 128, 256, 3, 1, 1, residual=True, tile=0x101, seed=50)
    _case(ops, cuda_dev, 2, 20, 20, 64, 128, 3, 2, 1, tile=0x101, seed=51)
    _case(ops, cuda_dev, 2, 19, 19, 256, 128, 1, 1, 1, tile=0x102, seed=52)
    _case(ops, cuda_dev, 1, 19, 19, 192, 136, 1, 1, 1, tile=0x104, seed=53)


def test_mish_epilogue(ops, cuda_dev):
    _case(ops, cuda_dev, 1, 16, 16, 64, 64, 3, 1, 2, seed=30)


def test_layout_round_trip(ops, cuda_dev):
    x = torch.randn(2, 3, 17, 23, device=cuda_dev)
    y = ops.nchw_f32_to_nhwc_bf16(x)
    assert y.shape == (2, 17, 23, 8) and bool((y[..., 3:] == 0).all())
    back = ops.nhwc_bf16_to_nchw_f32(y[..., :3])
    assert torch.equal(back, x.to(torch.bfloat16).float())


# ---------------------------------------------------------------- conv_mp.hip: the 256-channel multi-phase tile (tile 8..)
MP_CASES = [
    # (n, h, w, cin, cout, k, stride, act, kwargs)
    (2, 19, 19, 128, 256, 3, 1, 1, {}),                                   # KT 18, M 722 (3 tiles, ragged tail)
    (1, 16, 16, 128, 256, 1, 1, 1, {}),                                   # KT 2 (the minimum), exactly one tile
    (1, 20, 13, 192, 256, 1, 1, 0, {}),                                   # KT 3 (odd), linear
    (1, 13, 11, 64, 256, 3, 1, 2, {}),                                    # KT 9 (odd), mish, M 143 < one tile
    (2, 19, 19, 256, 512, 3, 1, 1, dict(residual=True)),                  # two channel tiles, fused shortcut
    (1, 38, 38, 128, 256, 3, 2, 1, {}),                                   # stride 2
    (1, 33, 31, 64, 256, 3, 2, 1, {}),                                    # stride 2, odd sizes
    (2, 10, 10, 512, 256, 1, 1, 1, dict(out_slice=(768, 0))),             # output into a concat slice
    (3, 47, 29, 128, 256, 3, 1, 1, dict(residual=True)),                  # 16 tiles over 16 workgroups... ragged, residual
    (8, 40, 40, 128, 256, 1, 1, 1, dict(residual=True)),                   # 50 tiles
    (1, 20, 20, 256, 512, 3, 1, 1, dict(residual=True, out_slice=(768, 256), in_slice=(640, 128))),
    (1, 19, 19, 512, 1024, 3, 1, 1, {}),                                  # KT 72, four channel tiles
]


@pytest.mark.parametrize("case", range(len(MP_CASES)))
@pytest.mark.parametrize("tile", [8, 11, 14, 9])  # conv_mp.hip: BM 256, BM 192, BM picked per shape; 9 = conv_mq.hip (two workgroups per CU)
def test_conv_mp_tile(ops, cuda_dev, case, tile):
    n, h, w, cin, cout, k, stride, act, kw = MP_CASES[case]
    _case(ops, cuda_dev, n, h, w, cin, cout, k, stride, act, tile=tile, seed=100 + case, **kw)


def test_conv_mp_prelu_slopes(ops, cuda_dev):
    # PReLU slopes outside [0, 1] take the compare/select form, slopes <= 1 the max form: same values
    for i, sl in enumerate((0.25, 1.0, 1.5, -0.3)):
        _case(ops, cuda_dev, 1, 20, 20, 128, 256, 3, 1, 1, tile=8, seed=160 + i, slope=sl)
        _case(ops, cuda_dev, 1, 20, 20, 128, 256, 3, 1, 1, tile=1, seed=160 + i, slope=sl)


def test_conv_auto_dispatch_takes_mp_for_3x3(ops, cuda_dev):
    a = _case(ops, cuda_dev, 2, 38, 38, 128, 256, 3, 1, 1, seed=141, ret_out=True, residual=True)            # auto
    b = _case(ops, cuda_dev, 2, 38, 38, 128, 256, 3, 1, 1, seed=141, ret_out=True, residual=True, tile=14)   # mp, BM picked
    assert torch.equal(a, b)


def test_conv_mp_matches_128_tile_bitwise_modulo_order(ops, cuda_dev):
    # same products, fp32 accumulation in a different order: results agree to 1 bf16 ulp of the output
    a = _case(ops, cuda_dev, 2, 38, 38, 128, 256, 3, 1, 1, seed=140, ret_out=True, tile=1)
    b = _case(ops, cuda_dev, 2, 38, 38, 128, 256, 3, 1, 1, seed=140, ret_out=True, tile=8)
    d = (a.float() - b.float()).abs()
    assert bool((d <= 2.0 ** -7 * b.float().abs() + 1e-3).all())


@pytest.mark.parametrize("tile", [8, 9])
def test_conv_mp_many_tiles_per_workgroup(ops, cuda_dev, tile):
    # 2 x 160 x 160 pixels = 200 (400) tiles x 2 channel tiles on <= 256 (512) persistent workgroups: the chunk stream crosses
    # output-tile boundaries (incl. a change of channel tile) inside a workgroup
    _case(ops, cuda_dev, 2, 160, 160, 64, 512, 3, 1, 1, residual=True, tile=tile, seed=150)
    _case(ops, cuda_dev, 2, 160, 160, 128, 256, 1, 1, 0, tile=tile, seed=151)
    _case(ops, cuda_dev, 6, 160, 160, 64, 256, 3, 1, 2, residual=True, tile=tile, seed=152)   # 1200 tiles: three per workgroup of conv_mq


def test_conv_mq_equals_conv_mp_bitwise(ops, cuda_dev, tune):
    # same K order, same MFMA order, same epilogue arithmetic: the two wide tiles must agree bit for bit (conv_mq's channel-major K order for
    # C_in >= 512 -- round 5, another fp32 summation order -- is switched off here and compared separately below)
    tune("RYOLO_MQ_KORDER", "0")
    for seed, (n, h, w, cin, cout, k, s_, kw) in enumerate([(4, 76, 76, 128, 256, 3, 1, dict(residual=True)), (2, 38, 38, 256, 512, 3, 1, {}),
                                                             (2, 77, 75, 128, 256, 3, 2, {}), (1, 19, 19, 512, 1024, 3, 1, dict(residual=True))]):
        a = _case(ops, cuda_dev, n, h, w, cin, cout, k, s_, 1, seed=170 + seed, ret_out=True, tile=8, **kw)
        b = _case(ops, cuda_dev, n, h, w, cin, cout, k, s_, 1, seed=170 + seed, ret_out=True, tile=9, **kw)
        assert torch.equal(a, b)


def test_conv_mq_channel_major_k_order(ops, cuda_dev, tune):
    # conv_mq visits the K tiles channel-slice-major for C_in >= 512 (the nine taps of a 64-channel slice back to back: L2 locality): same
    # products, another fp32 summation order -- against the oracle, and within 1 bf16 ulp of the tap-major result; forced on for a C_in 128
    # layer (two slices: the wrap from the last tap to the next slice, and from the last slice into the NEXT OUTPUT TILE's first K tile)
    for seed, (n, h, w, cin, cout, s_, kw, force) in enumerate([(2, 38, 38, 512, 256, 1, dict(residual=True), None), (1, 19, 19, 1024, 256, 1, {}, None),
                                                                 (3, 47, 29, 128, 256, 1, dict(residual=True), "1"), (2, 39, 37, 512, 256, 2, {}, None),
                                                                 (6, 80, 80, 192, 256, 1, {}, "1")]):
        tune("RYOLO_MQ_KORDER", "0")
        a = _case(ops, cuda_dev, n, h, w, cin, cout, 3, s_, 1, seed=180 + seed, ret_out=True, tile=9, **kw)
        if force is None:
            tune("RYOLO_MQ_KORDER", None)
        else:
            tune("RYOLO_MQ_KORDER", force)
        b = _case(ops, cuda_dev, n, h, w, cin, cout, 3, s_, 1, seed=180 + seed, ret_out=True, tile=9, **kw)
        # (each result is within the oracle's bars inside _case; against each other: an ulp of the magnitudes that were rounded -- with a fused
        # shortcut the pre-add value can be much larger than the sum, so the bar is taken from the tensor's scale, not per element)
        d = (a.float() - b.float()).abs()
        assert float(d.max()) <= 2.0 ** -6 * float(a.float().abs().max()), (float(d.max()), float(a.float().abs().max()))
        assert float((d > 0).float().mean()) < 0.2       # ... and most elements do not move at all
        assert not torch.equal(a, b) or cin < 256        # (the order really changed: some element moved by an ulp)


@pytest.mark.parametrize("tile", [8, 9])
def test_conv_mp_repeatable(ops, cuda_dev, tile):
    # race screen: the multi-phase pipeline must give bit-identical output on repeated launches (20 x, bs 8 at 76^2)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 76, 76, 128, generator=g).to(torch.bfloat16).to(cuda_dev)
    wt = (torch.randn(256, 128, 3, 3, generator=g) / 34.0).to(cuda_dev)
    packed = ops.pack_weights(wt, cin_pad=128)
    sc = torch.ones(256, device=cuda_dev)
    sh = torch.zeros(256, device=cuda_dev)
    ref = ops.conv2d_bn_act(x, packed, sc, sh, 256, 3, act=1, tile=1)
    first = None
    for _ in range(20):
        y = ops.conv2d_bn_act(x, packed, sc, sh, 256, 3, act=1, tile=tile)
        if first is None:
            first = y.clone()
            d = (first.float() - ref.float()).abs()
            assert bool((d <= 2.0 ** -7 * ref.float().abs() + 1e-3).all())
        else:
            assert torch.equal(y, first)


# ---- conv_stem.hip: 3x3, 32 -> 64 channels, stride 1 / 2 (tile 12; auto picks it for these shapes)
STEM_CASES = [  # n, h, w, stride, act, kwargs
    (2, 40, 48, 1, 1, dict(residual=True)),          # whole tiles of 8 x 32
    (3, 37, 53, 2, 1, {}),                             # ragged in both directions, stride 2 (odd input size)
    (1, 8, 32, 1, 2, {}),                              # exactly one tile, Mish
    (2, 19, 65, 2, 0, dict(residual=True)),            # linear + shortcut, stride 2, 10 x 33 outputs
    (2, 50, 70, 1, 1, dict(in_slice=(96, 32), out_slice=(128, 64))),     # channel slices of wider (concat) buffers
    (5, 61, 35, 1, 1, dict(residual=True)),            # more tiles than one round of a small grid would hold
]


@pytest.mark.parametrize("case", range(len(STEM_CASES)))
def test_conv_stem_kernel_against_the_oracle_and_the_igemm_tile(ops, cuda_dev, case):
    n, h, w, s_, act, kw = STEM_CASES[case]
    assert ops.conv_kernel_name(n, h, w, 32, 64, 3, s_).startswith("conv3x3_c32_halo")       # what auto dispatches
    a = _case(ops, cuda_dev, n, h, w, 32, 64, 3, s_, act, seed=300 + case, ret_out=True, tile=12, **kw)
    b = _case(ops, cuda_dev, n, h, w, 32, 64, 3, s_, act, seed=300 + case, ret_out=True, tile=2, **kw)
    c = _case(ops, cuda_dev, n, h, w, 32, 64, 3, s_, act, seed=300 + case, ret_out=True, tile=0, **kw)
    # one K = 32 MFMA per tap, taps in the same order as the implicit-GEMM K loop: bit-identical
    assert torch.equal(a, b) and torch.equal(a, c)


# ---- conv_stem.hip, round 6: 3x3 / 1, 64 -> 128 channels (tile 17; auto picks it): input patch staged once, the filter split over the waves
STEM64_CASES = [  # n, h, w, act, kwargs
    (2, 40, 64, 1, dict(residual=True)),             # whole tiles of 4 x 32
    (3, 37, 53, 1, {}),                                # ragged in both directions
    (1, 4, 32, 2, {}),                                 # exactly one tile, Mish
    (2, 19, 65, 0, dict(residual=True)),               # linear + shortcut
    (2, 22, 70, 1, dict(in_slice=(160, 64), out_slice=(256, 128))),      # channel slices of wider (concat) buffers
    (5, 61, 35, 1, dict(residual=True)),               # more tiles than one round of a small grid would hold
]


@pytest.mark.parametrize("case", range(len(STEM64_CASES)))
def test_conv_stem64_kernel_against_the_oracle_and_the_igemm_tiles(ops, cuda_dev, case):
    n, h, w, act, kw = STEM64_CASES[case]
    assert ops.conv_kernel_name(n, h, w, 64, 128, 3, 1) == "conv3x3_c64_halo"                  # what auto dispatches
    a = _case(ops, cuda_dev, n, h, w, 64, 128, 3, 1, act, seed=400 + case, ret_out=True, tile=17, **kw)
    b = _case(ops, cuda_dev, n, h, w, 64, 128, 3, 1, act, seed=400 + case, ret_out=True, tile=1, **kw)
    c = _case(ops, cuda_dev, n, h, w, 64, 128, 3, 1, act, seed=400 + case, ret_out=True, tile=0, **kw)
    # taps in the order 0..8, channels 0..31 then 32..63 inside a tap = the K order of the implicit-GEMM kernels: bit-identical
    assert torch.equal(a, b) and torch.equal(a, c)


def test_conv_stem64_kernel_statistics_and_repeatability(ops, cuda_dev):
    from rotate_yolov3_amd.model import hip_train_ops as tr
    g = torch.Generator().manual_seed(12)
    for (n, h, w) in ((3, 45, 70), (2, 77, 131), (1, 3, 5)):
        x = torch.randn(n, h, w, 64, generator=g).to(torch.bfloat16).to(cuda_dev)
        wt = (torch.randn(128, 64, 3, 3, generator=g) / 24.0).to(cuda_dev)
        packed = ops.pack_weights(wt, cin_pad=64)
        ones, zeros = torch.ones(128, device=cuda_dev), torch.zeros(128, device=cuda_dev)
        res = {}
        for tile in (17, 1):
            d = tr.make_desc(x, 128, 3, 1, 1, tile=tile)
            z = torch.full((n, h, w, 128), 7.0, dtype=torch.bfloat16, device=cuda_dev)
            part = tr.conv_fwd_stats(d, x, packed, ones, zeros, z)
            torch.cuda.synchronize()
            res[tile] = (z.clone(), part[:, 0, :128].sum(0).clone(), part[:, 1, :128].sum(0).clone())
        assert torch.equal(res[17][0], res[1][0])
        zf = res[17][0].double()
        assert torch.allclose(res[17][1], zf.sum((0, 1, 2)), rtol=1e-5, atol=1e-2)            # sums of the values as stored
        assert torch.allclose(res[17][2], (zf * zf).sum((0, 1, 2)), rtol=1e-5, atol=1e-2)
        assert torch.allclose(res[17][1], res[1][1], rtol=1e-5, atol=1e-2)
        first = None
        for _ in range(10):                                   # race screen: the double-buffered patch, one barrier per tile
            y = ops.conv2d_bn_act(x, packed, ones, zeros, 128, 3, stride=1, act=1, tile=17)
            if first is None:
                first = y.clone()
            else:
                assert torch.equal(y, first)


def test_conv_stem_kernel_statistics_and_repeatability(ops, cuda_dev):
    from rotate_yolov3_amd.model import hip_train_ops as tr
    g = torch.Generator().manual_seed(11)
    for s_, (n, h, w) in ((1, (3, 45, 70)), (2, (2, 77, 131))):
        x = torch.randn(n, h, w, 32, generator=g).to(torch.bfloat16).to(cuda_dev)
        wt = (torch.randn(64, 32, 3, 3, generator=g) / 17.0).to(cuda_dev)
        packed = ops.pack_weights(wt, cin_pad=32)
        ones, zeros = torch.ones(128, device=cuda_dev), torch.zeros(128, device=cuda_dev)
        ho, wo = (h + 2 - 3) // s_ + 1, (w + 2 - 3) // s_ + 1
        res = {}
        for tile in (12, 2):
            d = tr.make_desc(x, 64, 3, s_, 1, tile=tile)
            z = torch.full((n, ho, wo, 64), 7.0, dtype=torch.bfloat16, device=cuda_dev)
            part = tr.conv_fwd_stats(d, x, packed, ones, zeros, z)
            torch.cuda.synchronize()
            res[tile] = (z.clone(), part[:, 0, :64].sum(0).clone(), part[:, 1, :64].sum(0).clone())
        assert torch.equal(res[12][0], res[2][0])
        zf = res[12][0].double()
        assert torch.allclose(res[12][1], zf.sum((0, 1, 2)), rtol=1e-5, atol=1e-2)            # sums of the values as stored
        assert torch.allclose(res[12][2], (zf * zf).sum((0, 1, 2)), rtol=1e-5, atol=1e-2)
        assert torch.allclose(res[12][1], res[2][1], rtol=1e-5, atol=1e-2)
        first = None
        for _ in range(10):                                   # race screen: the double-buffered patch, one barrier per tile
            y = ops.conv2d_bn_act(x, packed, ones, zeros, 64, 3, stride=s_, act=1, tile=12)
            if first is None:
                first = y.clone()
            else:
                assert torch.equal(y, first)


def _pair_case(ops, dev, kind, n, h, w, act=1, seed=0):
    """two stem layers in one launch (csrc/conv_stem.hip conv_stem_pair_kernel) against the same two layers launched one after the
    other: bit for bit (same MFMAs in the same order, the tensor between them rounded to bf16 either way)"""
    g = torch.Generator().manual_seed(seed)
    if True:
        cin, first, second, shortcut = 64, dict(cout=32, ksize=1, stride=1, pad=0, act=act, slope=0.1), dict(cout=64, ksize=3, stride=1, pad=1, act=act, slope=0.25), True
        x = torch.randn(n, h, w, 64, generator=g)
        w1 = torch.randn(32, 64, 1, 1, generator=g) / 8.0
    w2 = torch.randn(64, 32, 3, 3, generator=g) / 288 ** 0.5
    xd = x.to(torch.bfloat16).to(dev)
    pk1 = ops.pack_weights(w1.to(dev), cin_pad=cin)
    pk2 = ops.pack_weights(w2.to(dev), cin_pad=32)
    sc1, sh1 = ops.pad_vec((torch.rand(32, generator=g) + 0.5).to(dev), 128), ops.pad_vec((torch.randn(32, generator=g) * 0.5).to(dev), 128)
    sc2, sh2 = ops.pad_vec((torch.rand(64, generator=g) + 0.5).to(dev), 128), ops.pad_vec((torch.randn(64, generator=g) * 0.5).to(dev), 128)
    assert ops.conv_pair_supported(xd, first, second, shortcut)
    mid = ops.conv2d_bn_act(xd, pk1, sc1, sh1, 32, first['ksize'], stride=1, pad=first['pad'], act=act, slope=0.1)
    want = ops.conv2d_bn_act(mid, pk2, sc2, sh2, 64, 3, stride=second['stride'], pad=1, act=act, slope=second['slope'],
                             residual=xd if shortcut else None)
    got = ops.conv2d_bn_act_pair(xd, first, second, pk1, sc1, sh1, pk2, sc2, sh2, shortcut_from_input=shortcut)
    torch.cuda.synchronize()
    assert got.shape == want.shape
    if act == 2:
        # Mish: exp / divide are expanded by the compiler per kernel (different contraction choices in the two epilogues): the 32-channel
        # values may differ in their last bf16 bit, and with them the sums downstream -- one bf16 ulp of the output's magnitude
        err = (got.float() - want.float()).abs()
        assert bool((err <= 2.0 ** -7 * want.float().abs() + 2.0 ** -6).all()), float(err.max())
        return
    assert torch.equal(got, want), float((got.float() - want.float()).abs().max())
    # into a channel slice of a wider buffer
    obuf = torch.full((want.shape[0], want.shape[1], want.shape[2], 96), 5.0, dtype=torch.bfloat16, device=dev)
    ops.conv2d_bn_act_pair(xd, first, second, pk1, sc1, sh1, pk2, sc2, sh2, shortcut_from_input=shortcut, out=obuf[..., 16:80])
    torch.cuda.synchronize()
    assert torch.equal(obuf[..., 16:80], want) and bool((obuf[..., :16] == 5.0).all()) and bool((obuf[..., 80:] == 5.0).all())


@pytest.mark.parametrize("n,h,w", [(2, 32, 32), (1, 48, 80), (3, 21, 37), (2, 304 // 2, 304 // 2)])
def test_stem_pair_layers_2_4_equals_two_launches(ops, cuda_dev, n, h, w):
    _pair_case(ops, cuda_dev, 2, n, h, w, seed=h + w)
    _pair_case(ops, cuda_dev, 2, n, h, w, act=2, seed=w)


def test_stem_pair_is_refused_where_it_does_not_apply(ops, cuda_dev):
    x = torch.zeros(1, 32, 32, 64, dtype=torch.bfloat16, device=cuda_dev)
    a1, b3 = dict(cout=32, ksize=1, stride=1, pad=0, act=1, slope=0.1), dict(cout=64, ksize=3, stride=1, pad=1, act=1, slope=0.1)
    assert ops.conv_pair_supported(x, a1, b3, True) and not ops.conv_pair_supported(x, a1, b3, False)
    assert not ops.conv_pair_supported(x, a1, dict(b3, cout=128), True)
    assert not ops.conv_pair_supported(x, dict(a1, cout=64), b3, True)
    assert not ops.conv_pair_supported(torch.zeros(1, 32, 32, 128, dtype=torch.bfloat16, device=cuda_dev), a1, b3, True)
