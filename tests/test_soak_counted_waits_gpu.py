"""GPU tier: soak test of the kernels whose correctness rests on COUNTED `s_waitcnt vmcnt(N)` arithmetic (VERDICT r4 weak #2 / next #4c):
conv_pw.hip, conv_mq.hip (256- and 128-channel tiles), conv_mp.hip, conv_stem.hip (halo forward, stem pair, one-launch data gradients,
layer-0 staged forward), conv0_bwd.hip, wgrad_wide.  A wrong count is a race, and round 4's conv_pw race showed once in ~300 runs of a
unit test on an otherwise idle chip.  Here every case is launched 200 times -- under different workgroup-per-CU grid caps where the
kernel has one, and while a SECOND stream keeps the chip busy with (a) an HBM-bound copy loop, (b) an MFMA-bound GEMM loop, so that
load latencies, arrival orders of the waves and the co-resident workgroup change from launch to launch -- and every launch must
reproduce the first launch bit for bit.  (The first launch itself is pinned to the oracle / fp32 ATen by the per-kernel tests.)"""
import pytest
import torch

pytestmark = pytest.mark.gpu

REPS = 200
ROUND = 20          # foreground launches per background burst


@pytest.fixture(scope="module")
def T(cuda_dev):
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.model import hip_ops, hip_train_ops

    class NS:
        ops, tr = hip_ops, hip_train_ops
    return NS


class _Background(object):
    """keeps a side stream busy: kind 0 = nothing, 1 = HBM-bound copies of a 256-MB tensor, 2 = MFMA-bound bf16 GEMMs"""

    def __init__(self, dev):
        self.dev = dev
        self.stream = torch.cuda.Stream(dev)
        self.src = torch.empty(128 << 20, dtype=torch.bfloat16, device=dev).normal_()
        self.dst = torch.empty_like(self.src)
        self.a = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
        self.b = torch.randn(4096, 4096, device=dev).to(torch.bfloat16)
        self.c = torch.empty(4096, 4096, dtype=torch.bfloat16, device=dev)

    def burst(self, kind, n=40):
        if kind == 0:
            return
        with torch.cuda.stream(self.stream):
            for _ in range(n):
                if kind == 1:
                    self.dst.copy_(self.src)
                else:
                    torch.mm(self.a, self.b, out=self.c)


def _soak(dev, bg, launch, outputs, label):
    """launch() REPS times; outputs() -> list of tensors that must be bit-identical to the first launch's every time"""
    first = None
    for rep in range(REPS):
        if rep % ROUND == 0:
            torch.cuda.synchronize(dev)
            bg.burst((rep // ROUND) % 3)
        launch()
        got = [t.clone() for t in outputs()]
        if first is None:
            torch.cuda.synchronize(dev)
            first = got
        else:
            for k, (a, b) in enumerate(zip(got, first)):
                assert torch.equal(a, b), "%s: launch %d differs from launch 0 in output %d (max |d| %g)" % (
                    label, rep, k, float((a.float() - b.float()).abs().max()))
    torch.cuda.synchronize(dev)


def _conv_inputs(dev, n, h, w, cin, cout, k, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16).to(dev)
    wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
    res = torch.randn(n, h, w, cout, generator=g).to(torch.bfloat16).to(dev)
    return x, wt, res


CONV_CASES = [
    # label, (n, h, w, cin, cout, k, stride), tile codes (low byte = pick, bits 16.. = conv_pw's workgroups-per-XCD cap), residual
    ("conv_pw 512->256", (8, 38, 38, 512, 256, 1, 1), [13, 13 | (2 << 16), 13 | (5 << 16)], False),
    ("conv_pw 256->128", (4, 76, 76, 256, 128, 1, 1), [13, 13 | (3 << 16)], False),
    ("conv_pw 768->256", (8, 38, 38, 768, 256, 1, 1), [13], False),
    ("conv_mq 128->256 3x3", (4, 76, 76, 128, 256, 3, 1), [9], True),
    ("conv_mq 256->512 3x3", (8, 38, 38, 256, 512, 3, 1), [9], False),
    ("conv_mp 512->1024 3x3", (8, 19, 19, 512, 1024, 3, 1), [8, 11], True),            # C_in 512: the channel-major K order (round 5)
    ("conv_mq 512->256 3x3", (8, 38, 38, 512, 256, 3, 1), [9], True),                  # ... in conv_mq, with the two-block store order
    ("conv_stem 32->64 s1", (4, 152, 152, 32, 64, 3, 1), [12], True),
    ("conv_stem 32->64 s2", (4, 152, 152, 32, 64, 3, 2), [12], False),
]


@pytest.mark.parametrize("case", range(len(CONV_CASES)))
def test_soak_forward_kernels(T, cuda_dev, case):
    label, (n, h, w, cin, cout, k, s), tiles, residual = CONV_CASES[case]
    ops, dev = T.ops, cuda_dev
    bg = _Background(dev)
    x, wt, res = _conv_inputs(dev, n, h, w, cin, cout, k, 40 + case)
    packed = ops.pack_weights(wt, cin_pad=cin)
    sc = torch.rand(ops.cpad(cout), device=dev) + 0.5
    sh = torch.randn(ops.cpad(cout), device=dev) * 0.3
    ho = (h + 2 * ((k - 1) // 2) - k) // s + 1
    wo = (w + 2 * ((k - 1) // 2) - k) // s + 1
    resd = res[:, :ho, :wo].contiguous() if residual else None
    for tile in tiles:
        out = torch.empty(n, ho, wo, cout, dtype=torch.bfloat16, device=dev)

        def launch():
            out.fill_(3.0)
            ops.conv2d_bn_act(x, packed, sc, sh, cout, k, stride=s, act=1, slope=0.1, residual=resd, out=out, tile=tile)
        _soak(dev, bg, launch, lambda: [out], "%s tile 0x%x" % (label, tile))


def test_soak_stem_pair_and_layer0(T, cuda_dev):
    ops, dev = T.ops, cuda_dev
    bg = _Background(dev)
    g = torch.Generator().manual_seed(77)
    # layers 2-4 as one launch (1x1 64 -> 32 into LDS, 3x3 32 -> 64 + shortcut)
    x = torch.randn(4, 152, 152, 64, generator=g).to(torch.bfloat16).to(dev)
    w1 = (torch.randn(32, 64, 1, 1, generator=g) / 8.0).to(dev)
    w2 = (torch.randn(64, 32, 3, 3, generator=g) / 17.0).to(dev)
    pk1, pk2 = ops.pack_weights(w1, cin_pad=64), ops.pack_weights(w2, cin_pad=32)
    sc1, sh1 = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.3
    sc2, sh2 = torch.rand(128, device=dev) + 0.5, torch.randn(128, device=dev) * 0.3
    first = dict(cout=32, ksize=1, stride=1, pad=0, act=1, slope=0.1)
    second = dict(cout=64, ksize=3, stride=1, pad=1, act=1, slope=0.1)
    assert ops.conv_pair_supported(x, first, second, True)
    out = torch.empty(4, 152, 152, 64, dtype=torch.bfloat16, device=dev)
    _soak(dev, bg, lambda: ops.conv2d_bn_act_pair(x, first, second, pk1, sc1, sh1, pk2, sc2, sh2, shortcut_from_input=True, out=out),
          lambda: [out], "conv_stem_pair")
    # layer 0 forward with its input patch staged in LDS (auto dispatch for 8 -> 32)
    x0 = torch.zeros(4, 160, 160, 8, dtype=torch.bfloat16, device=dev)
    x0[..., :3] = torch.rand(4, 160, 160, 3, generator=g).to(torch.bfloat16).to(dev)
    w0 = (torch.randn(32, 8, 3, 3, generator=g) / 5.0).to(dev)
    w0[:, 3:] = 0
    pk0 = ops.pack_weights(w0, cin_pad=8)
    out0 = torch.empty(4, 160, 160, 32, dtype=torch.bfloat16, device=dev)
    assert ops.conv_kernel_name(4, 160, 160, 8, 32, 3) == 'conv0_halo<c8>'
    _soak(dev, bg, lambda: ops.conv2d_bn_act(x0, pk0, sc1, sh1, 32, 3, act=1, slope=0.1, out=out0), lambda: [out0], "conv0_halo")


def test_soak_backward_kernels(T, cuda_dev):
    ops, tr, dev = T.ops, T.tr, cuda_dev
    bg = _Background(dev)
    g = torch.Generator().manual_seed(78)
    # conv0_bwd.hip: layer 0's whole backward in one pass
    n, h, w = 2, 96, 96
    x = torch.zeros(n, h, w, 8, dtype=torch.bfloat16, device=dev)
    x[..., :3] = torch.randn(n, h, w, 3, generator=g).to(torch.bfloat16).to(dev)
    wt = (torch.randn(32, 3, 3, 3, generator=g) / 5.0).to(dev)
    packed = ops.pack_weights(wt, cin_pad=8)
    ones, zeros = torch.ones(128, device=dev), torch.zeros(128, device=dev)
    d = tr.make_desc(x, 32, 3, 1, 1)
    part = tr.conv_fwd_stats(d, x, packed, ones, zeros, None)
    gamma, beta = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.3
    st = tr.bn_finalize(part, 32, n * h * w, gamma, beta)
    slope = torch.tensor([0.1], device=dev)
    dy = torch.randn(n, h, w, 32, generator=g).to(torch.bfloat16).to(dev)
    wsf = tr.conv0_bn_bwd_wgrad_ws(dev)
    gw, dg, db, ds = torch.zeros(32, 3, 3, 3, device=dev), torch.zeros(32, device=dev), torch.zeros(32, device=dev), torch.zeros(1, device=dev)

    def launch0():
        gw.zero_(); dg.zero_(); db.zero_(); ds.zero_()
        tr.conv0_bn_bwd_wgrad(d, x, packed, dy, st, 1, slope, dg, db, ds, gw, 3, True, wsf)
    _soak(dev, bg, launch0, lambda: [gw, dg, db, ds], "conv0_bwd")
    # the stem's one-launch data gradients (conv_stem.hip) and a conv_mq data gradient with accumulation
    for (n, hh, ww, cin, cout, k, s) in [(3, 70, 130, 32, 64, 3, 2), (3, 70, 130, 32, 64, 3, 1), (2, 70, 66, 64, 128, 3, 2),
                                         (4, 38, 38, 256, 512, 3, 1), (4, 76, 76, 128, 256, 3, 1)]:
        wt = (torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5).to(dev)
        xin = torch.empty(n, hh, ww, cin, dtype=torch.bfloat16, device=dev)
        dd = tr.make_desc(xin, cout, k, s, 1)
        pk = tr.pack_weights_dgrad(wt, s)
        ho, wo = (hh + 2 - k) // s + 1, (ww + 2 - k) // s + 1
        dz = torch.randn(n, ho, wo, cout, generator=g).to(torch.bfloat16).to(dev)
        prev = torch.randn(n, hh, ww, cin, generator=g).to(torch.bfloat16).to(dev)
        o1, z1 = torch.ones(ops.cpad(cin), device=dev), torch.zeros(ops.cpad(cin), device=dev)
        dx = torch.empty_like(prev)

        def launch1():
            dx.copy_(prev)
            tr.conv_dgrad(dd, dz, pk, o1, z1, dx, True)
        _soak(dev, bg, launch1, lambda: [dx], "dgrad %d->%d k%d s%d" % (cin, cout, k, s))
    # wgrad_wide<256,128> (three LDS stages, counted waits): 3x3 128 -> 256 at 38^2
    n, hh, cin, cout = 8, 38, 128, 256
    xw = torch.randn(n, hh, hh, cin, generator=g).to(torch.bfloat16).to(dev)
    dzw = torch.randn(n, hh, hh, cout, generator=g).to(torch.bfloat16).to(dev)
    dw_desc = tr.make_desc(xw, cout, 3, 1, 1)
    wsw = torch.empty(tr.wgrad_ws_bytes(dw_desc), dtype=torch.uint8, device=dev)
    gw2 = torch.zeros(cout, cin, 3, 3, device=dev)

    def launch2():
        gw2.zero_()
        tr.conv_wgrad(dw_desc, xw, dzw, cin, gw2, True, wsw)
    _soak(dev, bg, launch2, lambda: [gw2], "wgrad_wide")
