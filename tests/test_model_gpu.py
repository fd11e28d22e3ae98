"""GPU tier: the HIP engine (conv stack + decode through the C ABI) and the NMS wrapper vs the oracle / goldens."""
import os

import numpy as np
import pytest
import torch

import rotate_yolov3_amd  # noqa: F401
import oracle
from oracle import darknet_oracle as do
from rotate_yolov3_amd.cfg import make_cfg
from rotate_yolov3_amd.model.models import Darknet
from tests.procedural import fill_procedural

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
torch.set_num_threads(oracle.host_cores(16))


def _cmp(name, got, want, rel_max, rel_mean):
    # bars: DESIGN.md section 4 claims max 0.6 % / mean 0.02-0.1 % under the same bf16 contract (measured r2: <= 0.67 % / 0.12 %);
    # the asserts allow about 2x that
    got, want = got.float().cpu(), want.float()
    scale = want.abs().mean().item() + 1e-6
    err = (got - want).abs()
    e_max = (err / (want.abs() + scale)).max().item()
    e_mean = err.mean().item() / scale
    print("%s: max rel %.4g  mean rel %.4g  (scale %.4g)" % (name, e_max, e_mean, scale))
    assert e_max <= rel_max and e_mean <= rel_mean, (name, e_max, e_mean)


def _model(cfg, dev):
    m = fill_procedural(Darknet(cfg, {"context_factor": 1.0}).eval())
    return m, m.to(dev)


@pytest.mark.parametrize("bs,size", [(1, 64), (2, 96)])
def test_darknet53_engine_vs_oracle(cuda_dev, bs, size):
    cfg = make_cfg.darknet53()
    m, mg = _model(cfg, cuda_dev)
    x = torch.rand(bs, 3, size, size, generator=torch.Generator().manual_seed(bs))
    with torch.no_grad():
        io, p = mg(x.to(cuda_dev))
    torch.cuda.synchronize()
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    io_o, p_o = do.forward(cfg, sd, x, bf16=True)
    assert io.shape == io_o.shape and io.dtype == torch.float32
    for k in range(3):
        _cmp("p%d" % k, p[k], p_o[k], rel_max=0.015, rel_mean=0.0025)
    _cmp("io", io, io_o, rel_max=0.015, rel_mean=0.0025)


def test_darknet53_engine_golden_from_reference(cuda_dev):
    # fp32 reference output (tests/golden/forward_d53_64.npz) vs the bf16 engine: bf16-level agreement
    z = np.load(os.path.join(G, "forward_d53_64.npz"))
    cfg = make_cfg.darknet53()
    m, mg = _model(cfg, cuda_dev)
    with torch.no_grad():
        io, p = mg(torch.from_numpy(z["x"]).to(cuda_dev))
    _cmp("io vs fp32 reference", io, torch.from_numpy(z["io"]), rel_max=0.015, rel_mean=0.0025)


def test_tiny_engine_vs_oracle(cuda_dev):
    cfg = make_cfg.tiny()
    m, mg = _model(cfg, cuda_dev)
    x = torch.rand(2, 3, 96, 96, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        io, p = mg(x.to(cuda_dev))
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    io_o, p_o = do.forward(cfg, sd, x, bf16=True)
    _cmp("tiny p0", p[0], p_o[0], rel_max=0.015, rel_mean=0.0025)
    _cmp("tiny io", io, io_o, rel_max=0.015, rel_mean=0.0025)


def test_full_size_608_vs_oracle(cuda_dev):
    cfg = make_cfg.darknet53()
    m, mg = _model(cfg, cuda_dev)
    x = torch.rand(2, 3, 608, 608, generator=torch.Generator().manual_seed(7))
    with torch.no_grad():
        io, p = mg(x.to(cuda_dev))
    assert io.shape == (2, 545832, 7)
    sd = {k: v.cpu() for k, v in m.state_dict().items()}
    io_o, p_o = do.forward(cfg, sd, x, bf16=True)
    _cmp("608 io", io, io_o, rel_max=0.02, rel_mean=0.0025)
    _cmp("608 p2", p[2], p_o[2], rel_max=0.015, rel_mean=0.0025)
    # batch independence (size-independent property): image 1 alone gives the same rows, bit for bit
    with torch.no_grad():
        io1, _ = mg(x[1:].to(cuda_dev))
    assert torch.equal(io1[0], io[1])


def test_configs1_bs32_608_forward_equals_the_oracle_checked_bs2_rows(cuda_dev):
    """BASELINE configs[1] as written: Darknet-53 forward, bs 32, 608^2, through the product path (Darknet.forward -> HipEngine).
    The batch is the two images test_full_size_608_vs_oracle checks against the oracle, 16 times: (1) every copy of an image gives the
    same rows bit for bit (the kernels' tile walks differ per copy); (2) layer by layer, the bs-32 activations of images 0 / 1 equal
    the bs-2 engine's -- bit for bit as long as every earlier layer was and the dispatch names the same kernel or a kernel pinned
    bit-identical to it, within 2 bf16 ulp of the layer's scale at the first layer where the batch-dependent dispatch changes
    kernels; (3) the decoded rows agree with the bs-2 rows inside the bars of the oracle test; (4) the batch-dependent dispatch is
    what DESIGN 3.1 says it is at this size (conv_mq on the 76^2 / 38^2 wide layers, conv_mp<192> at 19^2, fused heads)."""
    from rotate_yolov3_amd.model.engine import HipEngine
    cfg = make_cfg.darknet53()
    m, mg = _model(cfg, cuda_dev)
    x2 = torch.rand(2, 3, 608, 608, generator=torch.Generator().manual_seed(7)).to(cuda_dev)
    x32 = x2.repeat(16, 1, 1, 1)
    with torch.no_grad():
        e2 = HipEngine(mg, x2.shape, cuda_dev)
        io2, p2 = e2(x2)
        e32 = HipEngine(mg, x32.shape, cuda_dev)
        io32, p32 = e32(x32)
    torch.cuda.synchronize()
    assert io32.shape == (32, 545832, 7)
    # (1) copies
    for i in range(2, 32):
        assert torch.equal(io32[i], io32[i % 2]), i
        for k in range(3):
            assert torch.equal(p32[k][i], p32[k][i % 2]), (i, k)
    # (4) dispatch at bs 32
    names32 = {o['layer']: o['name'] for o in e32.op_info if o['kind'] == 'conv'}
    names2 = {o['layer']: o['name'] for o in e2.op_info if o['kind'] == 'conv'}
    cnt = {}
    for v in names32.values():
        cnt[v] = cnt.get(v, 0) + 1
    print("bs-32 dispatch:", sorted(cnt.items()))
    assert cnt.get('conv_mq<k3,128x256>', 0) == 24 and cnt.get('conv_mp<k3,192x256>', 0) == 8, cnt
    assert sum(v for k, v in cnt.items() if k.endswith('+decode')) == 3, cnt
    # (2) layer by layer
    exact, first_change = True, None
    for i in sorted(names32):
        a, b = e32.views[i], e2.views[i]
        if a is None or b is None or a.shape[1:] != b.shape[1:]:
            continue
        same = torch.equal(a[:2], b)
        if exact and names32[i] == names2.get(i):
            assert same, ("layer %d: same kernel %s, same input bits, different output" % (i, names32[i]))
        if not same:
            if exact:
                first_change = (i, names32[i], names2.get(i))
            exact = False
            af, bf = a[:2].float(), b.float()
            scale = bf.abs().mean().item() + 1e-6
            assert float((af - bf).abs().max()) <= 0.05 * (float(bf.abs().max()) + scale), (i, names32[i], names2.get(i))
    print("bs 32 vs bs 2: activations bit-identical through every layer" if exact else
          "bs 32 vs bs 2: first differing layer %s" % (first_change,))
    # (3) decoded rows
    if exact:
        assert torch.equal(io32[:2], io2)
    _cmp("bs-32 io vs bs-2 io", io32[:2], io2.cpu(), rel_max=0.02, rel_mean=0.0025)
    _cmp("bs-32 p2 vs bs-2 p2", p32[2][:2], p2[2].cpu(), rel_max=0.015, rel_mean=0.0025)


def test_graph_replay_equals_eager(cuda_dev):
    from rotate_yolov3_amd.model.engine import HipEngine
    cfg = make_cfg.darknet53()
    m, mg = _model(cfg, cuda_dev)
    x = torch.rand(2, 3, 128, 128, generator=torch.Generator().manual_seed(9)).to(cuda_dev)
    with torch.no_grad():
        io_e, _ = mg(x)
        io_e = io_e.clone()
        eng = HipEngine(mg, x.shape, cuda_dev, use_graph=True)
        io_g, _ = eng(x)
        io_g2, _ = eng(x)
    torch.cuda.synchronize()
    assert torch.equal(io_g, io_e) and torch.equal(io_g2, io_e)


def test_decode_kernel_vs_oracle_and_golden(cuda_dev):
    import ctypes as C
    from rotate_yolov3_amd import _lib
    from rotate_yolov3_amd.model import engine  # noqa: F401  (declares the symbol)
    z = np.load(os.path.join(G, "decode_head0.npz"))
    head = torch.from_numpy(z["head"]).to(torch.bfloat16)                      # [1,504,4,4]
    want_io, want_p = do.decode(head.float(), z["anchors"], (128, 128))
    hd = head.permute(0, 2, 3, 1).contiguous().to(cuda_dev)                    # NHWC
    anchors = torch.tensor(z["anchors"], dtype=torch.float32, device=cuda_dev)
    io = torch.empty(1, 1152, 7, device=cuda_dev)
    p = torch.empty(1, 72, 4, 4, 7, device=cuda_dev)
    rc = _lib.lib().ryolo_yolo_decode(hd.data_ptr(), 504, 1, 4, 4, 72, 7, anchors.data_ptr(), 32.0, 1.0, 0,
                                      io.data_ptr(), 1152, 0, p.data_ptr(), _lib.stream_ptr(cuda_dev))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(p.cpu(), want_p)
    assert np.allclose(io.cpu().numpy(), want_io.numpy(), rtol=2e-5, atol=2e-5)
    # against the reference's own decode of the un-rounded head: bf16 input rounding only
    assert np.allclose(io.cpu().numpy(), z["io"], rtol=2e-2, atol=2e-2)


def test_decode_kernel_multiclass_generic_path(cuda_dev):
    from rotate_yolov3_amd import _lib
    from rotate_yolov3_amd.model import engine  # noqa: F401
    g = torch.Generator().manual_seed(4)
    na, nc, ny, nx, bs = 8, 3, 5, 7, 3
    no = nc + 6
    head = torch.randn(bs, na * no, ny, nx, generator=g).to(torch.bfloat16)
    anchors = np.abs(np.random.default_rng(0).normal(40, 10, (na, 3)))
    anchors[:, 2] = np.linspace(-1.2, 1.2, na)
    for cf in (1.0, 1.25):
        want_io, want_p = do.decode(head.float(), anchors, (ny * 16, nx * 16), cf=cf, arc="default", nc=nc)
        hd = head.permute(0, 2, 3, 1).contiguous().to(cuda_dev)
        io = torch.zeros(bs, 10 + na * ny * nx, no, device=cuda_dev)
        p = torch.empty(bs, na, ny, nx, no, device=cuda_dev)
        a = torch.tensor(anchors, dtype=torch.float32, device=cuda_dev)
        rc = _lib.lib().ryolo_yolo_decode(hd.data_ptr(), na * no, bs, ny, nx, na, no, a.data_ptr(), 16.0, cf, 0,
                                          io.data_ptr(), 10 + na * ny * nx, 10, p.data_ptr(), _lib.stream_ptr(cuda_dev))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(p.cpu(), want_p)
        assert np.allclose(io[:, 10:].cpu().numpy(), want_io.numpy(), rtol=2e-5, atol=2e-5)
        assert bool((io[:, :10] == 0).all())          # row offset respected


def test_nms_wrapper_matches_reference_golden(cuda_dev):
    from rotate_yolov3_amd.utils.nms.nms import non_max_suppression
    z = np.load(os.path.join(G, "nms_wrapper.npz"))
    pred = torch.from_numpy(z["pred"].copy()).to(cuda_dev)
    out = non_max_suppression(pred, 0.3, 0.5)
    assert np.array_equal(out[0].cpu().numpy(), z["det0"])
    assert np.array_equal(out[1].cpu().numpy(), z["det1"])
    assert np.array_equal(pred.cpu().numpy(), z["pred_after"], equal_nan=True)
    empty = non_max_suppression(torch.zeros(1, 10, 7, device=cuda_dev), 0.5, 0.5)
    assert empty == [None]


def test_batched_nms_wrapper_equals_loop_and_reference_golden(cuda_dev):
    """non_max_suppression_batched (one segmented NMS launch for all images x classes) returns exactly what the
    per-image / per-class loop returns -- on the reference's golden case and on a random multi-class batch with an
    empty image, score ties and rows that fail the size / finiteness filter; the in-place score update matches too."""
    import math
    from rotate_yolov3_amd.utils.nms.nms import non_max_suppression, non_max_suppression_batched
    z = np.load(os.path.join(G, "nms_wrapper.npz"))
    pred = torch.from_numpy(z["pred"].copy()).to(cuda_dev)
    out = non_max_suppression_batched(pred, 0.3, 0.5)
    assert np.array_equal(out[0].cpu().numpy(), z["det0"])
    assert np.array_equal(out[1].cpu().numpy(), z["det1"])
    assert np.array_equal(pred.cpu().numpy(), z["pred_after"], equal_nan=True)
    assert non_max_suppression_batched(torch.zeros(1, 10, 7, device=cuda_dev), 0.5, 0.5) == [None]

    g = torch.Generator().manual_seed(11)
    bs, n, nc = 5, 3000, 3
    p = torch.empty(bs, n, 6 + nc)
    p[..., 0:2] = torch.rand(bs, n, 2, generator=g) * 200
    p[..., 2:4] = 6 * 8 ** torch.rand(bs, n, 2, generator=g)
    p[..., 4] = (torch.rand(bs, n, generator=g) - 0.5) * math.pi
    p[..., 5] = torch.rand(bs, n, generator=g)
    p[..., 6:] = torch.rand(bs, n, nc, generator=g)
    p[1, :, 5] = 0.0                                   # an image with no detections
    p[2, :50, 5] = 0.75                                # score ties (class confidence differs -> not all equal after the product)
    p[2, :50, 6:] = 1.0                                # ... these ARE all equal: same score, same class
    p[3, 5, 0] = float("nan")
    p[3, 6, 2] = 1.0                                   # too small
    a = non_max_suppression(p.clone().to(cuda_dev), 0.4, 0.3)
    qb = p.clone().to(cuda_dev)
    b = non_max_suppression_batched(qb, 0.4, 0.3)
    qa = p.clone().to(cuda_dev)
    non_max_suppression(qa, 0.4, 0.3)
    assert torch.equal(torch.nan_to_num(qa), torch.nan_to_num(qb))
    assert [x is None for x in a] == [x is None for x in b]
    for x, y in zip(a, b):
        if x is not None:
            assert torch.equal(x, y)
    assert sum(len(x) for x in a if x is not None) > 100


def test_segmented_rnms_equals_per_set_rnms(cuda_dev):
    from rotate_yolov3_amd.utils.nms.r_nms import r_nms, r_nms_segmented
    from oracle import riou
    sizes = [1, 64, 65, 700, 0, 130, 2048]
    sets = []
    for k, nn in enumerate(sizes):
        d = torch.from_numpy(riou.random_boxes(max(nn, 1), seed=20 + k, extent=150.0))[:nn]
        d = d[(-d[:, 5]).argsort(stable=True)]
        sets.append(d)
    dets = torch.cat(sets).to(cuda_dev)
    off = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device=cuda_dev)
    flags = r_nms_segmented(dets, off, max(sizes), 0.3).cpu().numpy().astype(bool)
    lo = 0
    for d in sets:
        want = np.zeros(len(d), bool)
        if len(d):
            want[r_nms(d.to(cuda_dev), 0.3).cpu().numpy()] = True
        assert np.array_equal(flags[lo:lo + len(d)], want)
        lo += len(d)


def test_detect_pipeline_end_to_end(cuda_dev):
    # forward + NMS wrapper on the GPU vs the oracle pipeline fed with the SAME decoded predictions
    from rotate_yolov3_amd.utils.nms.nms import non_max_suppression
    cfg = make_cfg.darknet53()
    m, mg = _model(cfg, cuda_dev)
    x = torch.rand(2, 3, 160, 160, generator=torch.Generator().manual_seed(3)).to(cuda_dev)
    with torch.no_grad():
        io, _ = mg(x)
    thr = float(io[..., 5].flatten().kthvalue(int(io[..., 5].numel() * 0.9)).values)
    ref = do.non_max_suppression(io.cpu().clone(), thr, 0.3)
    got = non_max_suppression(io.clone(), thr, 0.3)
    for a, b in zip(got, ref):
        assert (a is None) == (b is None)
        if a is not None:
            assert np.array_equal(a.cpu().numpy(), b.numpy())


@pytest.mark.parametrize("which", ["d53", "tiny"])
def test_engine_detect_fused_decode_filter_equals_forward_plus_nms(cuda_dev, which):
    """HipEngine.detect (decode + confidence filter + compaction in one kernel per head, no `io`; one segmented NMS
    launch) returns exactly non_max_suppression(model(x)[0]) -- single class (Darknet-53) and 80 classes (tiny cfg)."""
    from rotate_yolov3_amd.utils.nms.nms import non_max_suppression
    cfg = make_cfg.darknet53() if which == "d53" else make_cfg.tiny()
    m, mg = _model(cfg, cuda_dev)
    x = torch.rand(3, 3, 160, 160, generator=torch.Generator().manual_seed(5)).to(cuda_dev)
    with torch.no_grad():
        io, _ = mg(x)
    score = io[..., 5] * io[..., 6:].max(2)[0]
    thr = float(score.flatten().kthvalue(int(score.numel() * 0.97)).values)
    want = non_max_suppression(io.clone(), thr, 0.3)
    eng = [e for e in mg._engines.values() if hasattr(e, "detect")][0]
    got = eng.detect(x, thr, 0.3, capacity=64)           # tiny capacity: exercises the grow-and-rerun path
    assert [a is None for a in want] == [b is None for b in got]
    for a, b in zip(want, got):
        if a is not None:
            assert torch.equal(a, b)
    assert sum(len(a) for a in want if a is not None) > 50


def test_eval_entry_point_and_ap(cuda_dev, tmp_path):
    # test.py's loop on synthetic data: runs forward + NMS + rotated-IoU matching + AP; and AP arithmetic on a known case
    import test as test_entry
    from rotate_yolov3_amd.utils.metrics import ap_per_class, match_predictions, skew_bbox_iou
    cfg = make_cfg.darknet53()
    m, mg = _model(cfg, cuda_dev)
    mg.nc = 1
    res, maps = test_entry.test(cfg, {"context_factor": 1.0}, model=mg, batch_size=2, img_size=160, n_images=4,
                                conf_thres=0.9, device=cuda_dev)
    assert len(res) == 7 and 0.0 <= res[2] <= 1.0
    pred = torch.tensor([[50, 50, 40, 10, 0.3, 0.9, 1, 0], [120, 80, 30, 8, -0.5, 0.8, 1, 0], [10, 10, 5, 5, 0, 0.7, 1, 0]],
                        device=cuda_dev)
    labels = torch.tensor([[0, 51, 50, 40, 10, 0.3], [0, 121, 80, 30, 8, -0.5]], device=cuda_dev)
    assert match_predictions(pred, labels, 0.5) == [1, 1, 0]
    iou = skew_bbox_iou(pred[0, :5], labels[:, 1:6])
    assert 0.85 < float(iou[0]) < 1.0 and float(iou[1]) == 0.0
    # two IDENTICAL rotated boxes: the evaluation path (shapely semantics, reference utils.py:663-699) says 1; the native NMS
    # kernel's arithmetic collects 8 coincident vertices, and its angular sort + triangle fan (kernel.cu:35-89/26-33) yields
    # half the area -> 1/3 (oracle and _ref agree) -- kept bit for bit where the reference uses it (r_nms / riou_*)
    from rotate_yolov3_amd.utils.metrics import riou_pairs
    same = skew_bbox_iou(pred[0, :5], pred[:1, :5])
    assert abs(float(same[0]) - 1.0) < 1e-6
    assert abs(float(riou_pairs(pred[:1, :5].contiguous(), pred[:1, :5].contiguous())[0]) - 1.0 / 3.0) < 1e-5
    p, r, ap, f1, cls = ap_per_class(np.array([1, 1, 0]), np.array([.9, .8, .7]), np.zeros(3), np.zeros(2))
    assert abs(ap[0] - 1.0) < 1e-9 and abs(r[0] - 1.0) < 1e-9 and abs(p[0] - 2 / 3) < 1e-9


@pytest.mark.parametrize("nc,cf", [(1, 1.0), (3, 1.25)])
def test_decode_filter_kernel_direct_vs_oracle(cuda_dev, nc, cf):
    """ryolo_yolo_decode_filter called DIRECTLY on a seeded head (NHWC bf16, the engine's layout) against
    oracle.darknet_oracle.decode + the filter half of non_max_suppression (reference models.py:198-221, nms.py:33-48): the same
    surviving rows (image, row index), values to fp32 rounding of exp / sigmoid / atan."""
    import ctypes as C
    from rotate_yolov3_amd import _lib
    from rotate_yolov3_amd.model import engine as _eng  # noqa: F401  (declares the entry point's signature)
    bs, ny, nx, na = 2, 5, 7, 6
    no = nc + 6
    g = torch.Generator().manual_seed(40 + nc)
    head = (torch.randn(bs, na * no, ny, nx, generator=g) * 1.2)
    head[:, 5::no] += 1.0                                   # some objectness above the threshold
    head = head.to(torch.bfloat16)
    anchors = torch.tensor([[30., 10., -0.6], [30., 10., 0.6], [60., 20., -0.6], [60., 20., 0.6], [90., 30., -0.6], [90., 30., 0.6]])
    img = (ny * 16, nx * 16)                                # stride 16 from the longer side: max(img) / max(nx, ny)
    stride = float(max(img)) / float(max(nx, ny))
    io, _ = do.decode(head.float(), anchors.numpy(), img, cf=cf, arc="default", nc=nc)
    thr = 0.55
    cc, cp = io[..., 6:].max(2)
    score = io[..., 5] * cc
    keep = (score > thr) & (io[..., 2:4] > 2.0).all(2) & torch.isfinite(io).all(2)
    safe = (score - thr).abs() > 1e-4                       # rows this close to the threshold may fall either way
    cpad = (na * no + 7) // 8 * 8
    hd = torch.zeros(bs, ny, nx, cpad, dtype=torch.bfloat16)
    hd[..., :na * no] = head.permute(0, 2, 3, 1)
    hd = hd.to(cuda_dev)
    cap = 4096
    cand = torch.zeros(cap, 8, device=cuda_dev)
    rows = torch.zeros(cap, dtype=torch.int64, device=cuda_dev)
    cnt = torch.zeros(1, dtype=torch.int32, device=cuda_dev)
    total = na * ny * nx
    rc = _lib.lib().ryolo_yolo_decode_filter(hd.data_ptr(), hd.stride(2), bs, ny, nx, na, no, anchors.to(cuda_dev).data_ptr(), stride, cf,
                                             0, thr, 2.0, total, 0, cand.data_ptr(), rows.data_ptr(), cnt.data_ptr(), cap,
                                             _lib.stream_ptr(cuda_dev))
    assert rc == 0
    m = int(cnt.item())
    rid, o = rows[:m].sort()
    got = cand[:m][o].cpu()
    rid = rid.cpu()
    want_ids = torch.nonzero(keep.flatten()).flatten()
    sure = set(torch.nonzero((keep & safe).flatten()).flatten().tolist())
    maybe = set(torch.nonzero((~safe).flatten()).flatten().tolist())
    got_ids = set(rid.tolist())
    assert sure <= got_ids and got_ids <= (set(want_ids.tolist()) | maybe) and len(sure) > 20
    flat = io.reshape(-1, no)
    for r, row in zip(rid.tolist(), got):
        ref = flat[r]
        want = torch.cat((ref[:5], (ref[5] * ref[6:].max()).view(1), ref[6:].max().view(1), ref[6:].argmax().float().view(1)))
        assert torch.allclose(row, want, rtol=2e-5, atol=1e-5), (r, row, want)


def test_forward_with_fused_stem_pairs_equals_one_launch_per_layer(cuda_dev, monkeypatch):
    """layers 2-4 of Darknet-53 run as one fused launch in the eval engine (csrc/conv_stem.hip conv_stem_pair_kernel): the
    whole forward is bit-identical to the one-launch-per-layer plan (RYOLO_STEM_PAIR=0), and the plan really contains the pairs"""
    import torch
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.model.models import Darknet
    from tests.procedural import fill_procedural
    cfg = make_cfg.darknet53(160, 160)
    m = fill_procedural(Darknet(cfg, {"context_factor": 1.0}).eval()).to(cuda_dev)
    x = torch.rand(2, 3, 160, 160, generator=torch.Generator().manual_seed(3)).to(cuda_dev)
    with torch.no_grad():
        io_a, p_a = m(x)
        eng = [e for e in m._engines.values() if hasattr(e, "op_info")][0]
        names = [o["name"] for o in eng.op_info]
        assert sum(n.startswith("conv_stem_pair") for n in names) == 1, names[:6]
        monkeypatch.setenv("RYOLO_STEM_PAIR", "0")
        m.refresh_engines()
        io_b, p_b = m(x)
        eng = [e for e in m._engines.values() if hasattr(e, "op_info")][0]
        assert not any(o["name"].startswith("conv_stem_pair") for o in eng.op_info)
    torch.cuda.synchronize()
    assert torch.equal(io_a, io_b)
    for a, b in zip(p_a, p_b):
        assert torch.equal(a, b)


def test_forward_with_fused_head_decode_equals_conv_then_decode(cuda_dev, monkeypatch):
    """every head of Darknet-53 (1x1 1024 / 512 / 256 -> 504, then YOLOLayer.forward) runs as ONE launch in the eval engine (conv_pw.hip MODE 4:
    the head values are rounded to bf16 into an LDS tile and decoded there): io and p are bit-identical to conv + decode
    (RYOLO_HEAD_DECODE=0), and detect() -- which needs the head tensor for its decode + filter kernel -- still returns the same rows"""
    import torch
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.model.models import Darknet
    from tests.procedural import fill_procedural
    cfg = make_cfg.darknet53(160, 192)
    m = fill_procedural(Darknet(cfg, {"context_factor": 1.0}).eval()).to(cuda_dev)
    x = torch.rand(3, 3, 160, 192, generator=torch.Generator().manual_seed(4)).to(cuda_dev)
    with torch.no_grad():
        io_a, p_a = m(x)
        io_a, p_a = io_a.clone(), [t.clone() for t in p_a]
        eng = [e for e in m._engines.values() if hasattr(e, "op_info")][0]
        assert sum(o["name"].endswith("+decode") for o in eng.op_info) == 3, [o["name"] for o in eng.op_info][-8:]
        thr = float(io_a[..., 5].flatten().kthvalue(int(io_a[..., 5].numel() * 0.97)).values)
        det_a = eng.detect(x, thr, 0.4)
        monkeypatch.setenv("RYOLO_HEAD_DECODE", "0")
        m.refresh_engines()
        io_b, p_b = m(x)
        eng = [e for e in m._engines.values() if hasattr(e, "op_info")][0]
        assert not any(o["name"].endswith("+decode") for o in eng.op_info)
        det_b = eng.detect(x, thr, 0.4)
    torch.cuda.synchronize()
    assert torch.equal(io_a, io_b)
    for a, b in zip(p_a, p_b):
        assert torch.equal(a, b)
    for a, b in zip(det_a, det_b):
        assert (a is None) == (b is None) and (a is None or torch.equal(a, b))
