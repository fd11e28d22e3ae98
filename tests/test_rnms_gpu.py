"""GPU tier: HIP rotated IoU / NMS (through the C ABI) vs the oracle -- bit-exact."""
import hashlib
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import riou

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ops(cuda_dev):
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.utils.nms import r_nms as m
    return m


def _t(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def test_known_answer(ops, cuda_dev):
    z = np.load(os.path.join(G, "riou_known4.npz"))
    m = ops.riou_matrix(_t(z["dets"], cuda_dev), _t(z["dets"], cuda_dev)).cpu().numpy()
    assert np.array_equal(m, riou.riou_matrix(z["dets"], z["dets"]))
    assert np.array_equal(m, z["iou"])
    keep = ops.r_nms(_t(z["dets"], cuda_dev), 0.1)
    assert keep.dtype == torch.int64 and keep.is_cuda
    assert keep.cpu().tolist() == [0, 3]


def test_iou_matrix_bit_exact(ops, cuda_dev):
    b = riou.random_boxes(512, seed=31, extent=140.0)
    got = ops.riou_matrix(_t(b, cuda_dev), _t(b, cuda_dev)).cpu().numpy()
    want = riou.riou_matrix(b, b)
    assert (want > 0).sum() > 20000
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_iou_pairs_near_coincident_bit_exact(ops, cuda_dev):
    # near-identical pairs: many candidate points, sometimes more than 8 (definition (b) buffers)
    rng = np.random.default_rng(5)
    a = riou.random_boxes(20000, seed=32, extent=50.0)
    b = a.copy()
    b[:, :5] += rng.normal(0, 1e-4, (20000, 5)).astype(np.float32) * np.abs(a[:, :5])
    b[::7] = a[::7]
    got = ops.riou_pairs(_t(a, cuda_dev), _t(b, cuda_dev)).cpu().numpy()
    want = riou.riou_pairs(a, b)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


@pytest.mark.parametrize("n,seed,extent,thr", [(1, 1, 50, 0.5), (63, 2, 60, 0.5), (64, 3, 60, 0.3), (65, 4, 60, 0.5),
                                               (1000, 5, 300, 0.5), (4097, 6, 400, 0.4), (20000, 7, 608, 0.5)])
def test_rnms_bit_exact_vs_oracle(ops, cuda_dev, n, seed, extent, thr):
    d = riou.random_boxes(n, seed=seed, extent=float(extent))
    got = ops.r_nms(_t(d, cuda_dev), thr).cpu().numpy()
    want = riou.rnms(d, thr, nthreads=oracle.host_cores(8))
    assert np.array_equal(got, want)


@pytest.mark.parametrize("n", [64, 1000, 8192])
def test_rnms_golden_from_reference(ops, cuda_dev, n):
    z = np.load(os.path.join(G, "rnms_keep_n%d.npz" % n))
    got = ops.r_nms(_t(z["dets"], cuda_dev), float(z["thr"])).cpu().numpy()
    assert np.array_equal(got, z["keep"])


def test_rnms_50k_full_size_golden(ops, cuda_dev):
    z = np.load(os.path.join(G, "rnms_keep_n50000.npz"))
    d = riou.random_boxes(int(z["n"]), seed=int(z["seed"]))
    got = ops.r_nms(_t(d, cuda_dev), float(z["thr"])).cpu().numpy()
    assert len(got) == len(z["keep"])
    assert hashlib.sha256(got.astype("<i8").tobytes()).hexdigest() == str(z["keep_sha256"])


def test_rnms_edge_cases(ops, cuda_dev):
    out = ops.r_nms(torch.zeros(0, 6, device=cuda_dev), 0.5)
    assert out.numel() == 0 and out.dtype == torch.int64 and not out.is_cuda      # rotate_polygon_nms.cpp:9-10
    with pytest.raises(RuntimeError):
        ops.r_nms(torch.zeros(3, 6), 0.5)                                        # AT_CHECK, cpp:3,8
    one = np.array([[10, 10, 4, 4, 0.3, 0.9]], np.float32)
    dup = np.repeat(one, 130, axis=0)
    assert ops.r_nms(_t(dup, cuda_dev), 0.5).cpu().tolist() == [0]
    z = np.array([[10, 10, 0, 0, 0, 0.9], [10, 10, 0, 0, 0, 0.8], [10, 10, 4, 4, 0, 0.7]], np.float32)
    assert ops.r_nms(_t(z, cuda_dev), 0.5).cpu().tolist() == [0, 1, 2]
    # column slice of an [n,8] tensor, as utils/nms/nms.py:64 passes it
    d = riou.random_boxes(3000, seed=9, extent=200.0)
    d8 = torch.cat([_t(d, cuda_dev), torch.ones(3000, 2, device=cuda_dev)], 1)
    got = ops.r_nms(d8[:, :6], 0.5).cpu().numpy()
    assert np.array_equal(got, riou.rnms(d, 0.5))
    # ties in score: stable order (lower index first)
    t = riou.random_boxes(2000, seed=10, extent=150.0)
    t[:, 5] = np.round(t[:, 5] * 8) / 8
    assert np.array_equal(ops.r_nms(_t(t, cuda_dev), 0.5).cpu().numpy(), riou.rnms(t, 0.5))
    # all boxes overlapping heavily: dense masks, long suppression chains
    c = riou.random_boxes(5000, seed=11, extent=12.0)
    assert np.array_equal(ops.r_nms(_t(c, cuda_dev), 0.5).cpu().numpy(), riou.rnms(c, 0.5, nthreads=oracle.host_cores(8)))
    # NaN / inf rows must behave like the oracle (never suppress, never suppressed unless IoU is defined)
    w = riou.random_boxes(500, seed=12, extent=80.0)
    w[5, 0] = np.nan
    w[17, 2] = np.inf
    w[40, 4] = np.nan
    assert np.array_equal(ops.r_nms(_t(w, cuda_dev), 0.5).cpu().numpy(), riou.rnms(w, 0.5))


def test_rnms_reject_premise_cases(ops, cuda_dev):
    """The bounding-circle reject must only skip pairs the reference arithmetic scores 0 (csrc/rnms.hip header).  Inputs that
    break its premise, each against the oracle: zero-width / zero-size boxes (the reference's in_rect reports every point of
    the plane inside them and the IoU becomes area/0 = inf: they suppress boxes ANYWHERE -- found in r2, the r1 kernel
    rejected those pairs), needle boxes thinner than the guard, a negative threshold (IoU == 0 suppresses), and -- inside the
    premise -- rows of equal rotated boxes whose edges are collinear to rounding noise."""
    d = np.array([[10, 10, 0, 0, 0, 0.9], [500, 500, 4, 4, 0, 0.8], [10, 10, 0, 8, 0, 0.7], [500, 10, 4, 4, 0, 0.6]], np.float32)
    want = riou.rnms(d, 0.5)
    assert want.tolist() == [0, 2]                        # the two zero-area boxes suppress the far 4x4 boxes
    assert np.array_equal(ops.r_nms(_t(d, cuda_dev), 0.5).cpu().numpy(), want)
    rng = np.random.default_rng(3)
    base = riou.random_boxes(4000, seed=31, extent=500.0)
    z = base.copy()
    idx = rng.choice(4000, 60, replace=False)
    z[idx[:20], 2] = 0.0                                  # zero width
    z[idx[20:40], 3] = 0.0                                # zero height
    z[idx[40:50], 2:4] = 0.0                              # points
    z[idx[50:], 3] = rng.uniform(1e-6, 1e-2, 10).astype(np.float32)   # needles far thinner than the guard
    for thr in (0.5, 0.1):
        assert np.array_equal(ops.r_nms(_t(z, cuda_dev), thr).cpu().numpy(), riou.rnms(z, thr, nthreads=oracle.host_cores(8)))
    # negative threshold: every pair with a defined IoU suppresses
    small = riou.random_boxes(700, seed=32, extent=300.0)
    assert np.array_equal(ops.r_nms(_t(small, cuda_dev), -0.1).cpu().numpy(), riou.rnms(small, -0.1))
    # rows of identical rotated boxes, edge lines collinear up to the rounding of the corner computation
    rows = []
    for k, ang in enumerate((0.3, -0.7, 1.1, 0.0, 1.5707964)):
        c, s_ = np.cos(ang), np.sin(ang)
        for i in range(120):
            t = 45.0 * i                                  # along the box axis: 40-long boxes, 5 apart
            for lat in (0.0, 14.0):                       # and a second row sharing the short-edge lines
                rows.append([50 + 30 * k + t * c - lat * s_, 80 + 11 * k + t * s_ + lat * c, 40.0, 12.0, ang, rng.uniform()])
    al = np.array(rows, np.float32)
    for thr in (0.0, 0.3):
        assert np.array_equal(ops.r_nms(_t(al, cuda_dev), thr).cpu().numpy(), riou.rnms(al, thr, nthreads=oracle.host_cores(8)))
    # the same through the segmented entry point (score-sorted sets)
    zs = z[np.argsort(-z[:, 5], kind="stable")]
    off = torch.tensor([0, 1500, 4000], dtype=torch.int32, device=cuda_dev)
    flags = ops.r_nms_segmented(_t(zs, cuda_dev), off, 2500, 0.5).cpu().numpy().astype(bool)
    for lo, hi in ((0, 1500), (1500, 4000)):
        keep = riou.rnms(zs[lo:hi], 0.5, nthreads=oracle.host_cores(8))
        assert np.array_equal(np.nonzero(flags[lo:hi])[0], np.sort(keep))


def test_rnms_clusters_and_long_rows(ops, cuda_dev):
    """the scan kernel's three tile forms in one call -- empty, listed (<= 7 suppressing pairs in the 16-byte summary) and
    dense (64 column words): tight clusters of near-duplicates scattered over a sparse background; and a call long enough
    (> 61 440 boxes) that the per-step (row, column) tile count exceeds what the appliers prefetch."""
    rng = np.random.default_rng(5)
    bg = riou.random_boxes(6000, seed=21, extent=900.0)
    centers = bg[rng.choice(6000, 40, replace=False)]
    cl = np.repeat(centers, 90, axis=0)
    cl[:, 0:2] += rng.normal(0, 1.0, (len(cl), 2)).astype(np.float32)
    cl[:, 2:4] *= np.exp(rng.normal(0, 0.05, (len(cl), 2))).astype(np.float32)
    cl[:, 4] += rng.normal(0, 0.03, len(cl)).astype(np.float32)
    cl[:, 5] = rng.uniform(0, 1, len(cl)).astype(np.float32)
    d = np.concatenate([bg, cl], 0).astype(np.float32)
    d = d[rng.permutation(len(d))]
    for thr in (0.3, 0.6):
        assert np.array_equal(ops.r_nms(_t(d, cuda_dev), thr).cpu().numpy(), riou.rnms(d, thr, nthreads=oracle.host_cores(8)))
    # the same clusters with scores ordered by cluster: near-duplicates adjacent in the sorted order -> dense DIAGONAL tiles
    d2 = d.copy()
    d2[:, 5] = np.argsort(np.argsort(np.round(d2[:, 0] / 20) * 1000 + np.round(d2[:, 1] / 20))).astype(np.float32) / len(d2)
    assert np.array_equal(ops.r_nms(_t(d2, cuda_dev), 0.5).cpu().numpy(), riou.rnms(d2, 0.5, nthreads=oracle.host_cores(8)))
    big = riou.random_boxes(70000, seed=22, extent=1400.0)
    assert np.array_equal(ops.r_nms(_t(big, cuda_dev), 0.5).cpu().numpy(), riou.rnms(big, 0.5, nthreads=oracle.host_cores(8)))


def test_rnms_split_scan_around_its_threshold(ops, cuda_dev):
    """Round 6: from 320 block rows (20 417 boxes) ryolo_rnms runs the block rows in two mask launches and the first 60 % of the scan's
    panel steps beside the second one (a second stream, events).  The last unsplit size, the first split one and a larger one, on boxes
    clustered so that kept boxes of the first part suppress boxes of the second (the state handed from the first scan launch to the
    second matters): the oracle's keep lists; and the same call ten times (a race between the pieces would show as a differing list)."""
    rng = np.random.default_rng(9)
    for n in (20416, 20417, 26000):
        bg = riou.random_boxes(n - 3000, seed=100 + n % 7, extent=700.0)
        centers = bg[rng.choice(len(bg), 30, replace=False)]
        cl = np.repeat(centers, 100, axis=0)
        cl[:, 0:2] += rng.normal(0, 1.5, (len(cl), 2)).astype(np.float32)
        cl[:, 5] = rng.uniform(0, 1, len(cl)).astype(np.float32)          # cluster members spread over the whole score order
        d = np.concatenate([bg, cl], 0).astype(np.float32)
        d = d[rng.permutation(len(d))]
        dt = _t(d, cuda_dev)
        want = riou.rnms(d, 0.45, nthreads=oracle.host_cores(8))
        first = ops.r_nms(dt, 0.45).cpu().numpy()
        assert np.array_equal(first, want), n
        for _ in range(9):
            assert np.array_equal(ops.r_nms(dt, 0.45).cpu().numpy(), first), n


def test_rnms_idempotent_and_sorted_properties(ops, cuda_dev):
    d = riou.random_boxes(30000, seed=13, extent=608.0)
    dt = _t(d, cuda_dev)
    keep = ops.r_nms(dt, 0.5)
    k = keep.cpu().numpy()
    assert np.all(np.diff(k) > 0)                           # ascending original indices
    keep2 = ops.r_nms(dt[keep], 0.5).cpu().numpy()          # survivors do not suppress each other
    assert np.array_equal(keep2, np.arange(len(k)))


@pytest.mark.parametrize("tiles", ["1", "2"])
def test_both_mask_kernels_on_every_edge_case(ops, cuda_dev, tiles):
    """Round 6: the mask kernel takes one column tile per wave below 128 block rows and two from there (8 128 boxes); the two
    instantiations walk their tiles, rings and summaries differently.  RYOLO_RNMS_TILES forces one of them for any size: every edge case
    of this file -- a single tile, ragged last tiles, dense clusters, NaN / inf rows, score ties, a negative threshold, the segmented call
    -- through both; the keep lists must be the oracle's (and therefore each other's)."""
    from rotate_yolov3_amd import _lib
    from rotate_yolov3_amd.utils.nms.r_nms import r_nms_segmented
    _lib.set_tuning("RYOLO_RNMS_TILES", tiles)
    try:
        for n, seed, extent, thr in [(4, 1, 30.0, 0.5), (64, 2, 50.0, 0.3), (65, 3, 50.0, 0.5), (129, 4, 60.0, 0.5), (1000, 5, 300, 0.5),
                                     (4097, 6, 400, 0.4), (8192, 8, 608, 0.5)]:
            d = riou.random_boxes(n, seed=seed, extent=extent)
            assert np.array_equal(ops.r_nms(_t(d, cuda_dev), thr).cpu().numpy(), riou.rnms(d, thr, nthreads=oracle.host_cores(8))), n
        dup = np.repeat(np.array([[10, 10, 4, 4, 0.3, 0.9]], np.float32), 130, axis=0)
        assert ops.r_nms(_t(dup, cuda_dev), 0.5).cpu().tolist() == [0]
        t = riou.random_boxes(2000, seed=10, extent=150.0)
        t[:, 5] = np.round(t[:, 5] * 8) / 8                                  # ties
        assert np.array_equal(ops.r_nms(_t(t, cuda_dev), 0.5).cpu().numpy(), riou.rnms(t, 0.5))
        c = riou.random_boxes(5000, seed=11, extent=12.0)                    # dense: every tile stored as 64 column words
        assert np.array_equal(ops.r_nms(_t(c, cuda_dev), 0.5).cpu().numpy(), riou.rnms(c, 0.5, nthreads=oracle.host_cores(8)))
        w = riou.random_boxes(500, seed=12, extent=80.0)
        w[5, 0] = np.nan
        w[17, 2] = np.inf
        w[40, 4] = np.nan
        assert np.array_equal(ops.r_nms(_t(w, cuda_dev), 0.5).cpu().numpy(), riou.rnms(w, 0.5))
        assert np.array_equal(ops.r_nms(_t(w, cuda_dev), -1.0).cpu().numpy(), riou.rnms(w, -1.0))      # negative threshold: no reject stage
        # segmented call: sets of ragged sizes (one of them a single box, one of 200 boxes: four tiles, odd tile count per row)
        sizes = [1, 200, 64, 333, 65]
        sets = [riou.random_boxes(m, seed=40 + k, extent=60.0) for k, m in enumerate(sizes)]
        sets = [b[np.argsort(-b[:, 5], kind="stable")] for b in sets]
        dets = _t(np.concatenate(sets), cuda_dev)
        off = torch.tensor(np.concatenate([[0], np.cumsum(sizes)]), dtype=torch.int32, device=cuda_dev)
        flags = r_nms_segmented(dets, off, max(sizes), 0.3).cpu().numpy().astype(bool)
        lo = 0
        for b in sets:
            want = np.zeros(len(b), bool)
            want[riou.rnms(b, 0.3)] = True
            assert np.array_equal(flags[lo:lo + len(b)], want)
            lo += len(b)
    finally:
        _lib.set_tuning("RYOLO_RNMS_TILES", None)
