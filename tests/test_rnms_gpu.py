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


def test_rnms_idempotent_and_sorted_properties(ops, cuda_dev):
    d = riou.random_boxes(30000, seed=13, extent=608.0)
    dt = _t(d, cuda_dev)
    keep = ops.r_nms(dt, 0.5)
    k = keep.cpu().numpy()
    assert np.all(np.diff(k) > 0)                           # ascending original indices
    keep2 = ops.r_nms(dt[keep], 0.5).cpu().numpy()          # survivors do not suppress each other
    assert np.array_equal(keep2, np.arange(len(k)))
