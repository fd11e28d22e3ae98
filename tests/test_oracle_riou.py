"""CPU tier: the oracle (oracle/riou_oracle.c) against the golden vectors generated from the reference's own
arithmetic (tests/golden/gen_riou_golden.py) and against the analytic known answers of SURVEY.md section 4."""
import hashlib
import os

import numpy as np
import pytest

import oracle
from oracle import riou

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_known_answer_four_boxes():
    # boxes of utils/nms/nms_wrapper_test.py:35-38; analytic: 8100/11900, octagon 2(sqrt2-1)/(2-2(sqrt2-1))
    z = np.load(os.path.join(G, "riou_known4.npz"))
    m = riou.riou_matrix(z["dets"], z["dets"])
    assert np.array_equal(m, z["iou"])                      # bit-exact vs the reference arithmetic
    assert abs(m[0, 1] - 8100.0 / 11900.0) < 1e-6
    assert abs(m[0, 2] - 2 * (np.sqrt(2) - 1) / (2 - 2 * (np.sqrt(2) - 1))) < 1e-6
    assert m[1, 2] != m[2, 1]                               # fp32 asymmetry: argument order matters
    assert np.array_equal(riou.rnms(z["dets"], 0.1), [0, 3])
    assert np.array_equal(z["keep_thr0p1"], [0, 3])


def test_iou_matrix_vs_reference_arithmetic():
    z = np.load(os.path.join(G, "riou_matrix256.npz"))
    m = riou.riou_matrix(z["boxes"], z["boxes"])
    # only difference: cos/sin definition (a) (correctly rounded vs glibc cosf) -> a few ulp on a few pairs
    assert np.abs(m - z["iou"]).max() < 5e-7
    assert (m != z["iou"]).mean() < 0.02
    assert np.array_equal(m > 0, z["iou"] > 0)


@pytest.mark.parametrize("n", [64, 1000, 8192])
def test_keep_lists_match_reference(n):
    z = np.load(os.path.join(G, "rnms_keep_n%d.npz" % n))
    assert np.array_equal(riou.rnms(z["dets"], float(z["thr"])), z["keep"])


def test_lazy_equals_bitmatrix_scan():
    d = riou.random_boxes(700, seed=21, extent=150.0)
    k1 = riou.rnms(d, 0.3)
    k2, mask = riou.rnms_bitmatrix(d, 0.3, return_mask=True)
    assert np.array_equal(k1, k2)
    assert mask.shape == (700, 11)


def test_keep_50k_golden():
    z = np.load(os.path.join(G, "rnms_keep_n50000.npz"))
    d = riou.random_boxes(int(z["n"]), seed=int(z["seed"]))
    assert hashlib.sha256(d.astype("<f4").tobytes()).hexdigest() == str(z["dets_sha256"])
    keep = riou.rnms(d, float(z["thr"]), nthreads=oracle.host_cores(8))
    assert np.array_equal(keep, z["keep"])
    assert hashlib.sha256(keep.astype("<i8").tobytes()).hexdigest() == str(z["keep_sha256"])


def test_edge_cases():
    assert len(riou.rnms(np.zeros((0, 6), np.float32), 0.5)) == 0
    one = np.array([[10, 10, 4, 4, 0.3, 0.9]], np.float32)
    assert np.array_equal(riou.rnms(one, 0.5), [0])
    # duplicates: identical boxes suppress each other (IoU == 1 > thr); stable order keeps the lower index
    dup = np.repeat(one, 5, axis=0)
    assert np.array_equal(riou.rnms(dup, 0.5), [0])
    # zero-area box: 0/0 = NaN, comparison false, never suppressed and never suppresses (kernel.cu:251-260)
    z = np.array([[10, 10, 0, 0, 0, 0.9], [10, 10, 0, 0, 0, 0.8], [10, 10, 4, 4, 0, 0.7]], np.float32)
    assert np.array_equal(riou.rnms(z, 0.5), [0, 1, 2])
    # strict threshold: IoU == thr does not suppress
    a = np.array([[0, 0, 2, 2, 0, 0.9], [1, 0, 2, 2, 0, 0.8]], np.float32)   # inter 2, union 6 -> 1/3
    iou = riou.riou_pairs(a[:1], a[1:])[0]
    assert np.array_equal(riou.rnms(a, float(iou)), [0, 1])
    assert np.array_equal(riou.rnms(a, float(np.nextafter(iou, np.float32(0)))), [0])
    # row stride > 6 (the [n,8] rows of utils/nms/nms.py:64)
    d = riou.random_boxes(300, seed=3, extent=100.0)
    d8 = np.concatenate([d, np.ones((300, 2), np.float32)], 1)
    assert np.array_equal(riou.rnms(d8, 0.5), riou.rnms(d, 0.5))


def test_sincos_correctly_rounded():
    rng = np.random.default_rng(1)
    for a in rng.uniform(-np.pi, np.pi, 4000).astype(np.float32):
        s, c = riou.sincos(a)
        assert s == np.float32(np.sin(np.float64(a))) and c == np.float32(np.cos(np.float64(a)))
    for a in (0.0, np.pi / 2, -np.pi / 2, np.pi / 4, 1e-30, 3.0e5):
        s, c = riou.sincos(np.float32(a))
        assert s == np.float32(np.sin(np.float64(np.float32(a)))) and c == np.float32(np.cos(np.float64(np.float32(a))))


def test_product_box_generator_equals_the_oracles():
    # bench.py / tools generate their NMS inputs with rotate-yolov3_amd/utils/synthetic.random_boxes (the product must not import
    # oracle/); the golden keep lists were generated from oracle.riou.random_boxes -- the two must stay the same function
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.utils.synthetic import random_boxes
    for n, seed, ext in ((64, 0, 608.0), (1000, 7, 160.0), (5000, 13, 50.0)):
        assert np.array_equal(random_boxes(n, seed=seed, extent=ext), riou.random_boxes(n, seed=seed, extent=ext))
