"""CPU tier: the fp64 polygon-IoU oracle (oracle/poly_iou.py, the evaluation path's skew_bbox_iou semantics) against analytic
known answers and against the fixture produced by the reference's own get_rotated_coors / skewiou / skew_bbox_iou
(tests/golden/gen_eval_golden.py); the product's get_rotated_coors mirror and AP arithmetic against the same fixtures."""
import math
import os

import numpy as np
import torch

from oracle import poly_iou as pi

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_known_answers():
    sq = [0, 0, 10, 10, 0.0]
    assert pi.skew_bbox_iou(sq, [sq])[0] == 1.0
    # a square and its 45-degree turn: the intersection is a regular octagon, area 2(sqrt2 - 1) s^2
    o = pi.skew_bbox_iou(sq, [[0, 0, 10, 10, math.pi / 4]])[0]
    inter = 2 * (math.sqrt(2) - 1) * 100
    assert abs(o - inter / (200 - inter)) < 1e-12
    assert abs(pi.skew_bbox_iou([0, 0, 20, 10, 0], [[5, 0, 20, 10, 0]])[0] - 150.0 / 250.0) < 1e-12
    assert abs(pi.skew_bbox_iou([0, 0, 20, 10, 0], [[0, 0, 10, 4, 0.5]])[0] - 40.0 / 200.0) < 1e-12     # contained
    assert pi.skew_bbox_iou(sq, [[10, 0, 10, 10, 0]])[0] == 0.0                                           # touching
    assert pi.skew_bbox_iou(sq, [[30, 30, 10, 10, 0.7]])[0] == 0.0
    assert pi.skew_bbox_iou(sq, [[0, 0, 10, 0, 0]])[0] == 0.0                                             # zero area
    # symmetry and rotation invariance
    a, b = [50, 40, 80, 12, 0.4], [55, 43, 70, 15, 0.1]
    assert abs(pi.skew_bbox_iou(a, [b])[0] - pi.skew_bbox_iou(b, [a])[0]) < 1e-12
    a2, b2 = list(a), list(b)
    a2[4] += math.pi
    assert abs(pi.skew_bbox_iou(a2, [b])[0] - pi.skew_bbox_iou(a, [b])[0]) < 1e-9          # a box equals its half turn
    # the nms_wrapper_test numbers of the reference (utils/nms/nms_wrapper_test.py:35-38): 8100 / 11900
    assert abs(pi.skew_bbox_iou([50, 50, 100, 100, 0], [[60, 60, 100, 100, 0]])[0] - 8100.0 / 11900.0) < 1e-12


def test_get_rotated_coors_order_and_values():
    c = pi.get_rotated_coors([10, 20, 8, 4, 0.0])
    assert np.allclose(c, [6, 18, 6, 22, 14, 22, 14, 18])                # (xmin,ymin) (xmin,ymax) (xmax,ymax) (xmax,ymin)
    c = pi.get_rotated_coors([0, 0, 2, 2, math.pi / 2])                  # rotation by +a (x cos a - y sin a, x sin a + y cos a),
    assert np.allclose(c[:2], [1, -1], atol=1e-12)                      # the convention of convert_region (kernel.cu:196-229): (-1,-1) -> (1,-1)


def test_oracle_and_product_mirror_match_the_reference_fixture():
    z = np.load(os.path.join(G, "skew_iou_ref.npz"))
    b1, b2 = z["boxes1"], z["boxes2"]
    for b, want in ((b1, z["corners1"]), (b2, z["corners2"])):
        got = np.stack([pi.get_rotated_coors(x.astype(np.float64)) for x in b])
        assert np.allclose(got, want, rtol=0, atol=1e-9)
    # the reference's skew_bbox_iou computes the corners from fp32 torch scalars (rounding ~1e-7 relative) and returns fp32
    got = pi.skew_bbox_iou(b1, b2)
    assert np.abs(got - z["iou"]).max() < 2e-5, np.abs(got - z["iou"]).max()
    assert abs(z["iou"][0] - 1.0) < 1e-6 and abs(z["iou"][1] - 1.0) < 1e-6            # identical boxes: 1, not 1/3
    gb = pi.skew_bbox_iou(b1[12], b2[10:40])
    assert np.abs(gb - z["iou_broadcast"]).max() < 2e-5
    # product mirror of get_rotated_coors (numpy and torch inputs)
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.utils.metrics import get_rotated_coors
    for i in (0, 5, 17, 60, 169):
        assert np.allclose(get_rotated_coors(b1[i].astype(np.float64)), z["corners1"][i], atol=1e-9)
        t = get_rotated_coors(torch.from_numpy(b1[i]))
        assert isinstance(t, torch.Tensor) and t.dtype == torch.float32 and t.shape == (8,)
        assert np.allclose(t.numpy(), z["corners1"][i], rtol=1e-5, atol=1e-3)


def test_ap_per_class_matches_the_reference_fixture():
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.utils.metrics import ap_per_class
    z = np.load(os.path.join(G, "eval_match.npz"))
    tp = np.concatenate([z["correct%d" % i] for i in range(3)])
    conf = np.concatenate([z["pred%d" % i][:, 5] for i in range(3)])
    pcls = np.concatenate([z["pred%d" % i][:, 7] for i in range(3)])
    tcls = np.concatenate([z["labels%d" % i][:, 0] for i in range(3)])
    p, r, ap, f1, cls = ap_per_class(tp, conf, pcls, tcls)
    for a, b in ((p, z["p"]), (r, z["r"]), (ap, z["ap"]), (f1, z["f1"])):
        assert np.allclose(a, b, rtol=1e-12, atol=1e-15)
    assert np.array_equal(cls, z["classes"])


def test_clip_against_an_independent_implementation_qhull():
    """the fixtures of the evaluation IoU ran the reference's functions with THIS oracle's Polygon standing in for
    shapely's (GEOS is not in the image), so the clip itself was pinned only by analytic cases.  Here the intersection area of the
    two rectangles is computed a second, independent way -- the feasible region of the eight edge half-planes via Qhull
    (scipy.spatial.HalfspaceIntersection around the Chebyshev centre from scipy.optimize.linprog, area from ConvexHull) -- on 400
    random pairs incl. near-touching, contained and thin boxes."""
    from tests.box_pairs import qhull_iou
    rng = np.random.default_rng(11)

    worst = 0.0
    n_pos = 0
    for t in range(400):
        b1 = np.array([rng.uniform(20, 80), rng.uniform(20, 80), rng.uniform(2, 60), rng.uniform(2, 60), rng.uniform(-3.2, 3.2)])
        kind = t % 4
        if kind == 0:      # overlapping neighbour
            b2 = b1 + np.array([rng.uniform(-15, 15), rng.uniform(-15, 15), rng.uniform(-1, 10), rng.uniform(-1, 10), rng.uniform(-1, 1)])
        elif kind == 1:    # independent box
            b2 = np.array([rng.uniform(20, 80), rng.uniform(20, 80), rng.uniform(2, 60), rng.uniform(2, 60), rng.uniform(-3.2, 3.2)])
        elif kind == 2:    # small box inside / near an edge
            b2 = np.array([b1[0] + rng.uniform(-0.5, 0.5) * b1[2], b1[1] + rng.uniform(-0.5, 0.5) * b1[3], rng.uniform(0.5, 5), rng.uniform(0.5, 5),
                           rng.uniform(-3.2, 3.2)])
        else:              # thin 9:1 boxes (HRSC ships)
            b1[2], b1[3] = rng.uniform(40, 90), rng.uniform(4, 10)
            b2 = b1 + np.array([rng.uniform(-6, 6), rng.uniform(-6, 6), rng.uniform(-5, 5), rng.uniform(-1, 1), rng.uniform(-0.3, 0.3)])
        b2[2:4] = np.abs(b2[2:4]) + 0.5
        want = qhull_iou(b1, b2, pi.get_rotated_coors, pi.shoelace)
        got = float(pi.skew_bbox_iou(b1, [b2])[0])
        worst = max(worst, abs(got - want))
        n_pos += want > 0.05
        assert abs(got - want) < 1e-7, (t, b1, b2, got, want)
    assert n_pos > 150, n_pos
