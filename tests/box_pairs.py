"""Shared test data / independent checker for the evaluation-path rotated IoU (tests/test_oracle_poly_iou.py, tests/test_skewiou_gpu.py)."""
import math

import numpy as np


def make_pairs(n_per=2000, seed=21):
    """Six families of rotated-box pairs (cx, cy, w, h, angle), n_per each, fp32: 0 overlapping neighbours, 1 independent boxes, 2 one box
    contained in the other, 3 thin 9:1 boxes (HRSC ships) a few pixels / degrees apart, 4 near-touching (shifted along the box's own
    axis by its width -/+ half a pixel), 5 near-parallel (angles 1e-4 .. 1e-2 rad apart, offset across the short axis).  Continuous
    random parameters: no exactly coincident or collinear edges."""
    rng = np.random.default_rng(seed)
    b1s, b2s, kinds = [], [], []
    for kind in range(6):
        for _ in range(n_per):
            b1 = np.array([rng.uniform(50, 550), rng.uniform(50, 550), rng.uniform(8, 120), rng.uniform(8, 120),
                           rng.uniform(-math.pi / 2, math.pi / 2)])
            if kind == 0:
                b2 = b1 + np.array([rng.uniform(-25, 25), rng.uniform(-25, 25), rng.uniform(-4, 20), rng.uniform(-4, 20), rng.uniform(-1, 1)])
            elif kind == 1:
                b2 = np.array([b1[0] + rng.uniform(-120, 120), b1[1] + rng.uniform(-120, 120), rng.uniform(8, 120), rng.uniform(8, 120),
                               rng.uniform(-math.pi / 2, math.pi / 2)])
            elif kind == 2:
                s = rng.uniform(0.1, 0.45)
                b2 = np.array([b1[0] + rng.uniform(-0.1, 0.1) * b1[2], b1[1] + rng.uniform(-0.1, 0.1) * b1[3], s * min(b1[2], b1[3]),
                               s * min(b1[2], b1[3]) * rng.uniform(0.5, 1.0), rng.uniform(-math.pi / 2, math.pi / 2)])
                if rng.uniform() < 0.5:
                    b1, b2 = b2, b1
            elif kind == 3:
                b1[2], b1[3] = rng.uniform(60, 180), rng.uniform(7, 20)
                b2 = b1 + np.array([rng.uniform(-8, 8), rng.uniform(-8, 8), rng.uniform(-10, 10), rng.uniform(-2, 2), rng.uniform(-0.3, 0.3)])
            elif kind == 4:
                d = b1[2] + rng.uniform(-0.5, 0.5)
                b2 = b1.copy()
                b2[0] += d * math.cos(b1[4])
                b2[1] += d * math.sin(b1[4])
                b2[4] += rng.uniform(-0.02, 0.02)
                b2[3] *= rng.uniform(0.6, 1.4)
            else:
                b2 = b1.copy()
                b2[4] += rng.choice([-1, 1]) * 10 ** rng.uniform(-4, -2)
                off = rng.uniform(0.05, 0.6) * b1[3]
                b2[0] += -off * math.sin(b1[4])
                b2[1] += off * math.cos(b1[4])
                b2[2] *= rng.uniform(0.7, 1.3)
            b2[2:4] = np.abs(b2[2:4]) + 0.5
            b1s.append(b1)
            b2s.append(b2)
            kinds.append(kind)
    return np.array(b1s, dtype=np.float32), np.array(b2s, dtype=np.float32), np.array(kinds)


def qhull_iou(b1, b2, get_rotated_coors, shoelace):
    """IoU of two rotated boxes with the intersection computed INDEPENDENTLY of the oracle's clip: the feasible region of the eight edge
    half-planes via Qhull (scipy.spatial.HalfspaceIntersection around the Chebyshev centre from scipy.optimize.linprog, area from
    ConvexHull)."""
    from scipy.optimize import linprog
    from scipy.spatial import ConvexHull, HalfspaceIntersection

    def halfplanes(c8):
        p = np.asarray(c8, dtype=np.float64).reshape(4, 2)
        if shoelace(p) < 0:
            p = p[::-1]
        hs = []
        for i in range(4):
            a, b = p[i], p[(i + 1) % 4]
            e = b - a
            n = np.array([e[1], -e[0]])                  # outward normal of a counter-clockwise polygon
            n = n / np.linalg.norm(n)
            hs.append([n[0], n[1], -float(n @ a)])       # n.x + off <= 0 inside
        return np.array(hs)

    b1 = np.asarray(b1, dtype=np.float64)
    b2 = np.asarray(b2, dtype=np.float64)
    hs = np.vstack([halfplanes(get_rotated_coors(b1)), halfplanes(get_rotated_coors(b2))])
    # Chebyshev centre: max r s.t. n.x + r <= -off
    res = linprog([0, 0, -1], A_ub=np.hstack([hs[:, :2], np.ones((8, 1))]), b_ub=-hs[:, 2], bounds=[(None, None), (None, None), (0, None)])
    inter = 0.0
    if res.success and res.x[2] > 1e-9:
        inter = float(ConvexHull(HalfspaceIntersection(hs, res.x[:2]).intersections).volume)
    return inter / (b1[2] * b1[3] + b2[2] * b2[3] - inter)
