"""oracle/poly_iou.py -- TEST INFRASTRUCTURE (never imported by the product): fp64 restatement of the reference's
*evaluation-path* rotated IoU, i.e. the shapely route, as opposed to the native NMS kernel's fp32 arithmetic that
oracle/riou_oracle.c restates.

Follows /root/reference/utils/utils.py:
  get_rotated_coors  :702-725   box (cx, cy, w, h, a) -> 8 corner coordinates, rotation matrix from
                                cv2.getRotationMatrix2D(angle = -a*180/pi, center = (cx, cy), scale = 1)
  skewiou            :663-699   Polygon(a).convex_hull / Polygon(b).convex_hull, 0 if either area is 0,
                                inter = poly1.intersection(poly2).area, union = area1 + area2 - inter (mode 'iou'),
                                0 if union == 0, else inter / union
  skew_bbox_iou      :290-320   box1 broadcast against box2 [n, 5], per-pair Python loop over the two functions above
called from test.py:146 (mAP matching) and the dead Python NMS branch nms.py:105.

Third-party arithmetic: shapely / GEOS (unpinned, absent from this image) and OpenCV's getRotationMatrix2D.  Restated
from their published definitions: getRotationMatrix2D(center, angle_deg, scale) = [[al, be, (1-al)cx - be*cy],
[-be, al, be*cx + (1-al)cy]], al = scale*cos, be = scale*sin; the intersection of two convex polygons by
Sutherland-Hodgman clipping in fp64 and areas by the shoelace formula -- for two convex quadrilaterals this is the exact
geometry GEOS computes, to fp64 rounding.  Pinning: analytic known answers (tests/test_oracle_poly_iou.py) and the fixture
tests/golden/skew_iou_ref.npz produced by the REFERENCE's own three functions imported in the build container with this
module's `Polygon` standing in for shapely's (tests/golden/gen_eval_golden.py) -- parity with GEOS itself stays unpinned
(no shapely here), stated in DESIGN.md.
"""
import math

import numpy as np


def rotation_matrix_2d(center, angle_deg, scale=1.0):
    """OpenCV's documented getRotationMatrix2D."""
    al = scale * math.cos(math.radians(angle_deg))
    be = scale * math.sin(math.radians(angle_deg))
    cx, cy = float(center[0]), float(center[1])
    return np.array([[al, be, (1 - al) * cx - be * cy], [-be, al, be * cx + (1 - al) * cy]], dtype=np.float64)


def get_rotated_coors(box):
    """utils.py:702-725 in fp64: corners (xmin,ymin) (xmin,ymax) (xmax,ymax) (xmax,ymin) of the axis-aligned box, each mapped
    by the rotation about the centre by -a (degrees: -a*180/pi) -> [x0,y0,x1,y1,x2,y2,x3,y3]."""
    cx, cy, w, h, a = [float(v) for v in box[:5]]
    xmin, xmax, ymin, ymax = cx - w * 0.5, cx + w * 0.5, cy - h * 0.5, cy + h * 0.5
    R = rotation_matrix_2d((cx, cy), -a * 180.0 / math.pi, 1.0)
    out = []
    for tx, ty in ((xmin, ymin), (xmin, ymax), (xmax, ymax), (xmax, ymin)):
        out.append(tx * R[0, 0] + ty * R[0, 1] + R[0, 2])
        out.append(tx * R[1, 0] + ty * R[1, 1] + R[1, 2])
    return np.array(out, dtype=np.float64)


def shoelace(pts):
    """signed area (counter-clockwise positive) of a polygon given as [n, 2]."""
    p = np.asarray(pts, dtype=np.float64)
    if len(p) < 3:
        return 0.0
    x, y = p[:, 0], p[:, 1]
    return 0.5 * float(np.dot(x, np.roll(y, -1)) - np.dot(np.roll(x, -1), y))


def _ccw(pts):
    p = np.asarray(pts, dtype=np.float64).reshape(-1, 2)
    return p if shoelace(p) >= 0 else p[::-1].copy()


def convex_clip(subject, clip):
    """Sutherland-Hodgman: the part of convex polygon `subject` inside convex polygon `clip` (both [n,2]); fp64."""
    out = [tuple(v) for v in _ccw(subject)]
    c = _ccw(clip)
    n = len(c)
    for i in range(n):
        if not out:
            break
        ax, ay = c[i]
        bx, by = c[(i + 1) % n]
        ex, ey = bx - ax, by - ay
        inp, out = out, []
        m = len(inp)
        for j in range(m):
            px, py = inp[j]
            qx, qy = inp[(j + 1) % m]
            sp = ex * (py - ay) - ey * (px - ax)      # > 0: left of the clip edge (inside for a ccw clip polygon)
            sq = ex * (qy - ay) - ey * (qx - ax)
            if sp >= 0:
                out.append((px, py))
            if (sp > 0 and sq < 0) or (sp < 0 and sq > 0):
                t = sp / (sp - sq)
                out.append((px + t * (qx - px), py + t * (qy - py)))
    return np.array(out, dtype=np.float64).reshape(-1, 2)


def skewiou(box1, box2):
    """utils.py:663-699, mode 'iou', on two 8-vectors of corners."""
    a = np.asarray(box1, dtype=np.float64).reshape(4, 2)
    b = np.asarray(box2, dtype=np.float64).reshape(4, 2)
    a1, a2 = abs(shoelace(a)), abs(shoelace(b))
    if a1 == 0 or a2 == 0:
        return 0.0
    inter = abs(shoelace(convex_clip(a, b)))
    union = a1 + a2 - inter
    if union == 0:
        return 0.0
    return inter / union


def skew_bbox_iou(box1, box2):
    """utils.py:290-320: box1 [5] or [n,5], box2 [n,5] -> float64 [n]."""
    b2 = np.asarray(box2, dtype=np.float64).reshape(-1, np.asarray(box2).shape[-1])
    b1 = np.asarray(box1, dtype=np.float64)
    if b1.ndim == 1:
        b1 = np.repeat(b1[None, :], len(b2), 0)
    elif len(b1) != len(b2):
        b1 = np.repeat(b1[:1], len(b2), 0)
    return np.array([skewiou(get_rotated_coors(b1[i]), get_rotated_coors(b2[i])) for i in range(len(b2))], dtype=np.float64)


def skew_iou_matrix(b1, b2):
    c1 = [get_rotated_coors(b) for b in np.asarray(b1, dtype=np.float64)]
    c2 = [get_rotated_coors(b) for b in np.asarray(b2, dtype=np.float64)]
    return np.array([[skewiou(p, q) for q in c2] for p in c1], dtype=np.float64).reshape(len(c1), len(c2))


# ---- stand-ins with shapely's call surface, for running the REFERENCE's skewiou in the build container
class Polygon(object):
    """The subset of shapely.geometry.Polygon that utils.py:663-699 touches, for CONVEX inputs (rectangles and their
    intersections): convex_hull, is_valid, area, intersection."""

    def __init__(self, pts):
        if isinstance(pts, Polygon):
            pts = pts.pts
        self.pts = np.array([[float(v) for v in p] for p in pts], dtype=np.float64).reshape(-1, 2)

    @property
    def convex_hull(self):
        return Polygon(_ccw(self.pts))

    @property
    def is_valid(self):
        return True

    @property
    def area(self):
        return abs(shoelace(self.pts))

    def intersection(self, other):
        return Polygon(convex_clip(self.pts, other.pts))


class MultiPoint(object):      # only reached by modes the reference's callers never use
    def __init__(self, pts):
        raise NotImplementedError("skewiou modes other than 'iou' are unreachable from the hot path")
