"""Golden vectors for the EVALUATION path (rotated IoU by the shapely route, mAP matching, AP arithmetic) produced by the
REFERENCE's own functions imported in the build container: utils/utils.py get_rotated_coors (:702-725), skewiou (:663-699),
skew_bbox_iou (:290-320), ap_per_class (:200-261).  Stubs for what this image lacks: cv2.getRotationMatrix2D = OpenCV's
documented formula, shapely.geometry.Polygon = the fp64 convex clip of oracle/poly_iou.py (shapely/GEOS itself is absent, so
parity with GEOS stays unpinned -- DESIGN.md), torch.cuda.FloatTensor = torch.FloatTensor (the function hard-codes it).
Nothing of the reference is written into the repo: only inputs and outputs (.npz).

    python tests/golden/gen_eval_golden.py        (needs /root/reference)

Fixtures:
  skew_iou_ref.npz   boxes1/boxes2 [n,5] fp32 (random pairs around each other + hand-made special cases: identical,
                     contained, touching, disjoint, thin slivers); corners = reference get_rotated_coors of every box (numpy
                     fp64 input); iou = reference skew_bbox_iou(torch fp32 boxes) per pair
  eval_match.npz     the matching loop of test.py:121-151 (restated in this script around the reference's skew_bbox_iou) on
                     3 seeded images: pred [k,8] rows, labels [nl,6] px, correct lists; then the reference's ap_per_class on
                     the concatenated statistics: p, r, ap, f1, classes
"""
import math
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

from oracle import poly_iou  # noqa: E402


def install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None
    cv2.getRotationMatrix2D = lambda center=None, angle=None, scale=None: poly_iou.rotation_matrix_2d(center, angle, scale)
    sys.modules["cv2"] = cv2
    sh = types.ModuleType("shapely")
    shg = types.ModuleType("shapely.geometry")
    shg.Polygon = poly_iou.Polygon
    shg.MultiPoint = poly_iou.MultiPoint
    sh.geometry = shg
    sys.modules["shapely"] = sh
    sys.modules["shapely.geometry"] = shg
    import matplotlib
    matplotlib.use("Agg")
    rn = types.ModuleType("utils.nms.r_nms")
    rn.r_nms = lambda dets, thr: (_ for _ in ()).throw(RuntimeError("not used"))
    sys.modules["utils.nms.r_nms"] = rn
    torch.cuda.FloatTensor = torch.FloatTensor       # skew_bbox_iou hard-codes the CUDA tensor type (utils.py:291)


def special_pairs():
    a = []
    a.append(([50, 50, 40, 10, 0.3], [50, 50, 40, 10, 0.3]))            # identical (the NMS kernel arithmetic gives 1/3 here)
    a.append(([100, 80, 60, 20, -1.1], [100, 80, 60, 20, -1.1]))        # identical
    a.append(([0, 0, 10, 10, 0.0], [0, 0, 10, 10, math.pi / 4]))        # square vs its 45-degree turn: octagon, 0.7071..
    a.append(([0, 0, 20, 10, 0.0], [5, 0, 20, 10, 0.0]))                # shifted by a quarter: 150 / 250
    a.append(([0, 0, 20, 10, 0.0], [0, 0, 10, 4, 0.5]))                 # contained
    a.append(([0, 0, 10, 10, 0.0], [10, 0, 10, 10, 0.0]))               # touching edges: 0
    a.append(([0, 0, 10, 10, 0.0], [30, 30, 10, 10, 0.7]))              # disjoint
    a.append(([10, 10, 100, 1, 0.2], [10, 10, 100, 1, 0.21]))           # thin slivers, 0.01 rad apart
    a.append(([300, 200, 128, 16, 1.5], [301, 200.5, 120, 18, 1.45]))   # ship-shaped near-duplicates
    a.append(([0, 0, 10, 10, 0.0], [0, 0, 10, 0, 0.0]))                 # zero-area box: 0
    return a


def main():
    assert os.path.isdir(REF)
    install_stubs()
    sys.path.insert(0, REF)
    os.chdir(REF)
    from utils import utils as ru

    rng = np.random.RandomState(11)
    n = 160
    b1 = np.stack([rng.uniform(0, 608, n), rng.uniform(0, 608, n), 8 * 16 ** rng.uniform(0, 1, n),
                   8 * 16 ** rng.uniform(0, 1, n), rng.uniform(-math.pi / 2, math.pi / 2, n)], 1)
    # partners: perturbed copies so that most pairs overlap substantially
    b2 = b1.copy()
    b2[:, 0] += rng.normal(0, 0.25, n) * b1[:, 2]
    b2[:, 1] += rng.normal(0, 0.25, n) * b1[:, 3]
    b2[:, 2] *= np.exp(rng.normal(0, 0.2, n))
    b2[:, 3] *= np.exp(rng.normal(0, 0.2, n))
    b2[:, 4] += rng.normal(0, 0.3, n)
    sp = special_pairs()
    b1 = np.concatenate([np.array([p[0] for p in sp], dtype=np.float64), b1]).astype(np.float32)
    b2 = np.concatenate([np.array([p[1] for p in sp], dtype=np.float64), b2]).astype(np.float32)
    corners1 = np.stack([ru.get_rotated_coors(b.astype(np.float64)) for b in b1])
    corners2 = np.stack([ru.get_rotated_coors(b.astype(np.float64)) for b in b2])
    iou = ru.skew_bbox_iou(torch.from_numpy(b1), torch.from_numpy(b2)).numpy()
    # the single-box broadcast form used by test.py:146
    iou_b = ru.skew_bbox_iou([torch.tensor(float(v)) for v in b1[12]], torch.from_numpy(b2[10:40])).numpy()
    np.savez_compressed(os.path.join(OUT, "skew_iou_ref.npz"), boxes1=b1, boxes2=b2, corners1=corners1, corners2=corners2,
                        iou=iou, iou_broadcast=iou_b, n_special=len(sp))
    print("skew_iou_ref: %d pairs, iou range %.4f..%.4f, identical pairs -> %s" % (len(b1), iou.min(), iou.max(), iou[:2]))

    # ---- matching loop (test.py:121-151) + ap_per_class
    iou_thres = 0.5
    stats = []
    per_image = {}
    for img in range(3):
        r = np.random.RandomState(100 + img)
        nl = [5, 3, 0][img]
        labels = np.zeros((nl, 6), dtype=np.float32)
        if nl:
            labels[:, 0] = r.randint(0, 2, nl)                                      # two classes
            labels[:, 1:3] = r.uniform(80, 520, (nl, 2))
            labels[:, 3] = r.uniform(60, 160, nl)
            labels[:, 4] = labels[:, 3] / r.uniform(3, 8, nl)
            labels[:, 5] = r.uniform(-1.5, 1.5, nl)
        k = 14
        pred = np.zeros((k, 8), dtype=np.float32)
        for i in range(k):
            if nl and i < 10:                           # detections near a target, jittered; some with the wrong class
                t = labels[r.randint(0, nl)]
                pred[i, :5] = t[1:6] + r.normal(0, 1, 5) * np.array([6, 6, 8, 3, 0.08]) * r.choice([0.15, 0.4, 1.2])
                pred[i, 7] = t[0] if r.rand() < 0.8 else 1 - t[0]
            else:
                pred[i, :5] = [r.uniform(50, 550), r.uniform(50, 550), r.uniform(40, 120), r.uniform(8, 30), r.uniform(-1.5, 1.5)]
                pred[i, 7] = r.randint(0, 2)
            pred[i, 5] = r.uniform(0.05, 0.99)
            pred[i, 6] = 1.0
        pred = pred[np.argsort(-pred[:, 5])]            # non_max_suppression returns score-descending rows
        predt, lab = torch.from_numpy(pred), torch.from_numpy(labels)
        tcls = lab[:, 0].tolist() if nl else []
        correct = [0] * len(predt)
        if nl:
            detected = []
            tcls_tensor = lab[:, 0]
            tbox = lab[:, 1:6]
            for i, (*pbox, pconf, pcls_conf, pcls) in enumerate(predt):
                if len(detected) == nl:
                    break
                if pcls.item() not in tcls:
                    continue
                m = (pcls == tcls_tensor).nonzero().view(-1)
                iou_i, bi = ru.skew_bbox_iou(pbox, tbox[m]).max(0)
                if iou_i > iou_thres and m[bi] not in detected:
                    correct[i] = 1
                    detected.append(m[bi])
        stats.append((correct, predt[:, 5].cpu(), predt[:, 7].cpu(), tcls))
        per_image["pred%d" % img] = pred
        per_image["labels%d" % img] = labels
        per_image["correct%d" % img] = np.array(correct, dtype=np.int64)
    st = [np.concatenate(x, 0) for x in list(zip(*stats))]
    p, r_, ap, f1, cls = ru.ap_per_class(*st)
    np.savez_compressed(os.path.join(OUT, "eval_match.npz"), iou_thres=iou_thres, p=p, r=r_, ap=ap, f1=f1, classes=cls,
                        **per_image)
    print("eval_match: correct", [per_image["correct%d" % i].tolist() for i in range(3)], "ap", ap)


if __name__ == "__main__":
    main()
