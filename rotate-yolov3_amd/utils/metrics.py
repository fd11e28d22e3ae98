"""Evaluation helpers -- mirrors of the reference's utils/utils.py: get_rotated_coors (:702-725), skew_bbox_iou (:290-320,
here one HIP launch instead of a Python + shapely loop per pair), ap_per_class (:200-261), compute_ap (:264-286),
scale_coords (:181-189), and the per-image greedy matching of test.py:114-151 expressed on the rotated-IoU matrix.

Two rotated IoUs exist in the reference and both are kept, each where the reference uses it:
  * the NATIVE NMS KERNEL's fp32 arithmetic (rotate_polygon_nms_kernel.cu:22-260) -> `r_nms`, `riou_pairs`, `riou_matrix`,
    bit-exact incl. its quirks (IoU(A, A) can come out as 1/3 for coincident boxes; harmless for suppression order);
  * the EVALUATION path's exact polygon IoU (shapely, fp64) -> `skew_bbox_iou`, `match_predictions`: `ryolo_skew_iou_*`,
    an fp64 convex clip (csrc/skewiou.hip), IoU(A, A) = 1, within 1e-6 of the fp64 polygon-clip checker used by the tests.
"""
import math

import numpy as np
import torch

from .nms.r_nms import riou_matrix, riou_pairs, skew_iou_matrix, skew_iou_pairs  # noqa: F401


def get_rotated_coors(box):
    """utils.py:702-725: (cx, cy, w, h, a) -> [x0,y0, x1,y1, x2,y2, x3,y3], the corners (xmin,ymin) (xmin,ymax) (xmax,ymax)
    (xmax,ymin) of the axis-aligned box rotated about its centre with cv2.getRotationMatrix2D(angle=-a*180/pi) -- OpenCV's
    documented matrix [[al, be, (1-al)cx - be*cy], [-be, al, be*cx + (1-al)cy]], al = cos, be = sin of that angle.
    A torch box gives a torch tensor (same dtype / device), anything else a float64 numpy array, like the reference."""
    assert len(box) > 0, 'Input valid box!'
    is_t = isinstance(box, torch.Tensor) or (len(box) and isinstance(box[0], torch.Tensor))
    v = [float(b) for b in box[:5]]
    cx, cy, w, h, a = v
    ang = math.radians(-a * 180 / math.pi)
    al, be = math.cos(ang), math.sin(ang)
    r02, r12 = (1 - al) * cx - be * cy, be * cx + (1 - al) * cy
    xmin, xmax, ymin, ymax = cx - w * 0.5, cx + w * 0.5, cy - h * 0.5, cy + h * 0.5
    out = []
    for tx, ty in ((xmin, ymin), (xmin, ymax), (xmax, ymax), (xmax, ymin)):
        out += [tx * al + ty * be + r02, -tx * be + ty * al + r12]
    if is_t:
        ref = box if isinstance(box, torch.Tensor) else box[0]
        return torch.tensor(out, dtype=ref.dtype, device=ref.device)
    return np.array(out, dtype=np.float64)


def skew_bbox_iou(box1, box2, GIoU=False):
    """box1 [5] / [n,5] / list of 5 scalars, box2 [n,5] (cx, cy, w, h, angle) -> FloatTensor[n] of rotated IoUs with the
    semantics of the reference's shapely route (exact polygon IoU; 0 for zero-area boxes), computed by one launch of the
    fp64 convex-clip kernel."""
    if GIoU:
        raise NotImplementedError("only mode 'iou' is reachable from the reference's callers (test.py:146, nms.py:105)")
    if isinstance(box1, (list, tuple)):
        box1 = torch.stack([torch.as_tensor(b) for b in box1]).to(box2.device)
    if box1.dim() < box2.dim():
        box1 = box1.unsqueeze(0)
    if box1.shape[0] != box2.shape[0]:
        box1 = box1.repeat(len(box2), 1)
    return skew_iou_pairs(box1[:, :5].float().contiguous(), box2[:, :5].float().contiguous())


def match_predictions(pred, labels_px, iou_thres=0.5):
    """test.py:114-151 for one image.  pred [k,8] rows (x,y,w,h,a,score,cls_conf,cls) sorted by score; labels_px
    [nl,6] (cls, x, y, w, h, a) in pixels.  Returns the `correct` list (0/1 per prediction)."""
    correct = [0] * len(pred)
    nl = len(labels_px)
    if nl == 0 or len(pred) == 0:
        return correct
    iou = skew_iou_matrix(pred[:, :5].contiguous(), labels_px[:, 1:6].contiguous()).cpu()       # one launch per image
    pcls = pred[:, 7].cpu()
    tcls = labels_px[:, 0].cpu()
    detected = []
    for i in range(len(pred)):
        if len(detected) == nl:
            break
        m = (pcls[i] == tcls).nonzero().view(-1)
        if len(m) == 0:
            continue
        v, bi = iou[i, m].max(0)
        if v > iou_thres and int(m[bi]) not in detected:
            correct[i] = 1
            detected.append(int(m[bi]))
    return correct


def ap_per_class(tp, conf, pred_cls, target_cls):
    """Per-class precision / recall / AP / F1 at the end of the confidence-ranked list -- the quantities of the reference's
    ap_per_class (utils/utils.py:200-260).  The detections are ranked once; every class then takes its own rows of the ranked
    list (a boolean mask) and two cumulative sums over them, so the working set is O(detections) per class (the
    one-hot [detections x classes] form of round 4 held three dense float64 matrices -- about 2 GB at conf_thres 0.001 with 80
    classes).  Returns (p, r, ap, f1, classes) over the classes that occur in target_cls."""
    tp = np.asarray(tp, dtype=np.float64).reshape(-1)
    conf, pred_cls, target_cls = np.asarray(conf).reshape(-1), np.asarray(pred_cls).reshape(-1), np.asarray(target_cls).reshape(-1)
    classes = np.unique(target_cls)
    rank = np.argsort(-conf)
    tp, pred_cls = tp[rank], pred_cls[rank]
    p, r, ap = np.zeros(len(classes)), np.zeros(len(classes)), np.zeros(len(classes))
    for k, c in enumerate(classes):
        rows = pred_cls == c
        n_gt = int((target_cls == c).sum())                             # > 0 by construction
        if not rows.any():
            continue                                                    # classes without detections keep p = r = ap = 0
        cum_tp = np.cumsum(tp[rows])
        cum_fp = np.cumsum(1.0 - tp[rows])
        recall = cum_tp / (n_gt + 1e-16)
        precision = cum_tp / (cum_tp + cum_fp)
        r[k], p[k], ap[k] = recall[-1], precision[-1], compute_ap(recall, precision)
    f1 = 2 * p * r / (p + r + 1e-16)
    return p, r, ap, f1, classes.astype('int32')


def compute_ap(recall, precision):
    """Area under the precision envelope (VOC2010+ all-point interpolation, the reference's compute_ap, utils/utils.py:263-286):
    precision is replaced by its running maximum from the right, and the area is summed over the points where recall steps."""
    rec = np.concatenate(([0.0], np.asarray(recall, dtype=np.float64), [1.0]))
    pre = np.concatenate(([0.0], np.asarray(precision, dtype=np.float64), [0.0]))
    envelope = np.maximum.accumulate(pre[::-1])[::-1]
    step = np.flatnonzero(rec[1:] != rec[:-1])
    return float(np.sum((rec[step + 1] - rec[step]) * envelope[step + 1]))


def scale_coords(img1_shape, coords, img0_shape):
    """Rescale (x, y, w, h) from the letterboxed img1_shape back to img0_shape (utils/utils.py:181-189)."""
    gain = max(img1_shape) / max(img0_shape)
    coords[:, [0]] -= (img1_shape[1] - img0_shape[1] * gain) / 2
    coords[:, [1]] -= (img1_shape[0] - img0_shape[0] * gain) / 2
    coords[:, :4] /= gain
    return coords
