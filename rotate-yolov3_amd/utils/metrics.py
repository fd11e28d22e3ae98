"""Evaluation helpers -- mirrors of the reference's utils/utils.py: skew_bbox_iou (:290-320, here one HIP launch
instead of a Python + shapely loop per pair), ap_per_class (:200-261), compute_ap (:264-286), scale_coords (:181-189),
and the per-image greedy matching of test.py:114-151 expressed on the rotated-IoU matrix."""
import numpy as np
import torch

from .nms.r_nms import riou_matrix, riou_pairs


def skew_bbox_iou(box1, box2, GIoU=False):
    """box1 [5] / [n,5], box2 [n,5] (cx, cy, w, h, angle) -> FloatTensor[n] of rotated IoUs (polygon IoU with the
    arithmetic of the reference's native kernel; the reference's Python path calls shapely/GEOS in fp64 -- same value
    to ~1e-6 on non-degenerate boxes)."""
    if GIoU:
        raise NotImplementedError("only mode 'iou' is reachable from the reference's callers (test.py:146, nms.py:105)")
    if isinstance(box1, (list, tuple)):
        box1 = torch.stack([torch.as_tensor(b) for b in box1]).to(box2.device)
    if box1.dim() < box2.dim():
        box1 = box1.unsqueeze(0)
    if box1.shape[0] != box2.shape[0]:
        box1 = box1.repeat(len(box2), 1)
    return riou_pairs(box1[:, :5].float().contiguous(), box2[:, :5].float().contiguous())


def match_predictions(pred, labels_px, iou_thres=0.5):
    """test.py:114-151 for one image.  pred [k,8] rows (x,y,w,h,a,score,cls_conf,cls) sorted by score; labels_px
    [nl,6] (cls, x, y, w, h, a) in pixels.  Returns the `correct` list (0/1 per prediction)."""
    correct = [0] * len(pred)
    nl = len(labels_px)
    if nl == 0 or len(pred) == 0:
        return correct
    iou = riou_matrix(pred[:, :5].contiguous(), labels_px[:, 1:6].contiguous()).cpu()       # one launch per image
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
    i = np.argsort(-conf)
    tp, conf, pred_cls = tp[i], conf[i], pred_cls[i]
    unique_classes = np.unique(target_cls)
    ap, p, r = [], [], []
    for c in unique_classes:
        i = pred_cls == c
        n_gt = (target_cls == c).sum()
        n_p = i.sum()
        if n_p == 0 and n_gt == 0:
            continue
        elif n_p == 0 or n_gt == 0:
            ap.append(0)
            r.append(0)
            p.append(0)
        else:
            fpc = (1 - tp[i]).cumsum()
            tpc = (tp[i]).cumsum()
            recall = tpc / (n_gt + 1e-16)
            r.append(recall[-1])
            precision = tpc / (tpc + fpc)
            p.append(precision[-1])
            ap.append(compute_ap(recall, precision))
    p, r, ap = np.array(p), np.array(r), np.array(ap)
    f1 = 2 * p * r / (p + r + 1e-16)
    return p, r, ap, f1, unique_classes.astype('int32')


def compute_ap(recall, precision):
    mrec = np.concatenate(([0.], recall, [1.]))
    mpre = np.concatenate(([0.], precision, [0.]))
    for i in range(mpre.size - 1, 0, -1):
        mpre[i - 1] = np.maximum(mpre[i - 1], mpre[i])
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def scale_coords(img1_shape, coords, img0_shape):
    """Rescale (x, y, w, h) from the letterboxed img1_shape back to img0_shape (utils/utils.py:181-189)."""
    gain = max(img1_shape) / max(img0_shape)
    coords[:, [0]] -= (img1_shape[1] - img0_shape[1] * gain) / 2
    coords[:, [1]] -= (img1_shape[0] - img0_shape[0] * gain) / 2
    coords[:, :4] /= gain
    return coords
