"""non_max_suppression -- mirror of the reference's utils/nms/nms.py:4-147 (the live `use_cuda_nms` branch :53-69).

    non_max_suppression(prediction[bs, N, 6+nc], conf_thres=0.5, nms_thres=0.5) -> list[bs] of [k, 8] or None
    rows: (x, y, w, h, a, obj_conf*class_conf, class_conf, class), sorted by score descending.

Kept quirks of the reference: `prediction` is modified in place (pred[:, 5] *= class_conf, nms.py:35); boxes with
w or h <= 2 px or any non-finite entry are dropped (:40); detections are NMS-ed per class with `r_nms` on the
score-sorted rows (:57-66).  The three argsorts are stable here (ties keep the lower row first) so that the
result is deterministic; the reference's are not (any tie order is a valid reference output).  The dead pure-Python branch (:71-144, with its live ipdb breakpoint) is not reproduced.
"""
import torch

from .r_nms import r_nms, r_nms_segmented


def non_max_suppression(prediction, conf_thres=0.5, nms_thres=0.5):
    min_wh = 2
    output = [None] * len(prediction)
    for image_i, pred in enumerate(prediction):
        if prediction.numel() == 0:
            continue
        class_conf, class_pred = pred[:, 6:].max(1)
        pred[:, 5] *= class_conf
        i = (pred[:, 5] > conf_thres) & (pred[:, 2:4] > min_wh).all(1) & torch.isfinite(pred).all(1)
        pred = pred[i]
        if len(pred) == 0:
            continue
        class_conf = class_conf[i]
        class_pred = class_pred[i].unsqueeze(1).float()
        det_max = []
        pred = torch.cat((pred[:, :6], class_conf.unsqueeze(1), class_pred), 1)
        pred = pred[(-pred[:, 5]).argsort(stable=True)]
        for c in pred[:, -1].unique():
            dc = pred[pred[:, -1] == c]
            dc = dc[(-dc[:, 5]).argsort(stable=True)]
            inds = r_nms(dc[:, :6], nms_thres)
            det_max.append(dc[inds.to(dc.device)])
        if len(det_max):
            det_max = torch.cat(det_max)
            output[image_i] = det_max[(-det_max[:, 5]).argsort(stable=True)]
    return output


def non_max_suppression_batched(prediction, conf_thres=0.5, nms_thres=0.5):
    """Same result as non_max_suppression (row for row, order included) for a GPU `prediction`, without the Python loop
    over images and classes: one vectorised filter over the whole batch, one stable (image, class, score) ordering, ONE
    segmented rotated-NMS launch over all (image, class) sets (r_nms_segmented), one final per-image score ordering.
    Three small device->host reads (candidate count, set sizes, rows per image) instead of several per image and class.
    Like the reference it scales prediction[..., 5] by the class confidence in place (nms.py:35)."""
    if not prediction.is_cuda:
        return non_max_suppression(prediction, conf_thres, nms_thres)
    bs, n, no = prediction.shape
    output = [None] * bs
    if prediction.numel() == 0:
        return output
    min_wh = 2
    class_conf, class_pred = prediction[..., 6:].max(2)
    prediction[..., 5] *= class_conf
    ok = (prediction[..., 5] > conf_thres) & (prediction[..., 2:4] > min_wh).all(2) & torch.isfinite(prediction).all(2)
    idx = ok.nonzero()                                            # [M, 2] (image, row), ascending     -- host read 1
    if idx.shape[0] == 0:
        return output
    img, row = idx[:, 0], idx[:, 1]
    rows = prediction[img, row]                                   # [M, no]
    cand = torch.cat((rows[:, :6], class_conf[img, row].unsqueeze(1), class_pred[img, row].unsqueeze(1).float()), 1)
    return nms_from_candidates(img, cand, bs, nms_thres, nc=no - 6)


def nms_from_candidates(img, cand, bs, nms_thres, nc=None):
    """Second half of non_max_suppression for a whole batch.  cand [M, 8] = (x, y, w, h, a, score, class_conf, class)
    of the rows that passed the confidence filter, in (image, row) order; img [M] their image index."""
    output = [None] * bs
    cp = cand[:, 7].long()
    if nc is None:
        nc = int(cp.max()) + 1 if cand.shape[0] else 1
    # (image, class) ascending, score descending, ties in row order: two stable sorts
    o1 = (-cand[:, 5]).argsort(stable=True)
    seg = (img * nc + cp)[o1]
    o2 = seg.argsort(stable=True)
    order = o1[o2]
    seg = seg[o2]
    det = cand[order]
    _, counts = torch.unique_consecutive(seg, return_counts=True)
    max_len = int(counts.max())                                   #                                    -- host read
    seg_off = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=det.device)
    seg_off[1:] = counts.cumsum(0)
    keep = r_nms_segmented(det[:, :6].contiguous(), seg_off, max_len, nms_thres).bool()
    det = det[keep]
    dimg = img[order][keep]
    # per image: score descending; ties keep (class, row) order -- what the reference's final argsort gives on the
    # class-by-class concatenation
    o3 = (-det[:, 5]).argsort(stable=True)
    o4 = dimg[o3].argsort(stable=True)
    det = det[o3[o4]]
    per_img = torch.bincount(dimg, minlength=bs).tolist()         #                                    -- host read
    start = 0
    for b, k in enumerate(per_img):
        if k:
            output[b] = det[start:start + k]
            start += k
    return output
