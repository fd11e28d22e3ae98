"""non_max_suppression -- mirror of the reference's utils/nms/nms.py:4-147 (the live `use_cuda_nms` branch :53-69).

    non_max_suppression(prediction[bs, N, 6+nc], conf_thres=0.5, nms_thres=0.5) -> list[bs] of [k, 8] or None
    rows: (x, y, w, h, a, obj_conf*class_conf, class_conf, class), sorted by score descending.

Kept quirks of the reference: `prediction` is modified in place (pred[:, 5] *= class_conf, nms.py:35); boxes with
w or h <= 2 px or any non-finite entry are dropped (:40); detections are NMS-ed per class with `r_nms` on the
score-sorted rows (:57-66).  The three argsorts are stable here (ties keep the lower row first) so that the
result is deterministic; the reference's are not (any tie order is a valid reference output).  The dead pure-Python branch (:71-144, with its live ipdb breakpoint) is not reproduced.
"""
import torch

from .r_nms import r_nms


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
