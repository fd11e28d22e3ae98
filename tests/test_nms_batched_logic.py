"""CPU tier: the ordering / segmentation logic of non_max_suppression_batched (utils/nms/nms.py), with the segmented HIP
NMS replaced by the CPU oracle per segment.  What runs on the GPU differs only in who evaluates each (image, class)
set; the (image, class, score) ordering, the segment table, the final per-image order and the in-place score update
are the code under test here and must reproduce the reference wrapper (oracle/darknet_oracle.non_max_suppression,
itself pinned to the golden captured from the reference)."""
import math

import numpy as np
import torch

import rotate_yolov3_amd  # noqa: F401
from oracle import darknet_oracle as do
from oracle import riou
from rotate_yolov3_amd.utils.nms import nms as nms_mod


def _oracle_segmented(dets, seg_off, max_len, thr):
    flags = torch.zeros(dets.shape[0], dtype=torch.uint8)
    off = seg_off.tolist()
    assert max(b - a for a, b in zip(off[:-1], off[1:])) <= max_len
    for a, b in zip(off[:-1], off[1:]):
        seg = dets[a:b]
        assert bool((seg[:-1, 5] >= seg[1:, 5]).all())            # the contract: score-sorted inside each set
        keep = riou.rnms(seg[:, :6].contiguous().numpy(), float(thr))
        flags[a + torch.from_numpy(keep)] = 1
    return flags


def _batched_on_cpu(prediction, conf_thres, nms_thres):
    # the first half of non_max_suppression_batched (vectorised filter), then the function under test
    bs, n, no = prediction.shape
    class_conf, class_pred = prediction[..., 6:].max(2)
    prediction[..., 5] *= class_conf
    ok = (prediction[..., 5] > conf_thres) & (prediction[..., 2:4] > 2).all(2) & torch.isfinite(prediction).all(2)
    idx = ok.nonzero()
    if idx.shape[0] == 0:
        return [None] * bs
    img, row = idx[:, 0], idx[:, 1]
    rows = prediction[img, row]
    cand = torch.cat((rows[:, :6], class_conf[img, row].unsqueeze(1), class_pred[img, row].unsqueeze(1).float()), 1)
    return nms_mod.nms_from_candidates(img, cand, bs, nms_thres, nc=no - 6)


def test_batched_wrapper_logic_equals_reference_wrapper(monkeypatch):
    monkeypatch.setattr(nms_mod, "r_nms_segmented", _oracle_segmented)
    g = torch.Generator().manual_seed(3)
    bs, n, nc = 4, 900, 3
    p = torch.empty(bs, n, 6 + nc)
    p[..., 0:2] = torch.rand(bs, n, 2, generator=g) * 120
    p[..., 2:4] = 6 * 8 ** torch.rand(bs, n, 2, generator=g)
    p[..., 4] = (torch.rand(bs, n, generator=g) - 0.5) * math.pi
    p[..., 5] = torch.rand(bs, n, generator=g)
    p[..., 6:] = torch.rand(bs, n, nc, generator=g)
    p[2, :, 5] = 0.0                       # an image without detections
    p[1, :40, 5] = 0.8                     # ties: same score and class
    p[1, :40, 6:] = 1.0
    p[3, 7, 1] = float("inf")
    p[3, 9, 3] = 1.5                       # too small
    a, b = p.clone(), p.clone()
    want = do.non_max_suppression(a, 0.45, 0.3)
    got = _batched_on_cpu(b, 0.45, 0.3)
    assert torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))   # same in-place score update
    assert [w is None for w in want] == [x is None for x in got]
    total = 0
    for w, x in zip(want, got):
        if w is not None:
            assert np.array_equal(w.numpy(), x.numpy())
            total += len(w)
    assert total > 100
