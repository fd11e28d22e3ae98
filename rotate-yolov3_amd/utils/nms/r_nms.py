"""r_nms -- mirror of the reference's native module `utils.nms.r_nms`
(utils/nms/src/rotate_polygon_nms.cpp:7-16; built by utils/nms/setup.py).

    r_nms(dets: Tensor[N,6] float32 on the GPU, threshold: float) -> LongTensor[K]

Same contract as the reference: columns (cx, cy, w, h, angle_rad, score); returns a new int64 tensor on dets'
device with the kept row indices in ascending order; N == 0 returns an empty int64 CPU tensor (cpp:9-10);
a non-GPU tensor raises RuntimeError like the reference's AT_CHECK (cpp:3,8).  `dets` may be a column slice
(utils/nms/nms.py:64 passes dc[:, :6] of an [n,8] tensor): the row stride is forwarded, nothing is copied.
The work happens in libryolo_hip.so (csrc/rnms.hip) through the C ABI `ryolo_rnms`; there is no CPU path.
"""
import torch

from ... import _lib

_workspaces = {}


def _workspace(device, nbytes):
    key = (device.type, device.index)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes * 1.25) if ws is not None else nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = ws
    return ws


def r_nms(dets, threshold):
    if not isinstance(dets, torch.Tensor) or not dets.is_cuda:
        raise RuntimeError("dets must be a CUDAtensor ")
    if dets.numel() == 0:
        return torch.empty(0, dtype=torch.long)
    if dets.dim() != 2 or dets.size(1) < 6:
        raise RuntimeError("dets must be [N, >=6] (cx, cy, w, h, angle, score)")
    if dets.dtype != torch.float32:
        dets = dets.float()
    if dets.stride(1) != 1 or dets.stride(0) < 6:
        dets = dets.contiguous()
    n = dets.size(0)
    L = _lib.lib()
    nbytes = L.ryolo_rnms_workspace_bytes(n)
    if nbytes == 0:
        _lib.check(-3, "ryolo_rnms")
    with torch.cuda.device(dets.device):
        ws = _workspace(dets.device, nbytes)
        keep = torch.empty(n, dtype=torch.long, device=dets.device)
        cnt = torch.empty(1, dtype=torch.int32, device=dets.device)
        rc = L.ryolo_rnms(dets.data_ptr(), n, dets.stride(0), float(threshold), keep.data_ptr(), cnt.data_ptr(),
                          ws.data_ptr(), ws.numel(), _lib.stream_ptr(dets.device))
        _lib.check(rc, "ryolo_rnms")
        k = int(cnt.item())   # the only host sync; the reference blocks on the whole n x n/64 mask instead
    return keep[:k]


def r_nms_segmented(dets, seg_offsets, max_seg_len, threshold):
    """Greedy rotated NMS of many independent sets in one launch (ryolo_rnms_segmented): rows [seg_offsets[s],
    seg_offsets[s+1]) of `dets` [M, >=6] are set s and must already be sorted by score, highest first.
    seg_offsets: int32 [S+1] on the device; max_seg_len: host int >= the longest set.  Returns uint8 keep flags [M]."""
    if not dets.is_cuda:
        raise RuntimeError("dets must be a CUDAtensor ")
    if dets.dtype != torch.float32 or dets.stride(1) != 1 or dets.stride(0) < 6:
        dets = dets.float().contiguous()
    m, S = dets.size(0), seg_offsets.numel() - 1
    flags = torch.zeros(m, dtype=torch.uint8, device=dets.device)
    if m == 0 or S <= 0:
        return flags
    seg_offsets = seg_offsets.to(device=dets.device, dtype=torch.int32).contiguous()
    L = _lib.lib()
    nbytes = L.ryolo_rnms_segmented_workspace_bytes(m, S, int(max_seg_len))
    if nbytes == 0:
        _lib.check(-3, "ryolo_rnms_segmented")
    with torch.cuda.device(dets.device):
        ws = _workspace(dets.device, nbytes)
        _lib.check(L.ryolo_rnms_segmented(dets.data_ptr(), m, dets.stride(0), seg_offsets.data_ptr(), S, int(max_seg_len),
                                          float(threshold), flags.data_ptr(), ws.data_ptr(), ws.numel(),
                                          _lib.stream_ptr(dets.device)), "ryolo_rnms_segmented")
    return flags


def riou_pairs(box1, box2):
    """IoU(box1[i], box2[i]) with the arithmetic of devRotateIoU (kernel.cu:251-260); rows (cx,cy,w,h,a,...)."""
    b1, b2 = _rows(box1), _rows(box2)
    assert b1.size(0) == b2.size(0)
    out = torch.empty(b1.size(0), dtype=torch.float32, device=b1.device)
    with torch.cuda.device(b1.device):
        rc = _lib.lib().ryolo_riou_pairs(b1.data_ptr(), b1.stride(0), b2.data_ptr(), b2.stride(0), b1.size(0),
                                         out.data_ptr(), _lib.stream_ptr(b1.device))
    _lib.check(rc, "ryolo_riou_pairs")
    return out


def riou_matrix(box1, box2):
    """out[i, j] = IoU(box1[i], box2[j])."""
    b1, b2 = _rows(box1), _rows(box2)
    out = torch.empty(b1.size(0), b2.size(0), dtype=torch.float32, device=b1.device)
    with torch.cuda.device(b1.device):
        rc = _lib.lib().ryolo_riou_matrix(b1.data_ptr(), b1.size(0), b1.stride(0), b2.data_ptr(), b2.size(0),
                                          b2.stride(0), out.data_ptr(), _lib.stream_ptr(b1.device))
    _lib.check(rc, "ryolo_riou_matrix")
    return out


def skew_iou_pairs(box1, box2):
    """IoU(box1[i], box2[i]) of the evaluation path (reference utils.py:290-320: exact polygon IoU in fp64)."""
    b1, b2 = _rows(box1), _rows(box2)
    assert b1.size(0) == b2.size(0)
    out = torch.empty(b1.size(0), dtype=torch.float32, device=b1.device)
    with torch.cuda.device(b1.device):
        rc = _lib.lib().ryolo_skew_iou_pairs(b1.data_ptr(), b1.stride(0), b2.data_ptr(), b2.stride(0), b1.size(0),
                                             out.data_ptr(), _lib.stream_ptr(b1.device))
    _lib.check(rc, "ryolo_skew_iou_pairs")
    return out


def skew_iou_matrix(box1, box2):
    """out[i, j] = evaluation-path IoU(box1[i], box2[j])."""
    b1, b2 = _rows(box1), _rows(box2)
    out = torch.empty(b1.size(0), b2.size(0), dtype=torch.float32, device=b1.device)
    with torch.cuda.device(b1.device):
        rc = _lib.lib().ryolo_skew_iou_matrix(b1.data_ptr(), b1.size(0), b1.stride(0), b2.data_ptr(), b2.size(0),
                                              b2.stride(0), out.data_ptr(), _lib.stream_ptr(b1.device))
    _lib.check(rc, "ryolo_skew_iou_matrix")
    return out


def _rows(b):
    if not b.is_cuda:
        raise RuntimeError("boxes must be a CUDAtensor ")
    if b.dim() != 2 or b.size(1) < 5:
        raise RuntimeError("boxes must be [N, >=5]")
    if b.dtype != torch.float32:
        b = b.float()
    if b.stride(1) != 1 or b.stride(0) < 5:
        b = b.contiguous()
    return b
