"""CPU restatement of the reference's model-level path: cfg walk, conv blocks, YOLO decode, NMS wrapper.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows:
  parse_model_cfg / cfg2anchors        /root/reference/utils/parse_config.py:37-59, :6-31
  create_modules (shapes, routes)      model/models.py:36-164
  Darknet.forward                      model/models.py:244-298
  YOLOLayer.forward + create_grids     model/models.py:183-227, model/model_utils.py:16-35
  non_max_suppression                  utils/nms/nms.py:4-69 (live branch)
Pinned by tests/test_oracle_model.py against tests/golden/{parser_ara,decode_head0,forward_d53_64,forward_tiny_64,
nms_wrapper}.npz, which tests/golden/gen_model_golden.py produced by importing and running the reference itself.

`forward(..., bf16=True)` evaluates the SAME graph under the arithmetic contract of include/ryolo.h (bf16 tensors,
fp32 accumulation, folded BN applied in fp32, one rounding after the activation and one after a shortcut add); that
mode is what the HIP engine is compared with.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import riou


def _r(t, on):
    return t.to(torch.bfloat16).to(torch.float32) if on else t


def parse_cfg(text):
    blocks = []
    for line in text.split("\n"):
        line = line.strip()
        if not line or line.startswith("#"):
            continue
        if line.startswith("["):
            blocks.append({"type": line[1:-1].strip()})
            if blocks[-1]["type"] == "convolutional":
                blocks[-1]["batch_normalize"] = 0
        else:
            k, v = line.split("=", 1)
            blocks[-1][k.strip()] = v.strip()
    return blocks


def anchors_of(val):
    """`ara a../r../d..` (parse_config.py:7-24, prefix optional) or stock `w,h,...` pairs x 12 angles (:25-31)."""
    v = val.replace("ara", "")
    if "/" in v:
        a, r, d = [[float(t) for t in part.split(",") if t.strip()] for part in v.split("/") if part.strip()]
        out = [[math.sqrt(ar * ra), math.sqrt(ar / ra), dg * math.pi / 180] for ar in a for ra in r for dg in d]
        return np.array(out, dtype=np.float64)
    nums = [float(t) for t in v.split(",") if t.strip()]
    ang = np.array([i for i in range(-6, 6)]) * math.pi / 12
    return np.array([[nums[i], nums[i + 1], a] for i in range(0, len(nums), 2) for a in ang], dtype=np.float64)


def mask_of(m):
    if "-" in m:
        lo, hi = [int(t) for t in m.split("-")]
        return list(range(lo, hi + 1))
    out = []
    for p in [int(t) for t in m.split(",")]:
        out.extend(range(12 * p, 12 * p + 12))
    return out


def decode(p, anchors, img_size, cf=1.0, arc="default", nc=1):
    """p: [bs, na*(nc+6), ny, nx] head tensor.  Returns (io [bs, na*ny*nx, nc+6], p5 [bs, na, ny, nx, nc+6])."""
    bs, _, ny, nx = p.shape
    na = len(anchors)
    no = nc + 6
    stride = max(img_size) / max(nx, ny)
    yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing="ij")
    grid = torch.stack((xv, yv), 2).float().view(1, 1, ny, nx, 2)
    av = torch.tensor(np.asarray(anchors), dtype=torch.float32).clone()
    av[:, :2] /= stride
    awh = av.view(1, na, 1, 1, 3)
    p5 = p.view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous()
    io = p5.clone()
    io[..., 0:2] = torch.sigmoid(io[..., 0:2]) + grid
    io[..., 2:4] = torch.exp(io[..., 2:4]) * awh[..., :-1]
    io[..., 4] = torch.atan(io[..., 4]) + awh[..., -1]
    io[..., :4] *= stride
    io[..., 3] /= cf
    io[..., 2] -= io[..., 3] * (cf - 1)
    if "default" in arc:
        io[..., 5:] = torch.sigmoid(io[..., 5:])
    elif "BCE" in arc:
        io[..., 6:] = torch.sigmoid(io[..., 6:])
        io[..., 5] = 1
    if nc == 1:
        io[..., 6] = 1
    return io.view(bs, -1, no), p5


def forward(cfg_text, sd, x, hyp=None, arc="default", bf16=False, return_layers=False):
    """Eval-mode forward of the cfg graph with weights from a reference-layout state_dict `sd`."""
    blocks = parse_cfg(cfg_text)
    net, defs = blocks[0], blocks[1:]
    cf = float((hyp or {}).get("context_factor", 1.0))
    img_size = x.shape[-2:]
    x = _r(x.float(), bf16)
    outs, ios, ps = [], [], []
    for i, d in enumerate(defs):
        t = d["type"]
        if t == "convolutional":
            pre = "module_list.%d." % i
            w = _r(sd[pre + "Conv2d.weight"].float(), bf16)
            k = w.shape[-1]
            pad = (k - 1) // 2 if int(d.get("pad", 0)) else 0
            y = F.conv2d(x, w, None, stride=int(d["stride"]), padding=pad)
            if int(d["batch_normalize"]):
                g, b = sd[pre + "BatchNorm2d.weight"].float(), sd[pre + "BatchNorm2d.bias"].float()
                m, v = sd[pre + "BatchNorm2d.running_mean"].float(), sd[pre + "BatchNorm2d.running_var"].float()
                if bf16:   # folded form, as the engine applies it
                    scale = g / torch.sqrt(v + 1e-5)
                    shift = b - m * scale
                    y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
                else:
                    y = F.batch_norm(y, m, v, g, b, False, 0.1, 1e-5)
            else:
                y = y + sd[pre + "Conv2d.bias"].float().view(1, -1, 1, 1)
            if d.get("activation") == "leaky":
                y = F.prelu(y, sd[pre + "activation.weight"].float())
            x = _r(y, bf16)
        elif t == "maxpool":
            k, s = int(d["size"]), int(d["stride"])
            if k == 2 and s == 1:
                x = F.max_pool2d(F.pad(x, (0, 1, 0, 1)), k, s, 0)
            else:
                x = F.max_pool2d(x, k, s, (k - 1) // 2)
        elif t == "upsample":
            x = F.interpolate(x, scale_factor=int(d["stride"]), mode="nearest")
        elif t == "route":
            ls = [int(v) for v in d["layers"].split(",")]
            x = outs[ls[0]] if len(ls) == 1 else torch.cat([outs[l] for l in ls], 1)
        elif t == "shortcut":
            x = _r(x + outs[int(d["from"])], bf16)
        elif t == "yolo":
            anchors = anchors_of(d["anchors"])[mask_of(d["mask"])]
            io, p5 = decode(x, anchors, img_size, cf, arc, int(d["classes"]))
            ios.append(io)
            ps.append(p5)
        outs.append(x)
    res = (torch.cat(ios, 1), ps)
    return res + (outs,) if return_layers else res


def non_max_suppression(prediction, conf_thres=0.5, nms_thres=0.5):
    """utils/nms/nms.py:4-69; r_nms evaluated by the CPU oracle.  Modifies `prediction` in place like the reference."""
    min_wh = 2
    output = [None] * len(prediction)
    for image_i, pred in enumerate(prediction):
        class_conf, class_pred = pred[:, 6:].max(1)
        pred[:, 5] *= class_conf
        i = (pred[:, 5] > conf_thres) & (pred[:, 2:4] > min_wh).all(1) & torch.isfinite(pred).all(1)
        pred = pred[i]
        if len(pred) == 0:
            continue
        class_conf = class_conf[i]
        class_pred = class_pred[i].unsqueeze(1).float()
        pred = torch.cat((pred[:, :6], class_conf.unsqueeze(1), class_pred), 1)
        pred = pred[(-pred[:, 5]).argsort(stable=True)]
        det_max = []
        for c in pred[:, -1].unique():
            dc = pred[pred[:, -1] == c]
            dc = dc[(-dc[:, 5]).argsort(stable=True)]
            inds = torch.from_numpy(riou.rnms(dc[:, :6].numpy(), float(nms_thres)))
            det_max.append(dc[inds])
        if len(det_max):
            det_max = torch.cat(det_max)
            output[image_i] = det_max[(-det_max[:, 5]).argsort(stable=True)]
    return output
