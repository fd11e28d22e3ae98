"""Generate golden vectors for the model-level host logic FROM THE REFERENCE's own Python, imported in the build
container with stub modules for the packages this image lacks (SURVEY.md section 8c recipe).  Nothing of the
reference is written into the repo: only inputs and expected outputs (.npz).

    python tests/golden/gen_model_golden.py        (needs /root/reference)

Fixtures:
  parser_ara.npz        anchors the reference parser yields for the `ara` anchor line used by cfg/make_cfg.py
  decode_head0.npz      YOLOLayer eval decode of a seeded head tensor [1, 504, 4, 4], img 128x128 -> io [1,1152,7]
  forward_d53_64.npz    Darknet-53 (cfg from make_cfg.darknet53, procedural weights) on rand(1,3,64,64): io, p[3],
                        per-conv-block output statistics (mean, mean|.|)
  forward_tiny_64.npz   yolov3-tiny topology (reference-loadable spelling of make_cfg.tiny) on rand(1,3,64,64): io
  nms_wrapper.npz       non_max_suppression(prediction[2,4096,7], 0.3, 0.5) with r_nms backed by the reference's
                        own IoU arithmetic (oracle/_ref): per-image [k,8] rows
"""
import math
import os
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"

from tests.procedural import fill_procedural  # noqa: E402
from oracle import riou  # noqa: E402


def install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None

    def getRotationMatrix2D(center, angle, scale):
        a = scale * math.cos(math.radians(angle))
        b = scale * math.sin(math.radians(angle))
        cx, cy = center
        return np.array([[a, b, (1 - a) * cx - b * cy], [-b, a, b * cx + (1 - a) * cy]], dtype=np.float64)
    cv2.getRotationMatrix2D = getRotationMatrix2D
    sys.modules["cv2"] = cv2
    sh = types.ModuleType("shapely")
    shg = types.ModuleType("shapely.geometry")
    shg.Polygon = object
    shg.MultiPoint = object
    sh.geometry = shg
    sys.modules["shapely"] = sh
    sys.modules["shapely.geometry"] = shg
    import matplotlib
    matplotlib.use("Agg")
    rn = types.ModuleType("utils.nms.r_nms")

    def r_nms(dets, thr):
        d = dets.detach().cpu().numpy().astype(np.float32)
        if d.shape[0] == 0:
            return torch.empty(0, dtype=torch.long)
        return torch.from_numpy(riou.rnms(np.ascontiguousarray(d), float(thr), use_ref=True))
    rn.r_nms = r_nms
    sys.modules["utils.nms.r_nms"] = rn


def main():
    assert os.path.isdir(REF)
    install_stubs()
    sys.path.insert(0, REF)
    cwd = os.getcwd()
    os.chdir(REF)
    from utils import parse_config as rpc
    from model import models as rmodels
    from utils.nms import nms as rnms_mod
    sys.path.insert(0, ROOT)
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.cfg import make_cfg

    tmp = tempfile.mkdtemp()
    # ---- parser
    cfg_txt = make_cfg.darknet53(width=608, height=608)
    cfg_path = os.path.join(tmp, "d53.cfg")
    open(cfg_path, "w").write(cfg_txt)
    defs = rpc.parse_model_cfg(cfg_path)
    yolo = [d for d in defs if d["type"] == "yolo"]
    np.savez_compressed(os.path.join(OUT, "parser_ara.npz"), anchors=yolo[0]["anchors"], n_blocks=len(defs),
                        types=np.array([d["type"] for d in defs]))

    hyp = {"context_factor": 1.0}
    # ---- decode
    model = rmodels.Darknet(cfg_path, hyp).eval()
    g = torch.Generator().manual_seed(3)
    head = torch.randn(1, 504, 4, 4, generator=g)
    yl = model.module_list[82]
    with torch.no_grad():
        io, p = yl(head.clone(), (128, 128))
    np.savez_compressed(os.path.join(OUT, "decode_head0.npz"), head=head.numpy(), io=io.numpy(), p=p.numpy(),
                        anchors=yolo[0]["anchors"][144:216])

    # ---- forward walk, Darknet-53 @64x64 (fresh model: the reference's create_grids shrinks anchors in place)
    model = rmodels.Darknet(cfg_path, hyp).eval()
    fill_procedural(model)
    x = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0))
    stats = {}

    def hook(i):
        def fn(m, inp, out):
            stats[i] = (float(out.mean()), float(out.abs().mean()))
        return fn
    for i, (d, m) in enumerate(zip(model.module_defs, model.module_list)):
        if d["type"] == "convolutional":
            m.register_forward_hook(hook(i))
    with torch.no_grad():
        io, p = model(x)
    idx = sorted(stats)
    np.savez_compressed(os.path.join(OUT, "forward_d53_64.npz"), x=x.numpy(), io=io.numpy(),
                        p0=p[0].numpy(), p1=p[1].numpy(), p2=p[2].numpy(), conv_idx=np.array(idx),
                        conv_stats=np.array([stats[i] for i in idx], dtype=np.float64))

    # ---- tiny: reference-loadable spelling (anchor txt file, range masks, head width 36*86)
    tiny_txt = make_cfg.tiny()
    pairs = os.path.join(tmp, "tiny_pairs.txt")
    nums = [float(v) for v in make_cfg.TINY_PAIRS.split(",")]
    open(pairs, "w").write("\n".join("%g %g" % (nums[i], nums[i + 1]) for i in range(0, len(nums), 2)) + "\n")
    tiny_ref = tiny_txt.replace("anchors = " + make_cfg.TINY_PAIRS, "anchors = " + pairs)
    tiny_ref = tiny_ref.replace("mask = 3,4,5", "mask = 36-71").replace("mask = 1,2,3", "mask = 12-47")
    tpath = os.path.join(tmp, "tiny.cfg")
    open(tpath, "w").write(tiny_ref)
    tm = rmodels.Darknet(tpath, hyp).eval()
    fill_procedural(tm)
    with torch.no_grad():
        io_t, p_t = tm(x)
    np.savez_compressed(os.path.join(OUT, "forward_tiny_64.npz"), x=x.numpy(), io=io_t.numpy())

    # ---- NMS wrapper
    g = torch.Generator().manual_seed(11)
    n = 4096
    pred = torch.empty(2, n, 7)
    pred[..., 0:2] = torch.rand(2, n, 2, generator=g) * 300
    pred[..., 2:4] = 1.0 + 60 * torch.rand(2, n, 2, generator=g)      # some w/h <= 2 px get filtered
    pred[..., 4] = (torch.rand(2, n, generator=g) - 0.5) * math.pi
    pred[..., 5] = torch.rand(2, n, generator=g)
    pred[..., 6] = 1.0
    pred[1, 7, 0] = float("nan")
    pred_in = pred.clone()
    torch.Tensor.cuda = lambda self, *a, **k: self
    out = rnms_mod.non_max_suppression(pred, 0.3, 0.5)
    np.savez_compressed(os.path.join(OUT, "nms_wrapper.npz"), pred=pred_in.numpy(), det0=out[0].numpy(),
                        det1=out[1].numpy(), pred_after=pred.numpy())
    os.chdir(cwd)
    print("io", io.shape, "tiny", io_t.shape, "nms", out[0].shape, out[1].shape)


if __name__ == "__main__":
    main()
