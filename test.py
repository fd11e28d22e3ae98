#!/usr/bin/env python
"""test.py -- evaluation entry point mirroring the reference's test.py (test() :17-201, flags :205-217):
forward (HIP engine) -> non_max_suppression -> per-image greedy matching on rotated IoU -> ap_per_class.
The reference matches predictions with a per-pair Python + shapely loop (test.py:134-151, utils/utils.py:290-320);
here one `ryolo_riou_matrix` launch per image.  Data: synthetic loader (the OpenCV loader is out of scope)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.model.models import Darknet  # noqa: E402
from rotate_yolov3_amd.utils.metrics import ap_per_class, match_predictions  # noqa: E402
from rotate_yolov3_amd.utils.nms.nms import non_max_suppression_batched as non_max_suppression  # noqa: E402  (same rows, one segmented NMS launch per batch; CPU tensors take the loop)
from rotate_yolov3_amd.utils.parse_config import hyp_parse  # noqa: E402
from rotate_yolov3_amd.utils.synthetic import SyntheticLoader  # noqa: E402


def test(cfg, hyp, weights=None, batch_size=16, img_size=608, iou_thres=0.5, conf_thres=0.001, nms_thres=0.5, model=None,
         n_images=32, device=None, nc=None):
    device = device or torch.device('cuda:0')
    if model is None:
        model = Darknet(cfg, hyp)
        if weights and weights.endswith('.pt'):
            model.load_state_dict(torch.load(weights, map_location='cpu')['model'])
        elif weights:                                # darknet format (reference test.py:47-48)
            from rotate_yolov3_amd.model.model_utils import load_darknet_weights
            load_darknet_weights(model, weights)
        model.to(device)
    model.eval()
    nc = int(nc) if nc else model.nc
    seen = 0
    stats = []
    with torch.no_grad():
        for imgs, targets, _, _ in SyntheticLoader(n_images, batch_size, img_size, seed=1, device=device):
            _, _, height, width = imgs.shape
            inf_out, train_out = model(imgs)
            output = non_max_suppression(inf_out, conf_thres=conf_thres, nms_thres=nms_thres)
            for si, pred in enumerate(output):
                labels = targets[targets[:, 0] == si, 1:].clone()
                nl = len(labels)
                tcls = labels[:, 0].tolist() if nl else []
                seen += 1
                if pred is None:
                    if nl:
                        stats.append(([], torch.Tensor(), torch.Tensor(), tcls))
                    continue
                if nl:
                    labels[:, [1, 3]] *= width
                    labels[:, [2, 4]] *= height
                correct = match_predictions(pred, labels, iou_thres)
                stats.append((correct, pred[:, 5].cpu(), pred[:, 7].cpu(), tcls))
    stats = [np.concatenate(x, 0) for x in list(zip(*stats))] if stats else []
    if len(stats):
        p, r, ap, f1, ap_class = ap_per_class(*stats)
        mp, mr, map_, mf1 = p.mean(), r.mean(), ap.mean(), f1.mean()
        nt = np.bincount(stats[3].astype(np.int64), minlength=nc)
    else:
        mp = mr = map_ = mf1 = 0.
        nt = np.zeros(1)
        ap, ap_class = [], []
    print(('%20s' + '%10s' * 6) % ('Class', 'Images', 'Targets', 'P', 'R', 'mAP', 'F1'))
    print(('%20s' + '%10.3g' * 6) % ('all', seen, nt.sum(), mp, mr, map_, mf1))
    maps = np.zeros(nc) + map_
    for i, c in enumerate(ap_class):
        maps[c] = ap[i]
    return (mp, mr, map_, mf1, 0., 0., 0.), maps


if __name__ == '__main__':
    from rotate_yolov3_amd.utils.cli import add_ignored, pick_device, report_ignored
    from rotate_yolov3_amd.utils.parse_config import parse_data_cfg
    parser = argparse.ArgumentParser(prog='test.py')                     # the reference's flags (test.py:206-216), same defaults
    parser.add_argument('--hyp', type=str, default='cfg/ICDAR/hyp.py', help='hyper-parameter path')
    parser.add_argument('--cfg', type=str, default='cfg/ICDAR/yolov3_608_se.cfg', help='cfg file path')
    parser.add_argument('--data', type=str, default='data/icdar_13+15.data', help='*.data file path (only `classes` is read: the image lists feed the out-of-scope OpenCV loader)')
    parser.add_argument('--weights', type=str, default='weights/best.pt', help='path to weights file (.pt or darknet .weights)')
    parser.add_argument('--batch-size', type=int, default=1, help='size of each image batch')
    parser.add_argument('--img-size', type=int, default=608, help='inference size (pixels)')
    parser.add_argument('--iou-thres', type=float, default=0.5, help='iou threshold required to qualify as detected')
    parser.add_argument('--conf-thres', type=float, default=0.001, help='object confidence threshold')
    parser.add_argument('--nms-thres', type=float, default=0.5, help='iou threshold for non-maximum suppression')
    parser.add_argument('--device', default='', help="device id (i.e. 0 or 0,1); the evaluation path has no CPU fallback")
    parser.add_argument('--synthetic', type=int, default=32, help='synthetic images to evaluate (this build has no image loader)')
    ignored = add_ignored(parser, [('--save-json', dict(action='store_true', help='save a cocoapi-compatible JSON results file'))])
    opt = parser.parse_args()
    print(opt)
    report_ignored(parser, opt, ignored)
    device = pick_device(opt.device)
    if device.type != 'cuda':
        sys.exit('test.py: the evaluation path (HIP forward, rotated NMS, rotated-IoU matching) needs a GPU; --device %r selects none' % opt.device)
    weights = opt.weights if os.path.isfile(opt.weights) else ''
    if opt.weights and not weights:
        print('NOTE: weights file %r not found: evaluating the freshly initialised model' % opt.weights)
    test(opt.cfg, hyp_parse(opt.hyp), weights, opt.batch_size, opt.img_size, opt.iou_thres, opt.conf_thres, opt.nms_thres,
         n_images=opt.synthetic, device=device,
         nc=int(parse_data_cfg(opt.data)['classes']) if os.path.isfile(opt.data) else None)
