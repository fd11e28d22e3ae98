#!/usr/bin/env python
"""detect.py -- inference entry point mirroring the reference's detect.py (detect() :142-275, flags :279-296):
load cfg (+ .pt / darknet .weights), forward on the HIP engine, rotated NMS, write one text row per detection
(x y w h angle score class).  Image decoding / letterboxing / ICDAR writers use OpenCV in the reference and are
out of scope; the source here is a .pt/.npy tensor file [n,3,H,W] in [0,1] or `--synthetic N`."""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.model.model_utils import load_darknet_weights  # noqa: E402
from rotate_yolov3_amd.model.models import Darknet  # noqa: E402
from rotate_yolov3_amd.utils.parse_config import hyp_parse  # noqa: E402


def detect(opt):
    device = torch.device('cuda:0')
    hyp = hyp_parse(opt.hyp) if opt.hyp else {'context_factor': 1.0}
    model = Darknet(opt.cfg, hyp)
    if opt.weights.endswith('.pt'):
        model.load_state_dict(torch.load(opt.weights, map_location='cpu')['model'])
    elif opt.weights:
        load_darknet_weights(model, opt.weights)
    model.to(device).eval()
    if opt.source.endswith('.npy'):
        imgs = torch.from_numpy(np.load(opt.source)).float()
    elif opt.source.endswith('.pt'):
        imgs = torch.load(opt.source).float()
    else:
        imgs = torch.rand(opt.synthetic, 3, opt.img_size, opt.img_size, generator=torch.Generator().manual_seed(0))
    os.makedirs(opt.output, exist_ok=True)
    t0 = time.time()
    n_det = 0
    with torch.no_grad():
        for b in range(0, len(imgs), opt.batch_size):
            # forward + non_max_suppression(pred, conf, nms) of the reference (detect.py:91-99), fused on the GPU
            for i, det in enumerate(model.detect(imgs[b:b + opt.batch_size].to(device), opt.conf_thres, opt.nms_thres)):
                with open(os.path.join(opt.output, 'img_%d.txt' % (b + i)), 'w') as f:
                    if det is not None:
                        n_det += len(det)
                        for *box, conf, _, cls in det.cpu().tolist():
                            f.write(('%g ' * 7 + '\n') % (*box, conf, cls))
    print('Done. %d images, %d detections (%.3fs)' % (len(imgs), n_det, time.time() - t0))


if __name__ == '__main__':
    parser = argparse.ArgumentParser()
    parser.add_argument('--cfg', type=str, required=True)
    parser.add_argument('--hyp', type=str, default='')
    parser.add_argument('--weights', type=str, default='')
    parser.add_argument('--source', type=str, default='synthetic')
    parser.add_argument('--output', type=str, default='output')
    parser.add_argument('--img-size', type=int, default=608)
    parser.add_argument('--batch-size', type=int, default=8)
    parser.add_argument('--conf-thres', type=float, default=0.3)
    parser.add_argument('--nms-thres', type=float, default=0.3)
    parser.add_argument('--synthetic', type=int, default=8)
    detect(parser.parse_args())
