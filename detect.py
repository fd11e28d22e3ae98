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
    device = getattr(opt, 'torch_device', None) or torch.device('cuda:0')
    hyp = hyp_parse(opt.hyp) if opt.hyp and os.path.isfile(opt.hyp) else {'context_factor': 1.0}
    model = Darknet(opt.cfg, hyp)
    if opt.weights and not os.path.isfile(opt.weights):
        print('NOTE: weights file %r not found: running the freshly initialised model' % opt.weights)
    elif opt.weights.endswith('.pt'):
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
    from rotate_yolov3_amd.utils.cli import add_ignored, pick_device, report_ignored
    parser = argparse.ArgumentParser()                                   # the reference's flags (detect.py:280-295), same defaults
    parser.add_argument('--hyp', type=str, default='cfg/ICDAR/hyp.py', help='hyper-parameter path')
    parser.add_argument('--cfg', type=str, default='cfg/ICDAR/yolov3_608_dh_o8_ga.cfg', help='cfg file path')
    parser.add_argument('--weights', type=str, default='weights/last.pt', help='path to weights file (.pt or darknet .weights)')
    parser.add_argument('--source', type=str, default='data/tiny/test', help='a .npy / .pt tensor file [n,3,H,W] in [0,1]; anything else: --synthetic random images (image decoding is out of scope)')
    parser.add_argument('--output', type=str, default='output', help='output folder')
    parser.add_argument('--img-size', type=int, default=608, help='inference size (pixels)')
    parser.add_argument('--conf-thres', type=float, default=0.5, help='object confidence threshold')
    parser.add_argument('--nms-thres', type=float, default=0.3, help='iou threshold for non-maximum suppression')
    parser.add_argument('--device', default='', help='device id (i.e. 0 or 0,1); the detection path has no CPU fallback')
    parser.add_argument('--batch-size', type=int, default=8, help='images per forward')
    parser.add_argument('--synthetic', type=int, default=8, help='random images when --source is not a tensor file')
    ignored = add_ignored(parser, [
        ('--data', dict(type=str, default='data/tiny.data', help='*.data file path (class names for the image writers)')),
        ('--fourcc', dict(type=str, default='mp4v', help='output video codec')),
        ('--half', dict(action='store_true', help='half precision FP16 inference (the HIP engine always computes in bf16 with fp32 accumulation)')),
        ('--view-img', dict(action='store_true', help='display results')),
        ('--multi-scale', dict(action='store_true', help='multi-scale testing'))])
    opt = parser.parse_args()
    print(opt)
    report_ignored(parser, opt, ignored)
    opt.torch_device = pick_device(opt.device)
    if opt.torch_device.type != 'cuda':
        sys.exit('detect.py: the detection path (HIP forward + rotated NMS) needs a GPU; --device %r selects none' % opt.device)
    detect(opt)
