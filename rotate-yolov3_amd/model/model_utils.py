"""Darknet binary weights I/O -- mirror of the reference's model/model_utils.py (load_darknet_weights :38-92,
save_weights :95-118).  File layout: int32 x3 version, int64 seen, then per convolutional block
bn_bias, bn_weight, bn_running_mean, bn_running_var (or conv_bias when the block has no BN), conv_weight, fp32."""
from pathlib import Path

import numpy as np
import torch


def load_darknet_weights(self, weights, cutoff=-1):
    file = Path(weights).name
    if file == 'darknet53.conv.74':
        cutoff = 75
    elif file == 'yolov3-tiny.conv.15':
        cutoff = 15
    with open(weights, 'rb') as f:
        self.version = np.fromfile(f, dtype=np.int32, count=3)
        self.seen = np.fromfile(f, dtype=np.int64, count=1)
        weights = np.fromfile(f, dtype=np.float32)
    ptr = 0

    def take(t):
        nonlocal ptr
        n = t.numel()
        t.data.copy_(torch.from_numpy(weights[ptr:ptr + n]).view_as(t))
        ptr += n

    for mdef, module in zip(self.module_defs[:cutoff], self.module_list[:cutoff]):
        if mdef['type'] == 'convolutional':
            conv_layer = module[0]
            if int(mdef['batch_normalize']):     # reference tests truthiness of the raw value (model_utils.py:60)
                bn_layer = module[1]
                take(bn_layer.bias)
                take(bn_layer.weight)
                take(bn_layer.running_mean)
                take(bn_layer.running_var)
            else:
                take(conv_layer.bias)
            take(conv_layer.weight)
    if hasattr(self, 'refresh_engines'):
        self.refresh_engines()
    return cutoff


def save_weights(self, path='model.weights', cutoff=-1):
    with open(path, 'wb') as f:
        self.version.tofile(f)
        self.seen.tofile(f)
        for mdef, module in zip(self.module_defs[:cutoff], self.module_list[:cutoff]):
            if mdef['type'] == 'convolutional':
                conv_layer = module[0]
                if int(mdef['batch_normalize']):
                    bn_layer = module[1]
                    bn_layer.bias.data.cpu().numpy().tofile(f)
                    bn_layer.weight.data.cpu().numpy().tofile(f)
                    bn_layer.running_mean.data.cpu().numpy().tofile(f)
                    bn_layer.running_var.data.cpu().numpy().tofile(f)
                else:
                    conv_layer.bias.data.cpu().numpy().tofile(f)
                conv_layer.weight.data.cpu().numpy().tofile(f)
