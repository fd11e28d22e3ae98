"""Darknet binary weights I/O -- mirror of the reference's model/model_utils.py (load_darknet_weights :38-92,
save_weights :95-118).  File layout: int32 x3 version, int64 seen, then per convolutional block
bn_bias, bn_weight, bn_running_mean, bn_running_var (or conv_bias when the block has no BN), conv_weight, fp32."""
from pathlib import Path

import numpy as np
import torch


_STEM_CUTOFFS = {'darknet53.conv.74': 75, 'yolov3-tiny.conv.15': 15}     # backbone-only files (model_utils.py:41-47)


def _file_order(model, cutoff):
    """The tensors of the first `cutoff` blocks in the order the darknet file stores them: per convolutional block the four
    BatchNorm vectors (bias, weight, running mean, running variance) or, without BatchNorm, the conv bias; then the filter."""
    for mdef, module in zip(model.module_defs[:cutoff], model.module_list[:cutoff]):
        if mdef['type'] != 'convolutional':
            continue
        conv = module[0]
        if int(mdef['batch_normalize']):     # the reference tests the truthiness of the raw value (model_utils.py:60)
            bn = module[1]
            yield from (bn.bias, bn.weight, bn.running_mean, bn.running_var)
        else:
            yield conv.bias
        yield conv.weight


def load_darknet_weights(self, weights, cutoff=-1):
    cutoff = _STEM_CUTOFFS.get(Path(weights).name, cutoff)
    with open(weights, 'rb') as f:
        self.version = np.fromfile(f, dtype=np.int32, count=3)
        self.seen = np.fromfile(f, dtype=np.int64, count=1)
        flat = torch.from_numpy(np.fromfile(f, dtype=np.float32))
    pos = 0
    for t in _file_order(self, cutoff):
        n = t.numel()
        t.data.copy_(flat[pos:pos + n].view_as(t))
        pos += n
    if hasattr(self, 'refresh_engines'):
        self.refresh_engines()
    return cutoff


def save_weights(self, path='model.weights', cutoff=-1):
    body = [t.detach().float().cpu().reshape(-1) for t in _file_order(self, cutoff)]
    with open(path, 'wb') as f:
        self.version.tofile(f)
        self.seen.tofile(f)
        if body:
            torch.cat(body).numpy().tofile(f)
