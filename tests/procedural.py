"""Procedural (file-free) weights for parity fixtures: every parameter/buffer is a deterministic function of its
state_dict position and element index, so goldens can be regenerated and checked without shipping a checkpoint."""
import math

import torch


def fill_procedural(model):
    sd = model.state_dict()
    with torch.no_grad():
        for li, (name, t) in enumerate(sd.items()):
            if t.dtype not in (torch.float32, torch.float64, torch.float16, torch.bfloat16):
                continue          # num_batches_tracked
            i = torch.arange(t.numel(), dtype=torch.float64)
            s = torch.sin(0.37 * i + 1.3 * li)
            if name.endswith("Conv2d.weight") or (name.endswith(".weight") and t.dim() == 4):
                fan_in = t[0].numel()
                v = s * math.sqrt(3.0 / fan_in)
            elif "running_var" in name:
                v = 1.0 + 0.1 * s
            elif "running_mean" in name:
                v = 0.02 * s
            elif "BatchNorm2d.weight" in name:
                v = 1.0 + 0.1 * s
            elif "activation.weight" in name:
                v = 0.1 + 0.02 * s
            else:                  # biases
                v = 0.05 * s
            t.copy_(v.view_as(t).to(t.dtype))
    return model
