"""CPU restatement of one Darknet `convolutional` block (+ the shortcut add / nearest upsample that consume
it) in the arithmetic contract of include/ryolo.h.  TEST INFRASTRUCTURE ONLY.

Follows the operator chain the reference builds and runs on its CPU path:
  nn.Conv2d(bias=not bn) -> nn.BatchNorm2d (eval: running stats) -> nn.PReLU(1)     model/models.py:49-66
  x + layer_outputs[from]                                                           model/models.py:281-282
  nn.Upsample(scale_factor=2, mode='nearest')                                       model/models.py:93-94
with the same ATen CPU operators (torch.nn.functional.conv2d in fp32).  The reference holds no test or golden
vector for these operators ("parity unpinned" at the cuDNN/ATen boundary); the pin is the
operator identity itself: this file calls the very operators the reference's model calls.
Inputs/weights are taken as bf16 values (exactly representable in fp32), accumulation in fp32, result rounded
to bf16 once after the activation and once more after the residual add.
"""
import torch
import torch.nn.functional as F


def bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float32)


def mish(x):
    return x * torch.tanh(F.softplus(x))


def conv_block(x_nchw, w_oihw, scale, shift, stride, pad, act="linear", slope=0.1, residual=None, upsample=1):
    """x_nchw, w_oihw, residual: fp32 tensors holding bf16-representable values.  Returns fp32 NCHW holding
    bf16-representable values."""
    y = F.conv2d(x_nchw.float(), w_oihw.float(), None, stride=stride, padding=pad)
    y = y * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if act == "leaky":
        y = torch.where(y > 0, y, y * slope)
    elif act == "mish":
        y = mish(y)
    y = bf16_round(y)
    if residual is not None:
        y = bf16_round(y + residual.float())
    if upsample != 1:
        y = F.interpolate(y, scale_factor=upsample, mode="nearest")
    return y
