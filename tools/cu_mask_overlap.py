"""Spatial partitioning experiment ("no design lets a streaming pass ride under an MFMA kernel").

The weight gradients (MFMA-bound, two 256-register workgroups per CU: nothing co-resides) are off the backward's dependency chain; the
BatchNorm backward passes (HBM-bound) are on it.  This script gives each its own HIP stream with a CU mask (hipExtStreamCreateWithCUMask:
S CUs for the streaming passes, 256 - S for the weight gradient; mask bit i -> XCD i % 8, so every XCD contributes S / 8 CUs) and times
  (1) the BatchNorm backward (reduce + finalize + apply) and forward passes alone on S CUs,
  (2) the 3x3 128->256 @76^2 weight-gradient tile kernel alone on 256 - S CUs,
  (3) both concurrently, work in the step's proportion (10.8 ms of weight gradients : 6.1 ms of BatchNorm backward), against the same work
      back to back on the whole chip.
python tools/cu_mask_overlap.py [S ...]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.model import hip_train_ops as tr  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
dev = torch.device("cuda:0")
torch.cuda.init()
torch.zeros(1, device=dev)


def masked_stream(lo, hi):
    words = (ctypes.c_uint32 * 8)()
    for b in range(lo, hi):
        words[b // 32] |= 1 << (b % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value, device=dev)


def wall(fn_list, reps=1):
    """fn_list: [(stream, callable)]: every callable enqueues its whole share on its stream; wall time until all streams drain"""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for st, fn in fn_list:
        with torch.cuda.stream(st):
            fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


bs, hw, c = 64, 76, 256
z = torch.randn(bs, hw, hw, c, device=dev).to(torch.bfloat16)
dy = torch.randn(bs, hw, hw, c, device=dev).to(torch.bfloat16)
y = torch.empty_like(z)
dzb = torch.empty_like(z)
scale = torch.rand(c, device=dev) + 0.5
shift = torch.randn(c, device=dev) * 0.1
mean = torch.randn(c, device=dev) * 0.1
invstd = torch.rand(c, device=dev) + 0.5
slope = torch.tensor([0.1], device=dev)
dg, db, dsl = torch.zeros(c, device=dev), torch.zeros(c, device=dev), torch.zeros(1, device=dev)
wsb = torch.empty(tr.bn_bwd_ws_bytes(bs * hw * hw, c), dtype=torch.uint8, device=dev)
el = bs * hw * hw * c

x = torch.randn(bs, hw, hw, 128, device=dev).to(torch.bfloat16)
dzw = torch.randn(bs, hw, hw, 256, device=dev).to(torch.bfloat16)
d = tr.make_desc(x, 256, 3, 1, 1)
wsw = torch.empty(tr.wgrad_ws_bytes(d), dtype=torch.uint8, device=dev)
gw = torch.zeros(256, 128, 3, 3, device=dev)
flop = 2.0 * 9 * 128 * 256 * hw * hw * bs


def bn_bwd(n):
    def f():
        for _ in range(n):
            tr.bn_act_bwd(z, dy, (mean, invstd, scale, shift), 1, slope, dzb, dg, db, dsl, wsb)
    return f


def bn_fwd(n):
    def f():
        for _ in range(n):
            tr.bn_act_fwd(z, scale, shift, 1, slope, y)
    return f


def wg(n):
    def f():
        for _ in range(n):
            tr.conv_wgrad_partials(d, x, dzw, 128, gw, True, wsw)
    return f


full = torch.cuda.current_stream()
NW, NB = 60, 16      # 60 tile launches (~11 ms) : 16 backward passes (~6 ms) on the whole chip
for _ in range(2):
    wall([(full, wg(5)), (full, bn_bwd(3)), (full, bn_fwd(3))])
t_w = min(wall([(full, wg(NW))]) for _ in range(3))
t_b = min(wall([(full, bn_bwd(NB))]) for _ in range(3))
t_f = min(wall([(full, bn_fwd(NB))]) for _ in range(3))
t_seq = min(wall([(full, wg(NW)), (full, bn_bwd(NB))]) for _ in range(3))
print("whole chip: wgrad %.1f us / launch (%.0f TF/s)   bn backward %.1f us (%.2f TB/s)   bn forward %.1f us (%.2f TB/s)" % (
    t_w / NW * 1e3, flop * NW / t_w / 1e9, t_b / NB * 1e3, el * 10 * NB / t_b / 1e9, t_f / NB * 1e3, el * 4 * NB / t_f / 1e9))
print("whole chip, back to back: %d wgrad + %d bn backward = %.2f ms" % (NW, NB, t_seq))
# two unmasked streams (what the side-stream experiment of round 5 did)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t_two = min(wall([(s1, wg(NW)), (s2, bn_bwd(NB))]) for _ in range(3))
print("two unmasked streams: %.2f ms" % t_two)

for S in [int(v) for v in sys.argv[1:]] or [32, 48, 64, 80, 96, 128]:
    sa, sb = masked_stream(0, S), masked_stream(S, 256)
    for _ in range(2):
        wall([(sb, wg(5)), (sa, bn_bwd(3)), (sa, bn_fwd(3))])
    a_b = min(wall([(sa, bn_bwd(NB))]) for _ in range(3))
    a_f = min(wall([(sa, bn_fwd(NB))]) for _ in range(3))
    a_w = min(wall([(sb, wg(NW))]) for _ in range(3))
    both = min(wall([(sb, wg(NW)), (sa, bn_bwd(NB))]) for _ in range(3))
    print("S = %3d CUs streaming / %3d MFMA: bn backward alone %.1f us (%.2f TB/s)  bn forward alone %.1f us (%.2f TB/s)  wgrad alone %.1f us "
          "(%.0f TF/s)  | concurrent %.2f ms vs back to back %.2f ms (%.2fx)" % (
              S, 256 - S, a_b / NB * 1e3, el * 10 * NB / a_b / 1e9, a_f / NB * 1e3, el * 4 * NB / a_f / 1e9, a_w / NW * 1e3,
              flop * NW / a_w / 1e9, both, t_seq, both / t_seq))
