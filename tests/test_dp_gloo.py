"""CPU tier, world_size = 2 over gloo: the data-parallel gradient exchange (rotate-yolov3_amd/dist.py).
Check: the averaged gradients of 2 ranks x B images equal a single-process pass over the
same 2B images with BatchNorm evaluated per B-image shard."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CFG = """
[net]
width=64
height=64
channels=3

[convolutional]
batch_normalize=1
filters=8
size=3
stride=2
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=3
stride=2
pad=1
activation=leaky

[convolutional]
batch_normalize=1
filters=16
size=1
stride=1
pad=1
activation=leaky

[shortcut]
from=-2
activation=linear

[convolutional]
size=1
stride=1
pad=1
filters=84
activation=linear

[yolo]
mask = 0-11
anchors = 20,8, 60,20
classes=1
num=2

[route]
layers = -3

[convolutional]
batch_normalize=1
filters=16
size=3
stride=2
pad=1
activation=leaky

[convolutional]
size=1
stride=1
pad=1
filters=84
activation=linear

[yolo]
mask = 12-23
anchors = 20,8, 60,20
classes=1
num=2
"""
HYP = {"giou": 0.1, "cls": 27.76, "cls_pw": 1.446, "obj": 20.35, "obj_pw": 3.941, "iou_t": 0.3, "ang_t": 0.2618,
       "reg": 1.0, "fl_gamma": 0.5, "context_factor": 1.0}


def _batch(rank):
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand(3, 3, 64, 64, generator=g)
    t = torch.tensor([[0, 0, 0.3, 0.4, 0.30, 0.10, 0.2], [1, 0, 0.6, 0.5, 0.45, 0.15, -0.8], [2, 0, 0.5, 0.7, 0.25, 0.08, 1.2]])
    t[:, 2:4] += 0.05 * rank
    return x, t


def _model():
    sys.path.insert(0, ROOT)
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.model.models import Darknet
    torch.manual_seed(7)
    m = Darknet(CFG, dict(HYP), arc="default")
    m.train()
    return m


def _step(m, x, t):
    from rotate_yolov3_amd.model.loss import compute_loss
    loss, items = compute_loss(m(x), t.clone(), m, m.hyp)
    loss.backward()
    return float(loss.detach())


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sys.path.insert(0, ROOT)
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.dist import GradientAllReducer
    m = _model()
    if rank == 1:          # a replica that starts from different weights must be overwritten by rank 0's
        with torch.no_grad():
            for p in m.parameters():
                p.add_(1.0)
    dp = GradientAllReducer(m, bucket_mb=0.01)           # tiny buckets -> several collectives in flight
    x, t = _batch(rank)
    _step(m, x, t)
    dp.finish()
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in m.parameters()], "nb": len(dp.buckets),
                    "bn_mean": m.module_list[0][1].running_mean.clone()}, out)
    # second step: re-armed hooks, accumulated on zeroed buckets
    dp.zero_grad()
    _step(m, x, t)
    dp.finish()
    g2 = [p.grad.clone() for p in m.parameters()]
    if rank == 0:
        d = torch.load(out)
        d["grads2"] = g2
        torch.save(d, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_per_shard_single_process(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    d = torch.load(out)
    assert d["nb"] > 2
    # single process: same initial weights (rank 0's), each shard through the model separately (per-shard BN stats)
    torch.set_num_threads(2)
    m = _model()
    for r in range(2):
        x, t = _batch(r)
        _step(m, x, t)
    ref = [p.grad / 2 for p in m.parameters()]
    for a, b in zip(d["grads"], ref):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), (a - b).abs().max()
    # step 2 on the same data with unchanged weights reproduces step 1 except for BN running-stat-independent terms
    for a, b in zip(d["grads2"], d["grads"]):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6)


def _worker_accum(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sys.path.insert(0, ROOT)
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.dist import GradientAllReducer
    m = _model()
    dp = GradientAllReducer(m, bucket_mb=0.01)
    # accumulate = 2: micro-batch A without a collective, micro-batch B with it (train.py: dp.sync = do_step)
    xa, ta = _batch(rank)
    xb, tb = _batch(rank + 2)
    dp.sync = False
    _step(m, xa, ta)
    assert all(b["handle"] is None for b in dp.buckets)          # nothing launched by the first micro-batch
    dp.sync = True
    _step(m, xb, tb)
    dp.finish()
    if rank == 0:
        torch.save({"grads": [p.grad.clone() for p in m.parameters()]}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_accumulate_two_micro_batches(tmp_path):
    """with --accumulate > 1 the bucket counters used to fire on the first micro-batch and never again."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dpa.pt")
    mp.spawn(_worker_accum, args=(2, port, out), nprocs=2, join=True)
    d = torch.load(out)
    torch.set_num_threads(2)
    m = _model()
    for r in (0, 2, 1, 3):                 # rank 0: batches 0 and 2; rank 1: batches 1 and 3; sum over everything / world
        x, t = _batch(r)
        _step(m, x, t)
    ref = [p.grad / 2 for p in m.parameters()]
    for a, b in zip(d["grads"], ref):
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-6), (a - b).abs().max()


def _worker_wire(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    sys.path.insert(0, ROOT)
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.dist import GradientAllReducer
    res = {}
    for name, wire, in_opt in (("fp32", None, False), ("bf16", torch.bfloat16, False), ("bf16_opt", torch.bfloat16, True)):
        m = _model()
        dp = GradientAllReducer(m, bucket_mb=0.01, wire_dtype=wire)
        dp.scale_in_optimizer = in_opt            # train.py: FusedSGD applies 1/world -> finish() must hand it the SUM
        x, t = _batch(rank)
        _step(m, x, t)
        dp.finish()
        res[name] = [p.grad.clone() for p in m.parameters()]
        dp.detach()
        assert getattr(m, "_dp_grad_views", None) is None and not dp._hooks
    if rank == 0:
        torch.save(res, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bf16_wire_matches_fp32_wire(tmp_path):
    """wire_dtype = bf16 sends each rank's share of the mean and lets the collective add the shares in bf16: the averaged
    gradient must equal the fp32-wire one to bf16 rounding of its addends (2^-8 relative per addend), with or without the 1/world
    being applied by the optimizer; detach() gives the model back."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dpw.pt")
    mp.spawn(_worker_wire, args=(2, port, out), nprocs=2, join=True)
    d = torch.load(out)
    for a, b, c in zip(d["fp32"], d["bf16"], d["bf16_opt"]):
        # (the two per-rank addends can be larger than their mean: the bar is relative to the tensor's largest entry)
        tol = 2.0 ** -7 * (a.abs() + float(a.abs().max())) + 1e-9
        assert bool(((a - b).abs() <= tol).all()), float((a - b).abs().max())
        assert bool(((2 * a - c).abs() <= 2 * tol).all()), float((2 * a - c).abs().max())      # the SUM, for an optimizer that scales by 1/world
