"""FusedSGD -- torch.optim.SGD (momentum, nesterov, per-group weight decay: what the reference's train.py:70-83 builds)
with the whole step as ONE HIP launch (csrc/optim.hip) instead of ATen's ~30 foreach kernels over the 222 parameter
tensors of Darknet-53.  Same arithmetic in the same order; same `param_groups` / `state[p]['momentum_buffer']` layout, so
`state_dict()` is interchangeable with torch.optim.SGD's.  Parameters that are not fp32 CUDA tensors, dampening,
maximize and differentiable are not supported (construct torch.optim.SGD for those)."""
import ctypes as C

import torch

from .. import _lib


class _Job(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("buf", C.c_void_p), ("n", C.c_longlong), ("group", C.c_int),
                ("first", C.c_int), ("block_begin", C.c_int), ("block_end", C.c_int)]


_lib.declare("ryolo_sgd_step", C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p])


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr, momentum=0.0, weight_decay=0.0, nesterov=False):
        if nesterov and momentum <= 0:
            raise ValueError("Nesterov momentum requires a momentum")
        super(FusedSGD, self).__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay, nesterov=nesterov,
                                                    dampening=0, maximize=False, foreach=None, differentiable=False,
                                                    fused=None))
        self._key = None
        self.grad_scale = 1.0        # data parallel: 1 / world (the all-reduce delivers the SUM; dist.py then skips its div_)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        entries, nesterov = [], None
        for gi, group in enumerate(self.param_groups):
            nesterov = bool(group['nesterov']) if nesterov is None else nesterov
            if bool(group['nesterov']) != nesterov:
                raise RuntimeError("FusedSGD: `nesterov` must be the same in every param group")
            for p in group['params']:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.grad.dtype == torch.float32 and p.is_contiguous()
                        and p.grad.is_contiguous()):
                    raise RuntimeError("FusedSGD handles contiguous fp32 CUDA parameters and gradients")
                st = self.state[p]
                first = 'momentum_buffer' not in st or st['momentum_buffer'] is None
                if first:
                    st['momentum_buffer'] = torch.empty_like(p)
                entries.append((p, p.grad, st['momentum_buffer'], gi, first))
        if not entries:
            return loss
        dev = entries[0][0].device
        key = tuple((p.data_ptr(), g.data_ptr(), b.data_ptr(), gi, first) for p, g, b, gi, first in entries)
        if key != self._key:                       # pointer table changed (first step, new gradient tensors): rebuild + upload
            arr = (_Job * len(entries))()
            blk = 0
            for q, (p, g, b, gi, first) in enumerate(entries):
                nb = max(1, min(1024, (p.numel() + 2047) // 2048))
                arr[q] = _Job(p.data_ptr(), g.data_ptr(), b.data_ptr(), p.numel(), gi, 1 if first else 0, blk, blk + nb)
                blk += nb
            self._jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self._n, self._blocks, self._key = len(entries), blk, key
        gs = 0.0 if self.grad_scale == 1.0 else float(self.grad_scale)
        hp = torch.tensor([[g['lr'], g['momentum'], g['weight_decay'], gs] for g in self.param_groups], dtype=torch.float32).to(dev)
        with torch.cuda.device(dev):
            _lib.check(_lib.lib().ryolo_sgd_step(self._jobs.data_ptr(), self._n, self._blocks, hp.data_ptr(), 1 if nesterov else 0,
                                                 _lib.stream_ptr(dev)), "ryolo_sgd_step")
        # the kernel wrote the parameters through raw pointers: tell autograd (and everything that watches tensor versions --
        # Darknet's cached eval engines hold packed copies of the weights and compare versions before they are reused)
        torch.autograd.graph.increment_version([p for p, _, _, _, _ in entries])
        return loss
