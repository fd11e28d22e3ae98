"""Darknet / YOLOLayer -- mirror of the reference's model/models.py (create_modules :36-164, YOLOLayer :170-227,
Darknet :230-313) and model/model_utils.py (create_grids :16-35, get_yolo_layers :10-11).

Same constructor, attributes and state_dict keys (`module_list.{i}.Conv2d.weight`, `.BatchNorm2d.*`,
`.activation.weight`), so reference checkpoints load.  What differs is what runs underneath:

  * eval-mode forward of a GPU tensor -> `HipEngine` (model/engine.py): the whole conv/BN/act/shortcut/route/
    upsample stack as hand-written MFMA kernels behind the C ABI, YOLO decode as a HIP kernel.  No fallback:
    if libryolo_hip.so is missing this raises.
  * training-mode forward of a GPU tensor -> `TrainEngine` (model/train_engine.py): forward with batch-statistics
    BatchNorm and the matching hand-written backward (dgrad / wgrad / BN+PReLU backward), wired into autograd.
  * a CPU tensor, or `model.backend = 'torch'` -> the same ATen operator chain the reference builds (nn.Conv2d,
    nn.BatchNorm2d, nn.PReLU, torch.cat, nn.Upsample), which is the reference's own CPU path.
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from ..utils.parse_config import parse_model_cfg, parse_model_cfg_text, yolo_mask

ARC_BIAS = {  # obj, cls bias targets per arc (models.py:133-146)
    'defaultpw': [-4, -3.6], 'Fdefaultpw': [-4, -3.6], 'default': [-5.5, -4.0], 'uBCE': [0, -8.5],
    'uCE': [10, -0.1], 'Fdefault': [-2.1, -1.8], 'uFBCE': [0, -6.5], 'uFBCEpw': [0, -6.5], 'uFCE': [7.7, -1.1]}


class Swish(nn.Module):
    def forward(self, x):
        return x * torch.sigmoid(x)


class Mish(nn.Module):
    """x * tanh(softplus(x)) -- named by the north star; the reference has no Mish."""

    def forward(self, x):
        return x * torch.tanh(F.softplus(x))


def create_modules(module_defs, arc, hyp):
    """cfg dicts -> nn.ModuleList + routes (models.py:36-164).  `module_defs` loses its [net] block, as in the
    reference; the popped block is returned third."""
    hyperparams = module_defs.pop(0)
    output_filters = [int(hyperparams['channels'])]
    module_list = nn.ModuleList()
    routes = []
    yolo_index = -1
    for i, mdef in enumerate(module_defs):
        modules = nn.Sequential()
        t = mdef['type']
        if t == 'convolutional':
            bn = int(mdef['batch_normalize'])
            filters = int(mdef['filters'])
            kernel_size = int(mdef['size'])
            pad = (kernel_size - 1) // 2 if int(mdef.get('pad', 0)) else 0
            # stock cfgs whose head width does not fit the rotated head: override with a notice (Appendix B.1)
            nxt = module_defs[i + 1] if i + 1 < len(module_defs) else None
            if nxt is not None and nxt['type'] == 'yolo':
                want = len(yolo_mask(nxt)) * (int(nxt['classes']) + 6)
                if filters != want:
                    print('NOTICE: layer %d filters=%d overridden to na*(nc+6)=%d for the rotated head' % (i, filters, want))
                    filters = want
                    mdef['filters'] = str(want)
            modules.add_module('Conv2d', nn.Conv2d(in_channels=output_filters[-1], out_channels=filters,
                                                   kernel_size=kernel_size, stride=int(mdef['stride']),
                                                   padding=pad, bias=not bn))
            if bn:
                modules.add_module('BatchNorm2d', nn.BatchNorm2d(filters, momentum=0.1))
            act = mdef.get('activation', 'linear')
            if act == 'leaky':
                modules.add_module('activation', nn.PReLU(num_parameters=1, init=0.10))   # models.py:63-66
            elif act == 'mish':
                modules.add_module('activation', Mish())
            elif act == 'swish':
                modules.add_module('activation', Swish())
        elif t == 'maxpool':
            kernel_size = int(mdef['size'])
            stride = int(mdef['stride'])
            maxpool = nn.MaxPool2d(kernel_size=kernel_size, stride=stride, padding=int((kernel_size - 1) // 2))
            if kernel_size == 2 and stride == 1:   # yolov3-tiny
                modules.add_module('ZeroPad2d', nn.ZeroPad2d((0, 1, 0, 1)))
                modules.add_module('MaxPool2d', maxpool)
            else:
                modules = maxpool
            filters = output_filters[-1]
        elif t == 'upsample':
            modules = nn.Upsample(scale_factor=int(mdef['stride']), mode='nearest')
            filters = output_filters[-1]
        elif t == 'route':
            layers = [int(x) for x in mdef['layers'].split(',')]
            filters = sum([output_filters[l + 1 if l > 0 else l] for l in layers])
            routes.extend([l if l > 0 else l + i for l in layers])
        elif t == 'shortcut':
            filters = output_filters[int(mdef['from'])]
            layer = int(mdef['from'])
            routes.extend([i + layer if layer < 0 else layer])
        elif t == 'reorg3d':
            filters = output_filters[-1]
        elif t == 'yolo':
            yolo_index += 1
            mask = yolo_mask(mdef)
            modules = YOLOLayer(anchors=mdef['anchors'][mask], nc=int(mdef['classes']), hyp=hyp,
                                yolo_index=yolo_index, arc=arc)
            # smart bias init of the preceding conv (models.py:132-155; the reference's version silently fails under
            # current autograd and leaves default biases -- done here under no_grad as intended)
            b = ARC_BIAS.get(arc)
            if b is not None:
                try:
                    with torch.no_grad():
                        conv = module_list[-1][0]
                        bias = conv.bias.view(len(mask), -1)
                        bias[:, 5] += b[0] - bias[:, 5].mean()
                        bias[:, 6:] += b[1] - bias[:, 6:].mean()
                except Exception:
                    print('WARNING: smart bias initialization failure.')
            filters = output_filters[-1]
        else:
            raise ValueError('Unsupported layer type in cfg: %r (se / d-convolutional / weight_from are out of scope, '
                             'SURVEY.md section 2)' % t)
        module_list.append(modules)
        output_filters.append(filters)
    return module_list, routes, hyperparams


def create_grids(self, img_size=416, ng=(13, 13), device='cpu', type=torch.float32):
    """grid offsets and per-stride anchors of one YOLO layer (model_utils.py:16-35).  Unlike the reference this never
    divides self.anchors in place (its CPU path shrinks the anchors on every grid rebuild)."""
    nx, ny = ng
    self.img_size = max(img_size)
    self.stride = self.img_size / max(ng)
    yv, xv = torch.meshgrid([torch.arange(ny), torch.arange(nx)], indexing='ij')
    self.grid_xy = torch.stack((xv, yv), 2).to(device).type(type).view((1, 1, ny, nx, 2))
    self.anchor_vec = self.anchors.clone().to(device)
    self.anchor_vec[:, :2] /= self.stride
    self.anchor_wh = self.anchor_vec.view(1, self.na, 1, 1, 3).to(device).type(type)
    self.ng = torch.Tensor(ng).to(device)
    self.nx = nx
    self.ny = ny


class YOLOLayer(nn.Module):
    def __init__(self, anchors, nc, yolo_index, arc, hyp):
        super(YOLOLayer, self).__init__()
        self.anchors = torch.Tensor(np.asarray(anchors))
        self.na = len(anchors)
        self.nc = nc
        self.nx = 0
        self.ny = 0
        self.arc = arc
        self.hyp = hyp if hyp is not None else {'context_factor': 1.0}
        self.yolo_index = yolo_index

    def forward(self, p, img_size, var=None):
        bs, ny, nx = p.shape[0], p.shape[-2], p.shape[-1]
        if (self.nx, self.ny) != (nx, ny) or self.grid_xy.device != p.device:
            create_grids(self, img_size, (nx, ny), p.device, p.dtype)
        p = p.view(bs, self.na, self.nc + 6, self.ny, self.nx).permute(0, 1, 3, 4, 2).contiguous()
        if self.training:
            return p
        cf = float(self.hyp.get('context_factor', 1.0))
        io = p.clone()
        io[..., 0:2] = torch.sigmoid(io[..., 0:2]) + self.grid_xy
        io[..., 2:4] = torch.exp(io[..., 2:4]) * self.anchor_wh[..., :-1]
        io[..., 4] = torch.atan(io[..., 4]) + self.anchor_wh[..., -1]
        io[..., :4] *= self.stride
        io[..., 3] /= cf
        io[..., 2] -= io[..., 3] * (cf - 1)
        if 'default' in self.arc:
            torch.sigmoid_(io[..., 5:])
        elif 'BCE' in self.arc:
            torch.sigmoid_(io[..., 6:])
            io[..., 5] = 1
        elif 'CE' in self.arc:
            io[..., 6:] = F.softmax(io[..., 5:], dim=4)[..., 1:]
            io[..., 5] = 1
        if self.nc == 1:
            io[..., 6] = 1
        return io.view(bs, -1, 6 + self.nc), p


def get_yolo_layers(model):
    return [i for i, x in enumerate(model.module_defs) if x['type'] == 'yolo']


class Darknet(nn.Module):
    """YOLOv3 rotated-box detector built from a Darknet cfg (path, or cfg text containing a newline)."""

    def __init__(self, cfg, hyp=None, arc='default'):
        super(Darknet, self).__init__()
        if isinstance(cfg, str) and '\n' in cfg:
            self.module_defs = parse_model_cfg_text(cfg)
        else:
            self.module_defs = parse_model_cfg(cfg)
        hyp = hyp if hyp is not None else {'context_factor': 1.0}
        self.module_list, self.routes, self.net_params = create_modules(self.module_defs, arc, hyp)
        self.yolo_layers = get_yolo_layers(self)
        self.hyp = hyp
        self.arc = arc
        self.nc = int(self.module_defs[self.yolo_layers[0]]['classes']) if self.yolo_layers else 0
        self.version = np.array([0, 2, 5], dtype=np.int32)
        self.seen = np.array([0], dtype=np.int64)
        self.backend = 'hip'      # 'hip' (eval + GPU) | 'torch' (explicit request for the ATen chain on any device)
        self._engines = {}

    # ------------------------------------------------------------------ ATen operator chain (reference CPU path)
    def _torch_forward(self, x):
        img_size = x.shape[-2:]
        layer_outputs = []
        output = []
        for i, (mdef, module) in enumerate(zip(self.module_defs, self.module_list)):
            mtype = mdef['type']
            if mtype in ['convolutional', 'upsample', 'maxpool']:
                x = module(x)
            elif mtype == 'route':
                layers = [int(v) for v in mdef['layers'].split(',')]
                if len(layers) == 1:
                    x = layer_outputs[layers[0]]
                else:
                    x = torch.cat([layer_outputs[j] for j in layers], 1)
            elif mtype == 'shortcut':
                x = x + layer_outputs[int(mdef['from'])]
            elif mtype == 'yolo':
                x = module(x, img_size)
                output.append(x)
            layer_outputs.append(x if i in self.routes else [])
        if self.training:
            return output
        io, p = list(zip(*output))
        return torch.cat(io, 1), p

    # ------------------------------------------------------------------ HIP engine (eval, GPU)
    def engine(self, x_shape, device):
        from .engine import HipEngine
        key = (tuple(x_shape), device.index)
        eng = self._engines.get(key)
        if eng is None:
            eng = HipEngine(self, x_shape, device)
            self._engines[key] = eng
        return eng

    def detect(self, x, conf_thres=0.5, nms_thres=0.5):
        """Inference step = forward + the reference's non_max_suppression (utils/nms/nms.py:4-69), same return value.
        On the GPU (eval mode, HIP backend) the yolo layers run the fused decode + confidence filter + compaction kernel
        and every (image, class) set goes through one segmented rotated-NMS launch (HipEngine.detect); elsewhere it is
        literally non_max_suppression(self(x)[0], ...)."""
        from ..utils.nms.nms import non_max_suppression
        if self.backend == 'torch' or not x.is_cuda or self.training:
            return non_max_suppression(self(x)[0], conf_thres, nms_thres)
        self._drop_stale_eval_engines()
        return self.engine(x.shape, x.device).detect(x, conf_thres, nms_thres)

    def _drop_stale_eval_engines(self):
        """The eval engines hold packed bf16 copies of the weights and the folded BatchNorm: rebuilt when anything changed since
        they were packed -- a HIP training forward (it updates the running statistics through raw pointers), an optimizer step
        (FusedSGD bumps the versions of the tensors it writes behind autograd's back), EMA or manual in-place edits (tensor
        versions).  The TrainEngine entries (their hipGraphs and buffers read the live parameters) are kept."""
        sig = self._param_signature()
        if getattr(self, '_eval_engines_stale', False) or sig != getattr(self, '_eval_sig', sig):
            self._engines = {k: v for k, v in self._engines.items() if k and k[0] == 'train'}
            self._eval_engines_stale = False
        self._eval_sig = sig

    def refresh_engines(self):
        """Call after the parameter TENSORS were replaced (load_state_dict, fuse): every engine is rebuilt.  In-place updates need
        no call: training forwards, optimizer steps, EMA and manual edits are noticed by the next eval forward
        (_drop_stale_eval_engines)."""
        self._engines = {}
        self._eval_engines_stale = False

    def load_state_dict(self, *a, **k):
        r = super(Darknet, self).load_state_dict(*a, **k)
        self.refresh_engines()
        return r

    def enable_fused_loss(self, capacity=512, impl='hip'):
        """loss.compute_loss on heads of the HIP TrainEngine becomes one hipGraph replay (model/loss_static.py): padded
        targets (up to `capacity` per step; more falls back to the eager mirror) in, loss items and head gradients out,
        no host synchronisation.  Same numbers as the eager mirror to fp32 summation order."""
        from .loss_static import FusedLoss
        self.fused_loss = FusedLoss(self, capacity, impl)
        return self

    def _param_signature(self):
        """(sum of the version counters, hash of the storage addresses) of every parameter and buffer.  The first changes with every
        in-place update (optimizer steps, EMA, manual edits); the second when a tensor is REPLACED -- `p.data = t`, `.to()`, `.half()`,
        load_state_dict(assign=True) -- where a fresh tensor's version counter starts again and the sum alone could coincide (ADVICE
        r3).  A kernel that writes parameters through raw pointers must bump the versions (FusedSGD and the HIP training forward do) or
        call refresh_engines().  ~60 us for Darknet-53's 366 tensors."""
        ts = list(self.parameters()) + list(self.buffers())
        return (sum(t._version for t in ts), hash(tuple(t.data_ptr() for t in ts)))

    def train_engine(self, x_shape, device):
        from .train_engine import TrainEngine
        key = ('train', tuple(x_shape), device.index)
        eng = self._engines.get(key)
        if eng is None:
            eng = TrainEngine(self, x_shape, device)
            self._engines[key] = eng
        return eng

    def forward(self, x, var=None):
        """models.py:262-298.  On the GPU (backend 'hip'): eval returns (io, p) as fresh tensors unless
        `self.zero_copy_outputs = True` (then they alias engine-owned buffers that the next forward of the same input shape
        overwrites); training returns the head tensors p of the TrainEngine, which ARE engine buffers (1 GB at bs 64: not
        copied) -- consume them (compute_loss + backward) before the next forward."""
        if self.backend == 'torch' or not x.is_cuda:
            return self._torch_forward(x)
        if self.training:
            # hand-written HIP forward + backward (conv/BN/PReLU/shortcut/route/upsample); the returned head tensors
            # are outputs of an autograd Function whose backward runs the HIP backward and fills param.grad
            self._eval_engines_stale = True      # an optimizer step follows: the eval engines' packed copies go stale
            return self.train_engine(x.shape, x.device)(x)
        self._drop_stale_eval_engines()
        out = self.engine(x.shape, x.device)(x)
        if getattr(self, 'zero_copy_outputs', False):
            return out       # (io, p) ARE the engine's buffers: the next forward of this input shape overwrites them
        # ATen semantics by default: the caller may keep the result across batches (e.g. collecting inf_out, test.py:118-146)
        io, p = out
        return io.clone(), [q.clone() for q in p]

    def fuse(self):
        """Conv+BN folding (models.py:300-313 / utils/torch_utils.py:45-69).  The HIP engine always applies the
        folded scale/shift in its conv epilogue, so there is nothing to do for it; the ATen chain is folded here."""
        from ..utils.torch_utils import fuse_conv_and_bn
        fused_list = nn.ModuleList()
        for a in list(self.children())[0]:
            if isinstance(a, nn.Sequential):
                for i, b in enumerate(a):
                    if isinstance(b, nn.modules.batchnorm.BatchNorm2d):
                        conv = a[i - 1]
                        fused = fuse_conv_and_bn(conv, b)
                        a = nn.Sequential(fused, *list(a.children())[i + 1:])
                        break
            fused_list.append(a)
        self.module_list = fused_list
        self.refresh_engines()
