#!/usr/bin/env python
"""train.py -- entry point mirroring the reference's train.py (train() :34-377, flags :383-404).

Kept: cfg / data / hyp files, Darknet(cfg, hyp, arc), the two optimizer parameter groups (weight decay only on
`Conv2d.weight`, train.py:70-83), SGD-nesterov / Adam, MultiStepLR(0.8, 0.9 of the epochs) behind a linear warm-up
(GradualWarmupScheduler semantics: lr0 -> lr0*multiplier over warm_epoch epochs, train.py:143-152), gradient
accumulation, the non-finite-loss abort, results.txt rows and the checkpoint dict (train.py:323-363).
Changed: data parallelism is ONE PROCESS PER GPU (launch with `python -m torch.distributed.run --nproc-per-node N
train.py ...`) with bucketed RCCL all-reduce overlapped with backward (rotate-yolov3_amd/dist.py) instead of the
reference's broken single-process DDP; the input pipeline is synthetic (`--synthetic N` images per epoch) because the
OpenCV/imgaug loader is out of scope.  On a GPU the training-mode forward AND backward run the hand-written HIP
TrainEngine (conv forward / dgrad / wgrad on MFMA, BatchNorm + PReLU kernels, fused loss, one-launch SGD;
rotate-yolov3_amd/model/train_engine.py); CPU tensors and `model.backend = 'torch'` take the ATen operator chain.
Per-epoch evaluation follows the reference's gate (train.py:306-314: test.test(model=model) every `test_interval`
epochs from epoch 10 on, always considered at the final epoch unless --notest), `last.pt` is written every epoch and
`best.pt` when the mAP fitness does not decrease (train.py:345-357).
"""
import argparse
import math
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.optim as optim

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd.dist import GradientAllReducer  # noqa: E402
from rotate_yolov3_amd.model.loss import compute_loss  # noqa: E402
from rotate_yolov3_amd.model.models import Darknet  # noqa: E402
from rotate_yolov3_amd.utils.parse_config import hyp_parse, parse_data_cfg  # noqa: E402
from rotate_yolov3_amd.utils.synthetic import SyntheticLoader  # noqa: E402
from rotate_yolov3_amd.utils.torch_utils import init_seeds  # noqa: E402

results_file = 'results.txt'


def make_optimizer(model, hyp, adam=False, fused=True):
    pg0, pg1 = [], []
    for k, v in dict(model.named_parameters()).items():
        (pg1 if 'Conv2d.weight' in k else pg0).append(v)
    if adam:
        optimizer = optim.Adam(pg0, lr=hyp['lr0'])
    elif fused and all(p.is_cuda and p.dtype == torch.float32 for p in pg0 + pg1):
        from rotate_yolov3_amd.utils.fused_sgd import FusedSGD      # same arithmetic, one launch for all 222 tensors
        optimizer = FusedSGD(pg0, lr=hyp['lr0'], momentum=hyp['momentum'], nesterov=True)
    else:
        optimizer = optim.SGD(pg0, lr=hyp['lr0'], momentum=hyp['momentum'], nesterov=True)
    optimizer.add_param_group({'params': pg1, 'weight_decay': hyp['weight_decay']})
    return optimizer


def lr_factor(epoch, epochs, multiplier, warm_epoch):
    """GradualWarmupScheduler(multiplier, total_epoch=warm_epoch, after=MultiStepLR([.8,.9]*epochs, 0.1))."""
    if warm_epoch > 0 and epoch <= warm_epoch:
        return (multiplier - 1.0) * epoch / warm_epoch + 1.0
    e = epoch - warm_epoch
    f = multiplier
    for m in (round(epochs * 0.8), round(epochs * 0.9)):
        if e >= m:
            f *= 0.1
    return f


def init_schedule(optimizer, hyp):
    """The schedule multiplies lr0 -- never the lr a restored optimizer state carries (that one was already scaled by the
    factor of the epoch it was saved in; the reference's schedulers keep `initial_lr` for the same reason)."""
    for g in optimizer.param_groups:
        g['initial_lr'] = float(hyp['lr0'])


def set_epoch_lr(optimizer, hyp, epoch, epochs):
    f = lr_factor(epoch, epochs, float(hyp.get('multiplier', 1.0)), float(hyp.get('warm_epoch', 0)))
    for g in optimizer.param_groups:
        g['lr'] = g['initial_lr'] * f
    return f


def train(opt, hyp):
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    from rotate_yolov3_amd.utils.cli import pick_device
    device = pick_device(opt.device, local_rank)
    use_cuda = device.type == 'cuda'
    if use_cuda:
        torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='nccl' if use_cuda else 'gloo', rank=rank, world_size=world)
    if 'pw' not in opt.arc:
        hyp['cls_pw'] = 1.
        hyp['obj_pw'] = 1.
    epochs = int(opt.epochs or hyp['epochs'])
    batch_size = int(opt.batch_size or hyp['batch_size'])
    init_seeds()
    nc = int(parse_data_cfg(opt.data)['classes']) if opt.data and os.path.isfile(opt.data) else 1

    model = Darknet(opt.cfg, hyp, arc=opt.arc).to(device)
    model.nc, model.arc, model.hyp = nc, opt.arc, hyp
    if use_cuda and not opt.eager_loss:
        model.enable_fused_loss(capacity=max(256, 8 * batch_size))   # compute_loss as one hipGraph replay (loss_static.py)
    optimizer = make_optimizer(model, hyp, opt.adam)
    start_epoch, best_fitness = 0, 0.
    if opt.resume and not os.path.isfile(opt.weights):
        # the reference fails in torch.load here (train.py:406 + :104); silently starting at epoch 0 would overwrite last.pt / best.pt / results.txt
        raise FileNotFoundError("--resume: no checkpoint at %r (check --wdir)" % opt.weights)
    if opt.weights and not opt.weights.endswith('.pt'):
        if os.path.isfile(opt.weights):     # darknet format (*.weights, darknet53.conv.74): reference train.py:115-117
            from rotate_yolov3_amd.model.model_utils import load_darknet_weights
            load_darknet_weights(model, opt.weights)
            model.refresh_engines()
        elif rank == 0:
            print('NOTE: weights file %r not found: training the freshly initialised model' % opt.weights)
    elif opt.weights and not os.path.isfile(opt.weights) and rank == 0:
        print('NOTE: weights file %r not found: training the freshly initialised model' % opt.weights)
    if opt.weights and opt.weights.endswith('.pt') and os.path.isfile(opt.weights):
        chkpt = torch.load(opt.weights, map_location=device)
        sd = {k: v for k, v in chkpt['model'].items() if k in model.state_dict() and model.state_dict()[k].numel() == v.numel()}
        model.load_state_dict(sd, strict=False)
        if chkpt.get('optimizer') is not None:
            optimizer.load_state_dict(chkpt['optimizer'])
            best_fitness = chkpt['best_fitness']
        if opt.resume:
            start_epoch = chkpt['epoch'] + 1
    dp = GradientAllReducer(model, bucket_mb=opt.bucket_mb)
    init_schedule(optimizer, hyp)         # after load_state_dict: base lr = lr0, not the saved (already scaled) lr
    if world > 1 and hasattr(optimizer, 'grad_scale'):
        # the all-reduce delivers the SUM over ranks; FusedSGD applies 1/world while it reads the gradient (no extra 250 MB pass)
        optimizer.grad_scale = 1.0 / world
        dp.scale_in_optimizer = True
    loader = SyntheticLoader(opt.synthetic, batch_size, opt.img_size, seed=rank, device=device)
    nb = len(loader)
    results = (0, 0, 0, 0, 0, 0, 0)
    t0 = time.time()
    for epoch in range(start_epoch, epochs):
        model.train()
        set_epoch_lr(optimizer, hyp, epoch, epochs)
        mloss = torch.zeros(4, device=device)
        s = ''
        for i, (imgs, targets, _, _) in enumerate(loader):
            ni = i + nb * epoch
            do_step = ni % opt.accumulate == 0
            dp.sync = do_step                 # micro-batches that do not step only accumulate locally (no collective)
            pred = model(imgs)
            loss, loss_items = compute_loss(pred, targets, model, hyp)
            if not torch.isfinite(loss):
                print('WARNING: non-finite loss, ending training ', loss_items)
                return results
            loss.backward()
            if do_step:
                dp.finish()
                optimizer.step()
                dp.zero_grad()
            mloss = (mloss * i + loss_items) / (i + 1)
            mem = torch.cuda.memory_reserved() / 1E9 if use_cuda else 0
            s = ('%10s' * 2 + '%10.3g' * 6) % ('%g/%g' % (epoch, epochs - 1), '%.3gG' % mem, *mloss.tolist(), len(targets),
                                               opt.img_size)
            if rank == 0 and (i % max(1, nb // 5) == 0 or i == nb - 1):
                print(s)
        final_epoch = epoch + 1 == epochs
        # evaluation gate of the reference (train.py:306-314); --test-from overrides its hard-coded "not before epoch 10"
        if not (opt.notest or (opt.nosave and epoch < opt.test_from)) or final_epoch:
            if use_cuda and epoch >= opt.test_from and epoch % int(hyp.get('test_interval', 1)) == 0 and epoch != 0 and not opt.notest:   # the evaluation path (rotated-IoU matching) has no CPU fallback
                import test as test_mod
                with torch.no_grad():
                    results, _ = test_mod.test(opt.cfg, hyp, batch_size=min(batch_size, 16), img_size=opt.img_size, model=model,
                                               conf_thres=0.001 if final_epoch else 0.1, n_images=opt.test_images,
                                               device=device)
                model.train()
        if rank == 0:
            with open(results_file, 'a') as fh:
                fh.write(s + '%10.3g' * 7 % tuple(results) + '\n')
            fitness = results[2]          # mAP
            if fitness > best_fitness:
                best_fitness = fitness
            if not opt.nosave or final_epoch:
                with open(results_file, 'r') as fh:
                    chkpt = {'epoch': epoch, 'best_fitness': best_fitness, 'training_results': fh.read(),
                             'model': model.state_dict(), 'optimizer': None if final_epoch else optimizer.state_dict()}
                os.makedirs(opt.wdir, exist_ok=True)
                torch.save(chkpt, os.path.join(opt.wdir, 'last.pt'))       # what --resume reads (reference train.py:406)
                if best_fitness == fitness:
                    torch.save(chkpt, os.path.join(opt.wdir, 'best.pt'))
                if epoch > 0 and epoch % int(hyp.get('save_interval', 1e9)) == 0:
                    torch.save(chkpt, os.path.join(opt.wdir, 'backup%g.pt' % epoch))
    if rank == 0:
        print('%g epochs completed in %.3f hours.\n' % (epochs - start_epoch, (time.time() - t0) / 3600))
    if world > 1:
        dist.destroy_process_group()
    return results


if __name__ == '__main__':
    from rotate_yolov3_amd.utils.cli import add_ignored, report_ignored
    parser = argparse.ArgumentParser()                                   # the reference's flags (train.py:383-404), same defaults
    parser.add_argument('--accumulate', type=int, default=1, help='batches to accumulate before optimizing')
    parser.add_argument('--hyp', type=str, default='cfg/HRSC/hyp.py', help='hyper-parameter path')
    parser.add_argument('--cfg', type=str, default='cfg/HRSC/yolov3-416.cfg', help='cfg file path')
    parser.add_argument('--data', type=str, default='data/hrsc.data', help='*.data file path (only `classes` is read)')
    parser.add_argument('--img-size', type=int, default=512, help='training size (pixels)')
    parser.add_argument('--resume', action='store_true', help='resume training from last.pt')
    parser.add_argument('--nosave', action='store_true', help='only save final checkpoint')
    parser.add_argument('--notest', action='store_true', help='only test final epoch')
    parser.add_argument('--weights', type=str, default='', help='initial weights')
    parser.add_argument('--arc', type=str, default='defaultpw', help='yolo architecture')  # defaultpw, uCE, uBCE
    parser.add_argument('--name', default='', help='renames results.txt to results_name.txt if supplied')
    parser.add_argument('--device', default='', help="device id (i.e. 0 or 0,1) or cpu; one process per GPU: a list selects entry LOCAL_RANK")
    parser.add_argument('--adam', action='store_true', help='use adam optimizer')
    # additions of this build
    parser.add_argument('--epochs', type=int, default=0, help='override hyp epochs')
    parser.add_argument('--batch-size', type=int, default=0, help='override hyp batch_size (per process)')
    parser.add_argument('--test-from', type=int, default=10, help='first epoch that may be evaluated (reference: 10)')
    parser.add_argument('--test-images', type=int, default=32, help='synthetic images per evaluation')
    parser.add_argument('--wdir', default='weights')
    parser.add_argument('--synthetic', type=int, default=64, help='synthetic images per epoch per process')
    parser.add_argument('--bucket-mb', type=float, default=64.0, help='gradient all-reduce bucket size')
    parser.add_argument('--eager-loss', action='store_true', help='eager compute_loss mirror instead of the graph-captured loss')
    ignored = add_ignored(parser, [
        ('--multi-scale', dict(action='store_true', help='adjust (67%% - 150%%) img_size every 10 batches (the engine plans one input shape)')),
        ('--rect', dict(action='store_true', help='rectangular training (an image-loader mode)')),
        ('--transfer', dict(action='store_true', help='transfer learning: train the yolo layers only (every layer trains here)')),
        ('--prebias', dict(action='store_true', help='transfer-learn yolo biases prior to training (every layer trains here)')),
        ('--evolve', dict(action='store_true', help='evolve hyperparameters')),
        ('--bucket', dict(type=str, default='', help='gsutil bucket')),
        ('--img-weights', dict(action='store_true', help='select training images by weight')),
        ('--cache-images', dict(action='store_true', help='cache images for faster training')),
        ('--var', dict(type=float, default=None, help='debug variable'))])
    opt = parser.parse_args()
    if opt.resume:
        opt.weights = os.path.join(opt.wdir, 'last.pt')          # reference train.py:406
    print(opt)
    report_ignored(parser, opt, ignored)
    hyp = hyp_parse(opt.hyp)
    train(opt, hyp)
    if opt.name and int(os.environ.get('RANK', '0')) == 0 and os.path.isfile(results_file):
        os.replace(results_file, 'results_%s.txt' % opt.name)    # reference train.py:366-368
