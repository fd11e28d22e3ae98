#!/bin/bash
# GPU sanity of the three entry points on the Darknet-53 cfg at a non-default size (416) and batch (6): 2 epochs of
# train.py on synthetic data (eager, eager, capture, replay ...), then test.py and detect.py on the saved checkpoint.
set -e
cd "$(dirname "$0")/.."
T=$(mktemp -d)
python - <<PY
import sys; sys.path.insert(0, '.')
import rotate_yolov3_amd
from rotate_yolov3_amd.cfg import make_cfg
open('$T/d53.cfg', 'w').write(make_cfg.darknet53(416, 416))
open('$T/hyp.py', 'w').write("giou: 0.1\ncls: 27.76\ncls_pw: 1.446\nobj: 20.35\nobj_pw: 3.941\niou_t: 0.3\nang_t: 3.1415926/12\n"
    "reg: 1.0\nfl_gamma: 0.5\ncontext_factor: 1.0\nlr0: 0.0001\nmultiplier:10\nwarm_epoch:1\nmomentum: 0.97\n"
    "weight_decay: 0.0004569\nepochs: 2\nbatch_size: 6\nsave_interval: 300\ntest_interval: 5\n")
PY
python train.py --cfg $T/d53.cfg --hyp $T/hyp.py --img-size 416 --synthetic 24 --wdir $T/w --notest 2>&1 | tail -3
ls $T/w
python test.py --cfg $T/d53.cfg --hyp $T/hyp.py --weights $T/w/best.pt --img-size 416 --batch-size 6 2>&1 | tail -3
python detect.py --cfg $T/d53.cfg --hyp $T/hyp.py --weights $T/w/best.pt --img-size 416 --batch-size 6 --synthetic 12 --conf-thres 0.001 --output $T/out 2>&1 | tail -2
ls $T/out | wc -l
echo ENTRY_SMOKE_OK
