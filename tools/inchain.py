"""Where do the 8-10 % go that a wide 3x3 launch loses inside the forward (VERDICT r4 weak #7b / next #3)?  Runs ONE launch site of
the bs-32 eval engine in three settings, to be traced by rocprofv3 (kernel trace; counter passes GRBM_GUI_ACTIVE / FETCH_SIZE /
WRITE_SIZE in separate runs, tools/inchain.sh):
  phase A  isolated      the target op alone, back to back (what tools/mp_ablate.py and the per-layer tables measure)
  phase B  alternating   (predecessor op, target op) pairs back to back: the 1x1 that feeds it in the network runs before every launch
  phase C  in the chain  whole forwards
Target: layer 14 (3x3 128->256 @76^2 + shortcut, conv_mq) by default; --layer picks another conv layer.  The script writes
gpurun_out/inchain_phases.json: how many dispatches of the target's kernel each phase issues and which of a forward's dispatches of
that kernel is the target, so that tools/inchain_summary.py can attribute the trace's dispatches to the phases by order."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rotate_yolov3_amd  # noqa: E402,F401
from bench import init_bench_weights  # noqa: E402
from rotate_yolov3_amd.cfg import make_cfg  # noqa: E402
from rotate_yolov3_amd.model.engine import HipEngine  # noqa: E402
from rotate_yolov3_amd.model.models import Darknet  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layer", type=int, default=14)
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--forwards", type=int, default=8)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "inchain_phases.json"))
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = init_bench_weights(Darknet(make_cfg.darknet53(608, 608), {"context_factor": 1.0}).eval(), seed=0).to(dev)
    x = torch.rand(32, 3, 608, 608, device=dev)
    eng = HipEngine(model, x.shape, dev)
    conv_ops = [j for j, o in enumerate(eng.op_info) if o["kind"] == "conv"]
    tj = [j for j in conv_ops if eng.op_info[j]["layer"] == a.layer][0]
    pj = conv_ops[conv_ops.index(tj) - 1]
    tname = eng.op_info[tj]["name"]
    same = [j for j in conv_ops if eng.op_info[j]["name"] == tname]
    with torch.no_grad():
        for _ in range(3):
            eng(x)                       # every buffer holds real activations
        torch.cuda.synchronize()
        for _ in range(a.reps):          # phase A
            eng.ops[tj]()
        torch.cuda.synchronize()
        for _ in range(a.reps):          # phase B
            eng.ops[pj]()
            eng.ops[tj]()
        torch.cuda.synchronize()
        for _ in range(a.forwards):      # phase C
            eng(x)
        torch.cuda.synchronize()
    info = {"target_layer": a.layer, "target_kernel": tname, "predecessor": eng.op_info[pj]["name"], "predecessor_layer": eng.op_info[pj]["layer"],
            "flops": eng.op_info[tj]["flops"], "bytes": eng.op_info[tj]["bytes"],
            "warm_forwards": 3, "reps": a.reps, "forwards": a.forwards, "launches_of_kernel_per_forward": len(same),
            "target_index_in_forward": same.index(tj), "pred_same_kernel": eng.op_info[pj]["name"] == tname}
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(info, open(a.out, "w"), indent=1)
    print(json.dumps(info))


if __name__ == "__main__":
    main()
