#!/usr/bin/env python
"""bench.py -- the driver's measurement contract for the MI355X hot path of rotated-YOLOv3.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...,
     or plain `python bench.py --gpus N`: without WORLD_SIZE in the environment the script re-executes itself under
     torch.distributed.run with N ranks on 127.0.0.1 and a free port)

Headline = BASELINE.json's metric, "images/sec fwd+bwd at 608^2": one "step" = one TRAINING step of Darknet-53 (forward with
batch-stat BatchNorm + the reference's loss + backward + gradient all-reduce + SGD-nesterov) on the hand-written HIP
TrainEngine -- configs[3] at N=1 (bs 64), configs[4] at N>1 (32 images per GPU, RCCL all-reduce of the 250 MB gradient);
W warm-up steps, exactly K timed steps between barriers, max over ranks.  Rank 0 prints ONE JSON line.  Besides the contract
keys it carries
  "roofline"      the dominant MFMA kernel OF THE TRAIN STEP, measured in this run: after the timed region the engine launches two
                  steps eagerly with HIP events around every library call on the launch stream ("train_step_kernels" is the whole
                  table); algorithmic FLOP of the kernel's calls / their summed duration vs the bf16 dense MFMA peak; "traffic" =
                  HBM bytes per launch from a committed rocprofv3 --pmc pass when one exists for that kernel (profiles/, FETCH_SIZE
                  doubled per the microarch guide), else null; "whole_step_frac" = 3 x forward FLOP / step time
  "forward"       BASELINE configs[1]: Darknet-53 eval forward, bs 32 per GPU, per-kernel table (events around every launch of the
                  timed eager steps), "forward.roofline" = its dominant kernel, "graph_replay" = the same forward as one hipGraph,
                  its own CPU baseline
  "build"         the id stamped into the loaded library vs the hash of this tree's sources
  "detect"        forward + fused decode/filter + segmented rotated NMS (serving step)
  "nms"           BASELINE configs[2]: rotated IoU + NMS on 50 000 boxes (box-pairs/s) with its VALU roofline
  "cpu_baseline"  N=1 only: the same training step (ATen fp32 chain + loss mirror + autograd) on the host cores at bs 2
  "plumbing"      BASELINE configs[0]: yolov3-tiny, 4x608^2, CPU forward + CPU rotated NMS (no GPU)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0      # bf16 dense, MI355X_MICROARCH.md
GFLOP_PER_IMAGE = 141.98       # sum over the 75 convs of 2*k*k*Cin*Cout*Ho*Wo at 608x608


def init_bench_weights(model, seed=0):
    """Random-init weights of the architecture (no checkpoint travels): variance-preserving uniform conv weights,
    BatchNorm statistics near identity -- keeps activations O(1..1e3) so the MFMA operands are full-range random
    data (zero-ish operands would flatter the clock, cdna_hip_programming.md rule 25)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, t in model.state_dict().items():
            if not t.dtype.is_floating_point:
                continue
            if t.dim() == 4:
                a = (6.0 / t[0].numel()) ** 0.5
                t.copy_((torch.rand(t.shape, generator=g) * 2 - 1) * a)
            elif "running_var" in name:
                t.copy_(1.0 + 0.1 * (torch.rand(t.shape, generator=g) - 0.5))
            elif "running_mean" in name:
                t.copy_(0.1 * (torch.rand(t.shape, generator=g) - 0.5))
            elif "BatchNorm2d.weight" in name:
                t.copy_(0.75 + 0.1 * (torch.rand(t.shape, generator=g) - 0.5))
            elif "activation.weight" in name:
                t.fill_(0.1)
            else:
                t.copy_(0.1 * (torch.rand(t.shape, generator=g) - 0.5))
    return model


_JSON_FD = None
_EMITTED = False
_EMIT_LOCK = __import__("threading").Lock()


def emit_json(obj):
    global _EMITTED
    with _EMIT_LOCK:        # exactly ONE line per run: the rank monitor's abort thread and the main thread can both get here at the very end
        if _EMITTED:
            return
        _EMITTED = True
    line = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--bs", type=int, default=32, help="images per GPU per step (configs[1]: 32)")
    ap.add_argument("--size", type=int, default=608)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-nms", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a hipGraph (no per-op events)")
    ap.add_argument("--mode", default="forward", choices=["forward", "train"],
                    help="forward = configs[1] on the HIP engine (default); train = configs[3]/[4] step on the ATen/MIOpen "
                         "chain (library-backed backward), reported separately")
    ap.add_argument("--no-train", action="store_true", help="skip the embedded train-step measurement")
    ap.add_argument("--train-steps", type=int, default=0, help="timed train steps (default: --steps)")
    ap.add_argument("--fwd-steps", type=int, default=0, help="timed steps of the forward leg (default: min(--steps, 20))")
    ap.add_argument("--train-bs", type=int, default=0, help="embedded train step: images per GPU (default 64 at N=1, 32 at N>1)")
    ap.add_argument("--train-backend", default="hip", choices=["hip", "torch"],
                    help="--mode train: hip = hand-written forward/backward kernels (TrainEngine); torch = ATen/MIOpen autograd")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the RCCL process group even with one rank (exercises the collective path on one GPU)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only for the one-device test of the N>1 command form")
    ap.add_argument("--share-gpu", action="store_true",
                    help="every rank uses cuda:0 (test of the N>1 path on a one-GPU box; needs --dist-backend gloo: RCCL "
                         "refuses two ranks on one device)")
    ap.add_argument("--grad-dtype", default="fp32", choices=["fp32", "bf16"],
                    help="wire format of the gradient all-reduce buckets (bf16 halves the xGMI bytes; each rank sends its share of the mean "
                         "and RCCL adds the shares IN bf16 -- gradients are fp32 again after the collective)")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip the traced eager steps behind train_step_kernels")
    ap.add_argument("--eager-loss", action="store_true", help="--mode train: the eager compute_loss mirror instead of the graph-captured one")
    ap.add_argument("--breakdown", action="store_true", help="--mode train: print GPU ms per phase to stderr")
    ap.add_argument("--dump-train-calls", default="", help="write the traced eager train step's library calls (kernel, shape, us) in launch order to this file")
    ap.add_argument("--dump-ops", default="", help="write the per-op event durations of the timed steps to this file")
    ap.add_argument("--no-graph", action="store_true",
                    help="eager launches everywhere (no hipGraph capture of the train step / loss): the fallback a launcher can use when a "
                         "runtime refuses stream capture under a multi-rank communicator; the engines also fall back by themselves")
    ap.add_argument("--rank-timeout", type=float, default=1500.0,
                    help="N > 1: seconds after which rank 0 prints a failure line (which rank was in which phase) instead of waiting forever "
                         "on a collective a dead rank never joins")
    args = ap.parse_args()
    if args.no_graph:
        os.environ["RYOLO_NO_GRAPH"] = "1"
    # stdout carries exactly ONE line (the JSON): libraries that write to fd 1 (RCCL prints a version banner from C
    # stdio, flushed at exit, i.e. AFTER our line) are sent to stderr for the whole run
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver may invoke it: become N ranks (one process per GPU, RCCL) by re-executing
        # under torch.distributed.run; the JSON line of rank 0 goes to the stdout this process was given
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.dup2(_JSON_FD, 1)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execve(sys.executable, cmd, env)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d needs cuda:%d but the node shows %d GPU(s) (one process per GPU; --share-gpu + "
                         "--dist-backend gloo only for the one-device test)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or args.force_dist
    args.use_dist = use_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=args.dist_backend, rank=rank, world_size=world)

    import rotate_yolov3_amd  # noqa: F401
    # N > 1: a side channel that turns "one rank died, the others hang in a collective, the launcher kills everything, the record is
    # empty" into ONE valid JSON line from rank 0 that says which rank failed in which phase (rotate-yolov3_amd/dist.py: RankMonitor)
    args.monitor = None
    if world > 1:
        from rotate_yolov3_amd.dist import RankMonitor

        def on_abort(report):
            emit_json(failure_line(args, world, "rank failure or timeout", report))
        try:
            # the side channel's port: rank 0 takes a free one and tells the others through the process group's own store (MASTER_PORT + 17 could
            # be taken on a busy node; the ranks must agree on the port before any of them can fail)
            port = None
            try:
                st0 = dist.distributed_c10d._get_default_store()
                if rank == 0:
                    import socket
                    with socket.socket() as sk:
                        sk.bind(("", 0))
                        port = sk.getsockname()[1]
                    st0.set("ryolo_rank_monitor_port", str(port))
                else:
                    port = int(st0.get("ryolo_rank_monitor_port").decode())
            except Exception:       # noqa: BLE001  (no default store: fall back to the fixed offset)
                port = None
            args.monitor = RankMonitor(rank, world, on_abort=on_abort if rank == 0 else None, timeout_s=args.rank_timeout, port=port)
        except Exception as e:      # noqa: BLE001  (no side channel: the run itself is unaffected)
            print("bench.py: rank monitor unavailable (%s: %s)" % (type(e).__name__, e), file=sys.stderr)
    try:
        _bench_body(args, world, rank, dev)
    except BaseException as e:      # noqa: BLE001
        if isinstance(e, SystemExit) and not e.code:
            raise
        import traceback
        msg = "%s: %s | %s" % (type(e).__name__, str(e)[:300], traceback.format_exc().strip().splitlines()[-3:][0].strip()[:160])
        if rank == 0:
            rep = args.monitor.report() if args.monitor is not None else {}
            rep.setdefault("failed", {})["0"] = msg
            emit_json(failure_line(args, world, "rank 0 failed", rep))
        elif args.monitor is not None:
            args.monitor.fail(msg)
        raise
    finally:
        if args.monitor is not None:
            args.monitor.close()


def failure_line(args, world, what, report):
    """the contract's keys with value null: a failed multi-rank run still leaves a parseable record that names the failure"""
    return {"metric": "images/sec fwd+bwd at %d^2 -- FAILED: %s" % (args.size, what), "value": None, "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[%d] train step" % (3 if world == 1 else 4), "parallelism": "dp%d" % world},
            "error": what, "rank_report": report, "launch_mode": "eager (--no-graph)" if args.no_graph else "hipGraph replay requested"}


def gather_floats(vals, world, dev, backend):
    """every rank's list of floats, on every rank (gloo gathers host tensors only)"""
    import torch
    import torch.distributed as dist
    mine = torch.tensor([float(v) for v in vals], dtype=torch.float64, device="cpu" if backend == "gloo" else dev)
    every = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(every, mine)
    return [[float(x) for x in t.cpu()] for t in every]


def _phase(args, name):
    if getattr(args, "monitor", None) is not None:
        args.monitor.phase(name)


def _bench_body(args, world, rank, dev):
    import torch
    import torch.distributed as dist
    _phase(args, "start")
    if args.mode == "train":
        return bench_train(args, world, rank, dev)
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.model.engine import HipEngine
    from rotate_yolov3_amd.model.models import Darknet

    cfg = make_cfg.darknet53(width=args.size, height=args.size)
    torch.manual_seed(0)
    model = init_bench_weights(Darknet(cfg, {"context_factor": 1.0}).eval(), seed=0)
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()} if rank == 0 else None
    model = model.to(dev)
    x = torch.rand(args.bs, 3, args.size, args.size, generator=torch.Generator().manual_seed(rank)).to(dev)
    eng = HipEngine(model, x.shape, dev, use_graph=args.graph)

    def barrier():
        if args.use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    fsteps = args.fwd_steps or min(args.steps, 20)
    _phase(args, "forward leg (configs[1])")
    with torch.no_grad():
        for _ in range(min(args.warmup, 5)):
            eng(x)
        barrier()
        op_ms = [0.0] * len(eng.ops)
        t0 = time.perf_counter()
        if args.graph:
            for _ in range(fsteps):
                eng(x)
        else:
            # eager steps with a HIP event after every op on the launch stream (torch's current stream): the
            # per-kernel durations behind "roofline" come from the timed region itself
            marks = []
            for _ in range(fsteps):
                ev = [torch.cuda.Event(enable_timing=True)]
                ev[0].record()
                n, c, h, w = x.shape
                eng._launch_input(x)
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                ev.append(e)
                for op in eng.ops:
                    op()
                    e = torch.cuda.Event(enable_timing=True)
                    e.record()
                    ev.append(e)
                marks.append(ev)
        barrier()
        elapsed = time.perf_counter() - t0
    if args.use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same forward as the serving path runs it: one hipGraph replay per batch (no events between the launches).  Reported
    # beside the instrumented eager leg, which stays the leg the per-kernel table and "roofline" are measured in.
    graph_ms = None
    if rank == 0 and not args.graph:
        try:
            eng_g = HipEngine(model, x.shape, dev, use_graph=True)
            with torch.no_grad():
                for _ in range(4):
                    eng_g(x)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(fsteps):
                    eng_g(x)
                torch.cuda.synchronize(dev)
                graph_ms = (time.perf_counter() - t1) / fsteps * 1e3
            del eng_g
            torch.cuda.empty_cache()
        except Exception:
            graph_ms = None

    # end-to-end detection step (forward + decode/filter + rotated NMS, SURVEY 8(d) batched shape: ~2000 candidates per
    # image): the serving-path number next to the forward-only headline
    detect_res = None
    if rank == 0 and not args.no_nms:
        try:
            with torch.no_grad():
                io, _ = eng(x)
                sc = io[..., 5].flatten()[::97]
                thr = float(sc.kthvalue(max(1, int(sc.numel() * (1.0 - 2000.0 / io.shape[1])))).values)
                for _ in range(2):
                    dets = eng.detect(x, thr, 0.5)
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(5):
                    dets = eng.detect(x, thr, 0.5)
                torch.cuda.synchronize(dev)
                dms = (time.perf_counter() - t1) / 5 * 1e3
            detect_res = {"workload": "forward + fused decode/score filter/compaction + segmented rotated NMS (thr 0.5), bs=%d, "
                                      "~2000 candidates per image" % args.bs,
                          "value": round(args.bs / dms * 1e3, 1), "unit": "images/s", "ms_per_batch": round(dms, 3),
                          "kept": int(sum(len(d) for d in dets if d is not None))}
        except Exception as e:
            detect_res = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}

    op_info, n_ops = list(eng.op_info), len(eng.ops)
    train_res, riou_res = None, None
    if not args.no_train:
        # configs[3]/[4] in the same run: every rank steps (the gradient all-reduce is a collective), rank 0 reports
        try:
            del eng
            torch.cuda.empty_cache()
            # configs[3] is quoted at bs=64 on one GPU, configs[4] at 32 per GPU
            # headline = BASELINE configs[3] as written: "full train.py step with riou loss"
            train_res = bench_train(args, world, rank, dev, embedded=True, steps=args.train_steps or args.steps,
                                    warmup=args.warmup, bs=args.train_bs or (64 if world == 1 else 32), riou=True)
        except Exception as e:      # never lose the headline line to the secondary measurement
            if world > 1:           # ... but a rank that failed must SAY so: its peers are inside collectives it will never join (main()'s handler
                raise               # publishes the failure through the rank monitor; rank 0 then prints the failure line)
            train_res = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
        # the same step with the reference's own loss (axis-aligned wh_iou term, model/loss.py:322: the parity mode) -- a short leg,
        # same batch, same kernels except the positives' IoU term
        try:
            r = bench_train(args, world, rank, dev, embedded=True, steps=min(args.train_steps or args.steps, 10),
                            warmup=min(args.warmup, 3), bs=args.train_bs or (64 if world == 1 else 32), riou=False)
            if r is not None:
                riou_res = {k: r[k] for k in ("value", "unit", "steps", "warmup", "ms_per_step", "loss_items")}
                riou_res["workload"] = r["config"]["workload"]
        except Exception as e:
            if world > 1:
                raise
            riou_res = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    if rank != 0:
        if args.use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / fsteps * 1e3
    value = args.bs * world * fsteps / elapsed

    roof = None
    kern = {}
    if not args.graph:
        for ev in marks:
            for j in range(n_ops):
                op_ms[j] += ev[j + 1].elapsed_time(ev[j + 2])
        for j, info in enumerate(op_info):
            k = kern.setdefault(info["name"], dict(ms=0.0, flops=0.0, bytes=0.0, launches=0))
            k["ms"] += op_ms[j] / fsteps
            k["flops"] += info["flops"]
            k["bytes"] += info["bytes"]
            k["launches"] += 1
        if args.dump_ops:
            with open(args.dump_ops, "w") as f:
                for j, info in enumerate(op_info):
                    ms = op_ms[j] / fsteps
                    f.write("%3d L%-3d %-26s %8.3f ms %8.1f TF/s %8.1f GB/s\n" % (
                        j, info["layer"], info["name"], ms, info["flops"] / ms / 1e9, info["bytes"] / ms / 1e6))
        dom_name = max(kern, key=lambda n: kern[n]["ms"])
        dom = kern[dom_name]
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": dom_name, "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "launches_per_step": dom["launches"], "avg_launch_us": round(dom["ms"] / dom["launches"] * 1e3, 2),
                "kernel_ms_per_step": round(dom["ms"], 3),
                "measured_in": "eager forward leg of this run (bs=%d), HIP events around every launch on the launch stream" % args.bs,
                "whole_forward_frac": round(GFLOP_PER_IMAGE * (args.size / 608.0) ** 2 * args.bs / ms_per_step
                                            / MFMA_PEAK_TFLOPS, 4)}
        roof.update(load_traffic(dom_name))

    fwd = {
        "metric": "images/sec, Darknet-53 eval forward at %d^2 (BASELINE configs[1])" % args.size,
        "value": round(value, 1), "unit": "images/s", "steps": fsteps, "ms_per_step": round(ms_per_step, 3),
        "config": {"workload": "configs[1]: yolov3.cfg Darknet-53 eval forward, bs=%d/GPU %dx%d bf16, random-init weights, "
                               "input NCHW fp32 resident in HBM" % (args.bs, args.size, args.size),
                   "global_batch": args.bs * world, "graph": bool(args.graph)},
    }
    if kern:
        fwd["kernels_ms_per_step"] = {n: round(v["ms"], 3) for n, v in sorted(kern.items(), key=lambda kv: -kv[1]["ms"])}
    if graph_ms is not None:
        fwd["graph_replay"] = {"ms_per_step": round(graph_ms, 3), "value": round(args.bs / graph_ms * 1e3, 1), "unit": "images/s",
                               "note": "the same forward replayed from one hipGraph (the serving path), rank 0, no per-launch events"}
    if world == 1 and not args.no_cpu_baseline:
        fwd["cpu_baseline"] = cpu_baseline_forward(cfg, sd_cpu, args.size)

    if train_res is not None and "error" not in train_res:
        # the headline: BASELINE.json's metric (fwd+bwd images/s) on configs[3] (N=1) / configs[4] (N>1)
        out = dict(train_res)
        step_frac = out["roofline"]["frac"]
        step_roof = out.pop("step_roofline", None)
        if roof is not None:
            fwd["roofline"] = roof          # the forward leg's dominant kernel stays with the forward leg
        if step_roof is not None:
            # the headline's roofline object: the dominant MFMA kernel of the TRAIN step, measured in this run
            step_roof["whole_step_frac"] = step_frac
            step_roof.update(load_traffic(step_roof["kernel"].split(" ")[0]))
            out["roofline"] = step_roof
        elif roof is not None:
            roof["whole_step_frac"] = step_frac
            out["roofline"] = roof
        out["forward"] = fwd
    else:
        out = {"metric": "images/sec fwd+bwd at %d^2 -- TRAIN LEG FAILED, forward-only numbers under 'forward'" % args.size,
               "value": None, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "configs[3]/[4] train step"}, "roofline": roof, "forward": fwd, "train_error": train_res}
    if riou_res is not None:
        out["train_step_hbb"] = riou_res
    if "train_step_kernels" not in out or "error" in out.get("train_step_kernels", {}):
        tk = load_train_kernel_table() if world == 1 else None      # (traced at N = 1, bs 64: not this step's table at 32 per GPU)
        if tk is not None:
            out["train_step_kernels_committed"] = tk
    if detect_res is not None:
        out["detect"] = detect_res
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_train(args.size)
        out["plumbing"] = cpu_plumbing_config0()
    if world == 1 and not args.no_nms:
        out["nms"] = bench_nms(dev, cpu=not args.no_cpu_baseline)
    out["build"] = build_identity()
    emit_json(out)
    if args.use_dist:
        dist.barrier()
        dist.destroy_process_group()


def build_identity():
    """the id stamped into the loaded library against the hash of this tree's sources"""
    try:
        import __graft_entry__ as ge
        from rotate_yolov3_amd import _lib
        lib_id = _lib.lib().ryolo_build_id().decode().split("=", 1)[1]
        src_id = ge.source_id()
        return {"library_id": lib_id, "sources_id": src_id, "library_is_this_tree": lib_id == src_id, "library": os.path.relpath(_lib.LIB_PATH, ROOT)}
    except Exception as e:      # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}


def load_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel.  rocprofv3 --pmc cannot run inside this process, so the number comes from the
    newest committed counter pass profiles/rNN_traffic.json (tools/traffic_pmc.sh + tools/traffic_summary.py on the dominant layer;
    FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950) -- and only if that pass measured the
    kernel this run found dominant (the file records the kernel's name); otherwise traffic is null rather than stale."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            t = json.load(open(path))
        except Exception:
            continue
        if t.get("kernel") != kernel_name:
            continue
        return {"traffic": t.get("hbm_bytes_per_launch"), "traffic_unit": "bytes per launch",
                "algorithmic_bytes_per_launch": t.get("algorithmic_bytes_per_launch"), "traffic_layer": t.get("shape"),
                "traffic_source": "profiles/%s (%s)" % (os.path.basename(path), t.get("source", ""))}
    return {"traffic": None, "traffic_source": "no committed counter pass for %s" % kernel_name}


_WGRAD_NAMES = {32: "wgrad<32>", 64: "wgrad<64>", 128: "wgrad<128>", 256: "wgrad_wide<256,128>", 257: "wgrad_wide<128,256>",
                258: "wgrad_wide<128,3x64>", 259: "wgrad_wide<128,128>", 260: "wgrad_wide<64,128>"}


def traced_train_table(model, step, dev, nsteps=2, dump_calls=""):
    """The train step's own per-kernel table, measured IN THIS RUN: after the timed region the engine and the
    fused loss are switched to eager launches and `nsteps` steps run with two HIP events around every library call on the launch
    stream (rotate-yolov3_amd/_lib.trace_calls).  A call is named after the kernel the library's own dispatch picks for it (the
    ryolo_conv*_kernel_choice dry runs); a weight gradient is timed as its two launches (tile kernel, split-K reduce), a stride-2 data
    gradient call is its parity-class launches together.  Returns (table, roofline of the step's dominant MFMA kernel) or (None, None)."""
    import ctypes
    import torch
    from rotate_yolov3_amd import _lib
    from rotate_yolov3_amd.model import hip_ops as ops
    engs = [e for e in getattr(model, "_engines", {}).values() if hasattr(e, "_segs")]
    if not engs:
        return None, None
    eng = engs[0]
    st = getattr(eng, "_fused_state", None)
    saved = (eng.force_eager, st.get("no_graph") if st else None)
    eng.force_eager = True
    if st is not None:
        st["no_graph"] = True
    try:
        step()                                           # one untraced eager step
        torch.cuda.synchronize(dev)
        with _lib.trace_calls() as log:
            for _ in range(nsteps):
                step()
        torch.cuda.synchronize(dev)
    finally:
        eng.force_eager = saved[0]
        if st is not None:
            st["no_graph"] = saved[1]
    L = _lib.lib()
    rows = {}
    calls = [] if dump_calls else None

    def conv_flops(d):
        ho = (d.H + 2 * d.pad - d.ksize) // d.stride + 1
        wo = (d.W + 2 * d.pad - d.ksize) // d.stride + 1
        cin = 3 if d.Cin == 8 and d.ksize == 3 and d.Cout == 32 else d.Cin          # layer 0: 3 real input channels padded to 8
        return 2.0 * d.ksize * d.ksize * cin * d.Cout * ho * wo * d.N

    for name, args, e0, e1 in log:
        if any(t in name for t in ("_bytes", "_rows", "kernel_choice", "supported", "strerror", "build_id", "abi_version", "job_fill", "tap_table")):
            continue
        ms = e0.elapsed_time(e1)
        flops, kname = 0.0, name[6:]
        d = getattr(args[0], "_obj", None) if args else None
        if isinstance(d, ops.ConvDesc):
            # a row of this table is ONE kernel symbol (so that rocprofv3's per-kernel averages can be laid beside it): the wide 3x3 tiles
            # have a second instantiation for launches whose GEMM channel count is >= 512 (channel-major K-tile order, csrc/conv_mq.hip)
            def korder(kn, kch):
                return " [channel-major K]" if (kn.startswith("conv_mq") or kn.startswith("conv_mp")) and d.ksize == 3 and kch >= 512 else ""
            if name == "ryolo_conv2d_bn_act_stats":
                code = L.ryolo_conv_kernel_choice(ctypes.byref(d), 1 if args[5] else 0, 1 if args[7] else 0)
                kname = ops.kernel_name_of(code, d.ksize, d.stride, d.Cin)
                kname, flops = kname + " fwd+stats" + korder(kname, d.Cin), conv_flops(d)
            elif name in ("ryolo_conv2d_dgrad", "ryolo_conv2d_dgrad_bnreduce"):
                code = L.ryolo_conv_dgrad_kernel_choice(ctypes.byref(d), 1 if name.endswith("bnreduce") else 0)
                kname = ops.kernel_name_of(code, d.ksize, 1, d.Cout)
                kname = kname + (" dgrad s%d" % d.stride) + (" +bn-reduce" if name.endswith("bnreduce") else "") + korder(kname, d.Cout)
                flops = conv_flops(d)
            elif name in ("ryolo_conv2d_wgrad", "ryolo_conv2d_wgrad_partials"):
                code = L.ryolo_conv_wgrad_kernel_choice(ctypes.byref(d))
                kname = (_WGRAD_NAMES.get(code) or ("wgrad_taps<v%d>" % (code - 1000) if code >= 1000 else "wgrad<?>")) + (
                    " +reduce" if name == "ryolo_conv2d_wgrad" else "")
                flops = conv_flops(d)
            elif name == "ryolo_conv2d_wgrad_reduce":
                kname = "wgrad_reduce (split-K partials -> gradient)"
            elif name.startswith("ryolo_conv0_"):
                kname = "layer0 " + name[12:]            # (bn_act_fwd: the LDS-staged forward; bn_bwd_wgrad: the one-pass backward)
                flops = conv_flops(d) * (2.0 if name.endswith("bn_bwd") else 1.0)      # the backward recomputes z in both passes
        if name == "ryolo_conv_wgrad_reduce_batch":
            kname = "wgrad_reduce_batch (the split-K reduces of a backward segment, one launch)"
        if calls is not None:       # --dump-train-calls: the calls of the traced steps in launch order
            shape = ("k%d s%d %d->%d @%dx%d" % (d.ksize, d.stride, d.Cin, d.Cout, d.H, d.W)) if isinstance(d, ops.ConvDesc) else \
                " ".join(str(a) for a in args if isinstance(a, int) and 0 < a < (1 << 31))
            calls.append("%-58s %-28s %8.1f us" % (kname, shape, ms * 1e3))
        r = rows.setdefault(kname, dict(ms=0.0, launches=0, flops=0.0))
        r["ms"] += ms
        r["launches"] += 1
        r["flops"] += flops
    if not rows:
        return None, None
    if dump_calls:
        with open(dump_calls, "w") as fh:
            fh.write("\n".join(calls[:len(calls) // nsteps]) + "\n")
    kernels = []
    for kname, r in sorted(rows.items(), key=lambda kv: -kv[1]["ms"]):
        e = {"kernel": kname, "calls_per_step": round(r["launches"] / nsteps, 1), "ms_per_step": round(r["ms"] / nsteps, 3)}
        if r["flops"] > 0:
            e["TFLOPs"] = round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1)
        kernels.append(e)
    table = {"source": "this run: %d eager steps after the timed region, HIP events around every library call on the launch stream "
                       "(the timed steps replay hipGraphs)" % nsteps,
             "ms_per_step_sum": round(sum(r["ms"] for r in rows.values()) / nsteps, 2), "kernels": kernels}
    dom = max((k for k in rows if rows[k]["flops"] > 0), key=lambda k: rows[k]["ms"], default=None)
    roof = None
    if dom is not None:
        r = rows[dom]
        ach = r["flops"] / (r["ms"] * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                "calls_per_step": round(r["launches"] / nsteps, 1), "avg_call_us": round(r["ms"] / r["launches"] * 1e3, 2),
                "kernel_ms_per_step": round(r["ms"] / nsteps, 3),
                "measured_in": "train step of this run (eager launches after the timed region), HIP events around every library call"}
    return table, roof


def load_train_kernel_table():
    """Per-kernel table of the TRAIN step from the newest committed rocprofv3 --kernel-trace summary of `bench.py --mode train`
    (profiles/rNN_train_kernel_stats.json, written by tools/rocpd_summary.py --json; the step replays hipGraphs, whose kernels
    cannot be bracketed by events inside this process)."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_train_kernel_stats.json")), reverse=True)
    if not paths:
        return None
    try:
        t = json.load(open(paths[0]))
        t["source"] = "profiles/%s: committed trace of the same command line, not this run" % os.path.basename(paths[0])
        return t
    except Exception:
        return None


def bench_train(args, world, rank, dev, embedded=False, steps=None, warmup=None, bs=None, riou=False):
    """configs[3] (N=1) / configs[4] (N>1): one optimisation step per "step": forward (batch-stat BatchNorm) + the
    reference's hbb loss mirror + backward + gradient all-reduce (rotate-yolov3_amd/dist.py over RCCL) + SGD-nesterov.
    --train-backend hip: the hand-written TrainEngine (conv fwd/dgrad/wgrad on MFMA, BN+PReLU fwd/bwd kernels);
    --train-backend torch: the ATen chain under bf16 autocast (MIOpen), kept as the library yardstick."""
    import torch
    import torch.distributed as dist
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.dist import GradientAllReducer
    bs = bs or args.bs
    from rotate_yolov3_amd.model.loss import compute_loss
    from rotate_yolov3_amd.model.models import Darknet
    from rotate_yolov3_amd.utils.synthetic import synthetic_targets
    hyp = {"giou": 0.1, "cls": 27.76, "cls_pw": 1.0, "obj": 20.35, "obj_pw": 1.0, "iou_t": 0.5, "ang_t": 3.1415926 / 12,
           "reg": 1.0, "fl_gamma": 0.5, "context_factor": 1.0, "lr0": 1e-4, "momentum": 0.97, "weight_decay": 0.0004569}
    if riou:
        hyp["riou"] = 1      # lreg's wh_iou term -> 1 - rotated IoU with the polygon-overlap gradient (csrc/riou_grad.h)
    torch.manual_seed(0)
    model = init_bench_weights(Darknet(make_cfg.darknet53(args.size, args.size), hyp), seed=0).to(dev).train()
    model.nc, model.arc, model.hyp = 1, "default", hyp
    model.backend = args.train_backend
    if args.train_backend == "hip" and not args.eager_loss:
        model.enable_fused_loss(capacity=max(256, 8 * bs))      # compute_loss = one hipGraph replay (loss_static.py)
    from train import make_optimizer
    opt = make_optimizer(model, hyp)
    dp = GradientAllReducer(model, wire_dtype=torch.bfloat16 if args.grad_dtype == "bf16" else None)
    if world > 1 and hasattr(opt, 'grad_scale'):
        opt.grad_scale = 1.0 / world          # the 1/world of the gradient average rides in the SGD kernel, not in a div_ pass
        dp.scale_in_optimizer = True
    x = torch.rand(bs, 3, args.size, args.size, generator=torch.Generator(device=dev).manual_seed(rank), device=dev)   # per-rank batch, generated on the device
    tg = synthetic_targets(bs, seed=1 + rank, device=dev)

    marks = []

    def mark():
        if args.breakdown:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)

    def step():
        mark()
        with torch.autocast(device_type="cuda", dtype=torch.bfloat16):
            pred = model(x)
        mark()
        loss, items = compute_loss([p.float() for p in pred], tg.clone(), model, hyp)
        mark()
        loss.backward()
        mark()
        dp.finish()
        opt.step()
        dp.zero_grad()
        mark()
        return items

    nsteps = steps or args.steps
    _phase(args, "train leg (%s loss): warm-up (eager steps, graph capture)" % ("riou" if riou else "hbb"))
    for _ in range(warmup if warmup is not None else args.warmup):
        step()
    if args.use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    _phase(args, "train leg (%s loss): timed steps" % ("riou" if riou else "hbb"))
    if os.environ.get("RYOLO_BENCH_FAIL_RANK") == str(rank):          # fault injection for tests/test_bench_contract_gpu.py: this rank dies here,
        raise RuntimeError("injected failure on rank %d (RYOLO_BENCH_FAIL_RANK)" % rank)   # its peers are about to enter a collective
    t0 = time.perf_counter()
    for _ in range(nsteps):
        items = step()
    if args.use_dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    elapsed_local = elapsed
    if args.breakdown and rank == 0:
        m = marks[-5 * nsteps:]
        names = ["forward", "loss", "backward", "allreduce+optimizer"]
        tot = [0.0] * 4
        for i in range(nsteps):
            for k in range(4):
                tot[k] += m[5 * i + k].elapsed_time(m[5 * i + k + 1])
        print("breakdown (GPU ms/step): " + "  ".join("%s %.2f" % (n, v / nsteps) for n, v in zip(names, tot)), file=sys.stderr,
              flush=True)
    if args.use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    comm = None
    if args.use_dist and riou:
        # the collective on its own (nothing to hide under) and what the step shows of it: the same steps without the
        # collectives (gradient-accumulation mode of the reducer), max over ranks
        _phase(args, "train leg: collectives alone / step without collectives")
        ar_ms = dp.time_collectives()
        dp.sync = False
        for _ in range(2):
            step()
        dist.barrier()
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        nloc = min(nsteps, 5)
        for _ in range(nloc):
            step()
        dist.barrier()
        torch.cuda.synchronize(dev)
        t = torch.tensor([(time.perf_counter() - t1) / nloc * 1e3, ar_ms], dtype=torch.float64, device=dev)
        # per rank: its own step time with and without the collectives (what the overlap leaves exposed on THAT rank)
        every = gather_floats([elapsed_local / nsteps * 1e3, float(t[0])], world, dev, args.dist_backend)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dp.sync = True
        nosync_ms, ar_ms = float(t[0]), float(t[1])
        wire = dp.wire_bytes()
        comm = {"buckets": len(dp.buckets), "wire_dtype": args.grad_dtype, "wire_MB": round(wire / 1e6, 1),
                "allreduce_ms_standalone": round(ar_ms, 3),
                "allreduce_busbw_GBps": round(2.0 * (world - 1) / world * wire / (ar_ms * 1e-3) / 1e9, 1) if world > 1 and ar_ms > 0 else None,
                "ms_per_step_without_collectives": round(nosync_ms, 2),
                "allreduce_ms_exposed": round(elapsed / nsteps * 1e3 - nosync_ms, 3), "backend": args.dist_backend,
                "per_rank_ms_per_step": [round(float(v[0]), 3) for v in every],
                "per_rank_allreduce_ms_exposed": [round(float(v[0] - v[1]), 3) for v in every]}
    fallback_ranks = []
    if args.use_dist:           # which ranks fell back to eager launches after a refused capture (every rank reports, rank 0 prints)
        eng_me = [e for e in getattr(model, "_engines", {}).values() if hasattr(e, "_segs")]
        every_fb = gather_floats([1.0 if (eng_me and eng_me[0].graph_fallback) else 0.0], world, dev, args.dist_backend)
        fallback_ranks = [r for r, v in enumerate(every_fb) if float(v[0]) > 0 and r != 0]
    table, step_roof = (None, None)
    if rank == 0 and (riou or not embedded) and not args.use_dist and args.train_backend == "hip" and not args.no_kernel_table:
        try:
            table, step_roof = traced_train_table(model, step, dev, dump_calls=args.dump_train_calls)
        except Exception as e:      # noqa: BLE001  the table is a secondary measurement
            table = {"error": "%s: %s" % (type(e).__name__, str(e)[:200])}
    res = None
    if rank == 0:
        ms = elapsed / nsteps * 1e3
        res = {
            "metric": "images/sec fwd+bwd at %d^2 (train step, backend=%s)" % (args.size, args.train_backend),
            "value": round(bs * world * nsteps / elapsed, 1), "unit": "images/s", "n_gpus": world,
            "steps": nsteps, "warmup": warmup if warmup is not None else args.warmup, "ms_per_step": round(ms, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "configs[%d]: Darknet-53 train step (fwd + %s loss + bwd + grad all-reduce + SGD), bs=%d/GPU "
                                   "%dx%d, synthetic HRSC-shaped targets" % (3 if world == 1 else 4, "riou" if riou else "hbb", bs,
                                                                             args.size, args.size),
                       "global_batch": bs * world, "parallelism": "dp%d" % world,
                       "gradients": "%.0f MB fp32 in %d flat buckets, %s on the wire" % (dp.grad_bytes() / 1e6, len(dp.buckets), args.grad_dtype)},
            "roofline": {"bound": "mfma", "achieved": round(3 * GFLOP_PER_IMAGE * bs / ms, 1), "peak": MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(3 * GFLOP_PER_IMAGE * bs / ms / MFMA_PEAK_TFLOPS, 4),
                         "traffic": None, "note": "whole step (fwd + loss + bwd + optimizer), 3 x forward FLOP"},
            "loss_items": [round(float(v), 4) for v in items]}
        if comm is not None:
            res["allreduce"] = comm
            # the N = 1 line is configs[3] (bs 64 on one GPU), this line is configs[4] (32 per GPU), so value(N) / value(1)
            # of two records is not like for like.  The same-run yardstick: rank 0's own step on its 32 images with the collectives off
            # (gradient-accumulation mode of the reducer, measured above) -- scaling efficiency = value / (n_gpus x images_per_s here).
            r0 = comm["per_rank_ms_per_step"][0] - comm["per_rank_allreduce_ms_exposed"][0]
            res["single_rank_same_bs"] = {"bs": bs, "ms_per_step": round(r0, 3), "images_per_s": round(bs / r0 * 1e3, 1), "rank": 0,
                                          "note": "rank 0, %d images, no collectives, same process and run; weak-scaling efficiency = "
                                                  "value / (n_gpus x images_per_s); the N = 1 record of this bench is configs[3] at bs 64 "
                                                  "and is NOT the denominator" % bs}
            res["scaling_efficiency_vs_single_rank_same_bs"] = round(res["value"] / (world * bs / r0 * 1e3), 4)
        if table is not None:
            res["train_step_kernels"] = table
        if step_roof is not None:
            res["step_roofline"] = step_roof
        eng_ = [e for e in getattr(model, "_engines", {}).values() if hasattr(e, "_segs")]
        if args.no_graph:
            res["launch_mode"] = "eager launches (--no-graph)"
        elif eng_ and eng_[0].graph_fallback:
            res["launch_mode"] = "eager launches (hipGraph capture failed: %s)" % eng_[0].graph_fallback
        elif fallback_ranks:
            res["launch_mode"] = "hipGraph replay on rank 0; eager launches after a failed capture on rank(s) %s" % fallback_ranks
        else:
            res["launch_mode"] = "hipGraph replay (forward, loss and each backward segment captured after two eager steps)"
        if eng_ and getattr(eng_[0], "reduce_fallback", None):
            res["launch_mode"] += "; " + eng_[0].reduce_fallback
        if not embedded:
            emit_json(res)
    del model, opt, dp
    torch.cuda.empty_cache()
    if args.use_dist and not embedded:
        dist.barrier()
        dist.destroy_process_group()
    return res


def cpu_baseline_forward(cfg, sd, size):
    """The oracle's restatement of the reference's CPU path (same ATen operators, fp32) on a bounded sample."""
    import torch
    from oracle import darknet_oracle as do
    import oracle
    cores = oracle.host_cores(32)
    torch.set_num_threads(cores)
    n_img = 2
    x = torch.rand(n_img, 3, size, size, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        do.forward(cfg, sd, x[:1])           # warm-up (oneDNN primitive creation)
        t0 = time.perf_counter()
        reps = 0
        while True:
            do.forward(cfg, sd, x)
            reps += 1
            if time.perf_counter() - t0 > 10.0 or reps >= 8:
                break
        dt = time.perf_counter() - t0
    return {"value": round(n_img * reps / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d passes of %d images %dx%d, fp32 ATen/oneDNN conv+BN+PReLU chain (oracle/darknet_oracle.py), "
                      "%d torch threads" % (reps, n_img, size, size, cores)}


def cpu_baseline_train(size):
    """The same training step on the host: the Darknet module's ATen fp32 operator chain (what the reference dispatches on
    CPU tensors) + the loss mirror + autograd backward + SGD, bs 2, a bounded number of steps."""
    import torch
    import oracle
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.model.loss import compute_loss
    from rotate_yolov3_amd.model.models import Darknet
    from rotate_yolov3_amd.utils.synthetic import synthetic_targets
    cores = oracle.host_cores(32)
    torch.set_num_threads(cores)
    hyp = {"giou": 0.1, "cls": 27.76, "cls_pw": 1.0, "obj": 20.35, "obj_pw": 1.0, "iou_t": 0.5, "ang_t": 3.1415926 / 12,
           "reg": 1.0, "fl_gamma": 0.5, "context_factor": 1.0, "lr0": 1e-4, "momentum": 0.97, "weight_decay": 0.0004569}
    torch.manual_seed(0)
    model = init_bench_weights(Darknet(make_cfg.darknet53(size, size), hyp), seed=0).train()
    model.nc, model.arc, model.hyp = 1, "default", hyp
    from train import make_optimizer
    opt = make_optimizer(model, hyp, fused=False)
    bs = 2
    x = torch.rand(bs, 3, size, size, generator=torch.Generator().manual_seed(0))
    tg = synthetic_targets(bs, seed=1, device=torch.device("cpu"))

    def step():
        opt.zero_grad()
        loss, _ = compute_loss(model(x), tg.clone(), model, hyp)
        loss.backward()
        opt.step()
    step()                                   # warm-up
    t0 = time.perf_counter()
    reps = 0
    while True:
        step()
        reps += 1
        if time.perf_counter() - t0 > 15.0 or reps >= 6:
            break
    dt = time.perf_counter() - t0
    return {"value": round(bs * reps / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d training steps of bs %d at %dx%d: ATen/oneDNN fp32 conv+BN+PReLU chain of the Darknet module on CPU "
                      "tensors + loss mirror + autograd backward + SGD, %d torch threads" % (reps, bs, size, size, cores)}


def cpu_plumbing_config0():
    """BASELINE configs[0]: yolov3-tiny (rotated head mapping of SURVEY 8(d) config 1), 4 x 608 x 608 random tensors, CPU
    forward + the Python rotated-NMS wrapper over the C oracle -- plumbing, no GPU."""
    import torch
    import oracle
    from oracle import darknet_oracle as do
    from rotate_yolov3_amd.cfg import make_cfg
    from rotate_yolov3_amd.model.models import Darknet
    cores = oracle.host_cores(32)
    torch.set_num_threads(cores)
    cfg = make_cfg.tiny()
    torch.manual_seed(0)
    model = Darknet(cfg, {"context_factor": 1.0}).eval()
    sd = model.state_dict()
    x = torch.rand(4, 3, 608, 608, generator=torch.Generator().manual_seed(0))
    with torch.no_grad():
        # the PRODUCT's host path on CPU tensors (cfg parser -> create_modules -> ATen forward -> YOLO decode: VERDICT r5 weak #10) ...
        model(x[:1])                                     # (one-time allocations / thread-pool start outside the timed call)
        tp = time.perf_counter()
        io, _ = model(x)
        t0 = time.perf_counter()
        # ... checked against, and timed beside, the oracle's functional forward; the rotated NMS on the CPU is the oracle's (the
        # product has no CPU NMS: r_nms raises on CPU tensors like the reference's op)
        io_o, _ = do.forward(cfg, sd, x)
        t1 = time.perf_counter()
        err = float((io - io_o).abs().max() / io_o.abs().max())
        score = (io[..., 5:6] * io[..., 6:]).max(2)[0]
        thr = float(score.flatten().kthvalue(int(score.numel() * 0.999)).values)     # ~65 candidates per image
        det = do.non_max_suppression(io.clone(), thr, 0.5)
        t2 = time.perf_counter()
    return {"workload": "configs[0]: yolov3-tiny, 4x608x608 random tensors on the CPU: the product's Darknet module (parser, module list, ATen "
                        "forward, decode) + the oracle's rotated-NMS wrapper (C oracle); no GPU",
            "forward_s": round(t0 - tp, 3), "oracle_forward_s": round(t1 - t0, 3), "forward_vs_oracle_max_rel_err": err,
            "nms_s": round(t2 - t1, 3), "images_per_s": round(4 / ((t0 - tp) + (t2 - t1)), 3), "cores": cores,
            "detections": int(sum(len(d) for d in det if d is not None)), "io_shape": list(io.shape)}


def bench_nms(dev, cpu=True, n=50000, reps=5):
    import torch
    from rotate_yolov3_amd.utils.nms.r_nms import r_nms
    from rotate_yolov3_amd.utils.synthetic import random_boxes
    d = random_boxes(n, seed=0)
    dt = torch.from_numpy(d).to(dev)
    for _ in range(2):
        keep = r_nms(dt, 0.5)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        keep = r_nms(dt, 0.5)
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / reps * 1e3
    pairs = n * (n - 1) / 2
    # pairs whose exact polygon IoU was evaluated (the rest left through the 6-flop bounding-circle reject): one counted call
    import ctypes
    from rotate_yolov3_amd import _lib
    cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.lib().ryolo_rnms_count_pairs(ctypes.c_void_p(cnt.data_ptr()))
    r_nms(dt, 0.5)
    torch.cuda.synchronize(dev)
    _lib.lib().ryolo_rnms_count_pairs(None)
    evaluated = int(cnt.item())
    VALU_PEAK = 157.3      # fp32 vector TFLOP/s, MI355X_MICROARCH.md
    flop_ref = 421.0 * pairs                          # 421 fp32 flop per disjoint pair in the reference formulation
    flop_done = 421.0 * evaluated + 6.0 * (pairs - evaluated)
    res = {"workload": "configs[2]: %d random rotated boxes (SURVEY 8(d) distribution), thr 0.5, sort + IoU mask + greedy "
                       "scan + index output" % n,
           "pairs_per_s": float("%.4g" % (pairs / ms * 1e3)), "ms": round(ms, 3), "kept": int(keep.numel()), "unit": "box-pairs/s",
           "pairs": int(pairs), "pairs_evaluated": evaluated,
           "roofline": {"bound": "valu_fp32", "achieved": round(flop_done / ms / 1e9, 2), "peak": VALU_PEAK, "unit": "TFLOP/s",
                        "frac": round(flop_done / ms / 1e9 / VALU_PEAK, 4),
                        "note": "EXECUTED flops (421 per pair whose polygon IoU was evaluated + 6 per pair retired by the bounding-"
                                "circle test) / whole-call time; the mask kernel is bound by VALU ISSUE slots (99 % of SIMD-cycles, "
                                "profiles/r06_pmc_rnms.txt) at about half of the lanes live, not by the fp32 flop rate",
                        "reference_formulation_tflops": round(flop_ref / ms / 1e9, 2)}}
    # the batched-detection shape, 32 images x 2000 candidates, as ONE segmented launch (every (image,
    # class) set of a batch at once -- what non_max_suppression_batched calls)
    from rotate_yolov3_amd.utils.nms.r_nms import r_nms_segmented
    sets, per = 32, 2000
    dd = []
    for k in range(sets):
        b = torch.from_numpy(random_boxes(per, seed=100 + k))
        dd.append(b[(-b[:, 5]).argsort(stable=True)])
    dd = torch.cat(dd).to(dev)
    off = torch.arange(0, sets * per + 1, per, dtype=torch.int32, device=dev)
    for _ in range(2):
        fl = r_nms_segmented(dd, off, per, 0.5)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        fl = r_nms_segmented(dd, off, per, 0.5)
    torch.cuda.synchronize(dev)
    bms = (time.perf_counter() - t0) / reps * 1e3
    res["batched"] = {"workload": "%d sets x %d boxes (score-sorted), one segmented launch" % (sets, per),
                      "ms": round(bms, 3), "pairs_per_s": float("%.4g" % (sets * per * (per - 1) / 2 / bms * 1e3)),
                      "kept": int(fl.sum())}
    if cpu:
        from oracle import riou            # the CPU baseline leg: the only use of oracle/ in this function
        ns = 8192
        ds = random_boxes(ns, seed=13)
        t0 = time.perf_counter()
        k, npairs = riou.rnms(ds, 0.5, nthreads=1, return_pairs=True)
        dtc = time.perf_counter() - t0
        # "single-thread and with all cores on the row loop": the same call with the column loop of a row split over
        # OpenMP threads (oracle_rnms, nthreads > 1); same keep list by construction (asserted)
        import numpy as np
        import oracle
        nc = oracle.host_cores(64)
        t0 = time.perf_counter()
        k2, npairs2 = riou.rnms(ds, 0.5, nthreads=nc, return_pairs=True)
        dta = time.perf_counter() - t0
        assert np.array_equal(k, k2) and npairs == npairs2
        res["cpu_baseline"] = {"value": float("%.4g" % (npairs / dtc)), "unit": "evaluated box-pairs/s", "cores": 1, "kind": "port",
                               "value_all_pairs": float("%.4g" % (ns * (ns - 1) / 2 / dtc)),
                               "value_all_cores": float("%.4g" % (npairs / dta)), "cores_all": nc,
                               "value_all_pairs_all_cores": float("%.4g" % (ns * (ns - 1) / 2 / dta)),
                               "sample": "oracle/riou_oracle.c greedy NMS of %d boxes; `value` = single thread, counts the %d IoU "
                                         "evaluations the lazy greedy loop performs (compare with pairs_evaluated / time on the GPU), "
                                         "`value_all_pairs` counts n(n-1)/2 like the GPU's pairs_per_s; `value_all_cores` = the same call "
                                         "with the column loop of each row split over %d OpenMP threads" % (ns, npairs, nc)}
    return res


if __name__ == "__main__":
    main()
