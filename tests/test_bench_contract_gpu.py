"""GPU tier: bench.py's output contract -- exactly one JSON line on stdout with the driver's keys, the roofline and
cpu_baseline objects, and the secondary measurements; run small (bs 2, 160x160) so it takes seconds."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_keys(cuda_dev):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--bs", "2", "--size", "160",
                        "--train-bs", "2"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["unit"] == "images/s" and d["dtype"] == "bf16" and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) <= 0.02 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 2500.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    # the headline is BASELINE.json's metric (fwd+bwd); the forward-only leg, the serving step, NMS and the CPU legs ride along
    assert "fwd+bwd" in d["metric"] and "configs[3]" in d["config"]["workload"] and "measured_in" in rf
    fw = d["forward"]
    assert fw["value"] > 0 and "kernels_ms_per_step" in fw and fw["cpu_baseline"]["value"] > 0
    assert fw["graph_replay"]["value"] > 0 and fw["graph_replay"]["ms_per_step"] > 0          # the serving path: one hipGraph per batch
    assert "riou loss" in d["config"]["workload"]              # the headline is configs[3] as BASELINE.json words it
    tr = d["train_step_hbb"]
    assert tr["value"] > 0 and "hbb loss" in tr["workload"] and len(tr["loss_items"]) == 4
    assert d["detect"]["value"] > 0 and d["nms"]["pairs_per_s"] > 1e6
    nr = d["nms"]["roofline"]
    assert nr["bound"] == "valu_fp32" and nr["peak"] == 157.3 and 0 < d["nms"]["pairs_evaluated"] < d["nms"]["pairs"]
    assert 0 < nr["frac"] < 1 and abs(nr["frac"] - nr["achieved"] / nr["peak"]) < 1e-3      # executed flops: a fraction of the peak
    assert d["plumbing"]["images_per_s"] > 0 and d["plumbing"]["io_shape"][0] == 4
    # the train step's own per-kernel table, measured in THIS run: eager steps after the timed region with events
    # around every library call; the headline's roofline object is the step's dominant MFMA kernel, the forward leg keeps its own
    tk = d["train_step_kernels"]
    assert "this run" in tk["source"] and len(tk["kernels"]) > 10
    assert abs(sum(k["ms_per_step"] for k in tk["kernels"]) - tk["ms_per_step_sum"]) < 0.05 * tk["ms_per_step_sum"]
    assert any("wgrad" in k["kernel"] for k in tk["kernels"]) and any("dgrad" in k["kernel"] for k in tk["kernels"])
    assert "train step of this run" in rf["measured_in"] and rf["kernel"] in [k["kernel"] for k in tk["kernels"]]
    assert fw["roofline"]["kernel"] in fw["kernels_ms_per_step"] and "forward leg" in fw["roofline"]["measured_in"]
    assert "hipGraph replay" in d["launch_mode"]
    # the loaded library is this tree's sources
    assert d["build"]["library_is_this_tree"] is True and len(d["build"]["library_id"]) == 16


def test_bench_gpus_2_launches_itself_as_two_ranks(cuda_dev):
    """`python bench.py --gpus 2` as the driver invokes it (no torchrun wrapper, no WORLD_SIZE): the script re-executes itself
    under torch.distributed.run with two ranks.  On this one-GPU box both ranks share cuda:0 and the process group is gloo
    (RCCL refuses two ranks on one device); on the 8-GPU node the same path runs one rank per GPU over RCCL."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo",
                        "--steps", "2", "--warmup", "1", "--bs", "2", "--size", "160", "--train-bs", "2", "--no-nms"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4
    assert "configs[4]" in d["config"]["workload"] and d["value"] > 0
    ar = d["allreduce"]
    assert ar["buckets"] >= 1 and ar["allreduce_ms_standalone"] > 0 and ar["wire_MB"] > 200 and ar["backend"] == "gloo"
    assert "cpu_baseline" not in d            # reported on rank 0 at N = 1 only
    assert "train_step_kernels" not in d      # the committed table is the N = 1, bs 64 step
    # round 5: what every rank saw of the collectives, and how the ranks launched
    assert len(ar["per_rank_ms_per_step"]) == 2 and len(ar["per_rank_allreduce_ms_exposed"]) == 2
    assert d["launch_mode"].startswith("hipGraph replay")
    # round 6: the line carries its own weak-scaling yardstick (rank 0, the same per-GPU batch, collectives off, same run)
    one = d["single_rank_same_bs"]
    assert one["bs"] == 2 and one["rank"] == 0 and one["images_per_s"] > 0 and one["ms_per_step"] > 0
    assert abs(d["scaling_efficiency_vs_single_rank_same_bs"] - d["value"] / (2 * one["images_per_s"])) < 1e-3


def test_bench_two_ranks_one_dies_leaves_a_record_that_says_so(cuda_dev):
    """a rank that dies inside the train leg must not turn the multi-rank run into `rc != 0` and an empty record.
    Rank 1 raises right before its timed steps (RYOLO_BENCH_FAIL_RANK=1) while rank 0 walks into the barrier; rank 0's monitor thread
    prints ONE valid line -- the contract's keys, value null, which rank failed with what, and the phase every rank was in -- and
    the launcher returns non-zero without hanging."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RYOLO_BENCH_FAIL_RANK"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "gloo",
                        "--steps", "2", "--warmup", "1", "--bs", "2", "--size", "160", "--train-bs", "2", "--no-nms", "--rank-timeout", "150"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.split("\n") if ln.strip().startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and d["unit"] == "images/s" and "error" in d
    assert "injected failure on rank 1" in d["rank_report"]["failed"]["1"]
    assert "timed steps" in d["rank_report"]["phase"]["1"]


def test_bench_no_graph_switch(cuda_dev):
    """`bench.py --no-graph`: the same line from eager launches (the fallback for a runtime that refuses stream capture)"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--mode", "train", "--no-graph", "--steps", "2", "--warmup", "3", "--bs", "2",
                        "--size", "160", "--no-cpu-baseline", "--no-nms", "--no-kernel-table"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.split("\n") if ln.strip().startswith("{")][0])
    assert d["value"] > 0 and d["launch_mode"] == "eager launches (--no-graph)"
