"""CPU tier: the failure side channel of multi-rank runs (rotate-yolov3_amd/dist.py: RankMonitor).  Two gloo ranks:
rank 1 fails inside its step; rank 0 sits in a collective rank 1 never joins.  Rank 0 must print ONE parseable line that names the
failed rank, its phase and its error within seconds, and both processes must end (no hang, no empty record)."""
import json
import os
import socket
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch
    import torch.distributed as dist
    import rotate_yolov3_amd  # noqa: F401
    from rotate_yolov3_amd.dist import RankMonitor
    rank, world, mode = int(os.environ["RANK"]), 2, sys.argv[1]
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def on_abort(rep):
        sys.stdout.write(json.dumps({"value": None, "rank_report": rep}) + "\\n")
        sys.stdout.flush()
    mon = RankMonitor(rank, world, on_abort=on_abort if rank == 0 else None, timeout_s=float(sys.argv[2]), poll_s=0.2, linger_s=10.0)
    mon.phase("warm-up")
    dist.barrier()
    mon.phase("timed steps")
    if mode == "fail" and rank == 1:
        try:
            raise RuntimeError("kernel launch failed on rank 1")
        except RuntimeError as e:
            mon.fail("%%s: %%s" %% (type(e).__name__, e))
            sys.exit(7)
    if mode == "hang" and rank == 1:
        time.sleep(60)           # a rank that neither fails nor joins: only the deadline ends the run
        sys.exit(0)
    if mode == "ok":
        dist.barrier()
        mon.close()
        if rank == 0:
            print(json.dumps({"value": 1.0}))
        sys.exit(0)
    t = torch.zeros(1)
    dist.all_reduce(t)           # rank 0 blocks here: rank 1 never joins
    print(json.dumps({"value": "unreachable"}))
""") % ROOT


def _run(tmp_path, mode, timeout_s):
    script = tmp_path / "w.py"
    script.write_text(SCRIPT)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, str(script), mode, str(timeout_s)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=120) for p in procs]
    return [p.returncode for p in procs], outs


def test_failed_rank_is_reported_by_rank_0_instead_of_a_hang(tmp_path):
    rcs, outs = _run(tmp_path, "fail", 300)
    # (rank 0 ends through the monitor (3) or through the exception its collective raises once rank 1 has gone (1); rank 1 through its own
    # exit (7) or through its monitor thread when it sees rank 0's 'abort' first (3): non-zero either way)
    assert rcs[0] in (1, 3) and rcs[1] in (3, 7), (rcs, outs[0][1][-800:], outs[1][1][-800:])
    lines = [l for l in outs[0][0].splitlines() if l.strip().startswith("{")]      # (gloo prints a banner to stdout; bench.py re-routes fd 1)
    assert len(lines) == 1, outs[0][0]
    rec = json.loads(lines[0])
    assert rec["value"] is None
    assert "kernel launch failed on rank 1" in rec["rank_report"]["failed"]["1"]
    assert rec["rank_report"]["phase"]["1"] == "timed steps" and rec["rank_report"]["phase"]["0"] == "timed steps"


def test_silent_rank_hits_the_deadline(tmp_path):
    rcs, outs = _run(tmp_path, "hang", 4)
    assert rcs[0] in (1, 3) and rcs[1] == 3, (rcs, outs[0][1][-800:], outs[1][1][-800:])       # rank 1 is told to end as well
    rec = json.loads([l for l in outs[0][0].splitlines() if l.strip().startswith("{")][0])
    assert rec["value"] is None and "timeout" in rec["rank_report"] and rec["rank_report"]["failed"] == {}


def test_healthy_run_is_untouched(tmp_path):
    rcs, outs = _run(tmp_path, "ok", 300)
    assert rcs == [0, 0], (rcs, outs[0][1][-800:], outs[1][1][-800:])
    assert json.loads(outs[0][0].strip().splitlines()[-1]) == {"value": 1.0}
