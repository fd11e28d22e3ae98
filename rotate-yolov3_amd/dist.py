"""Data-parallel gradient exchange for one process per GPU -- replaces the reference's single call site
`dist.init_process_group('nccl', world_size=1) + DistributedDataParallel(model)` (train.py:169-175), which relies on a
single-process-multi-device mode that no longer exists and crashes in compute_loss (model/loss.py:313).

Design for MI355X / xGMI: 8 GPUs fully connected by point-to-point links; a ring all-reduce of the
249.6 MB fp32 gradient moves 2*(7/8)*S through ONE link per direction (~2.9 ms at 153 GB/s), so the exchange is
issued as a few LARGE flat buckets (default 64 MiB: per-collective latency amortised, still 4 buckets to overlap) as
soon as the last gradient of a bucket has been accumulated, on RCCL's own stream, overlapped with the rest of backward.
Gradients live IN the flat bucket (`p.grad` is a view), so there is no gather copy and the optimizer reads the reduced
values in place.  BatchNorm statistics stay per replica (no SyncBN), as in the reference.
With the HIP TrainEngine (whose backward is a list of kernel launches replayed from hipGraphs, not autograd nodes) the
engine cuts that list where a bucket has received its last gradient (model._dp_buckets), flushes the segment's
gradients into the bucket views and runs the same hooks, so buckets go out while the earlier layers' backward runs.

    dp = GradientAllReducer(model)        # broadcasts parameters + buffers from rank 0, builds buckets, installs hooks
    loss.backward(); dp.finish()          # wait for the in-flight buckets; grads now hold the world average
Gradient accumulation: set `dp.sync = False` before the backward of every micro-batch but the last (gradients only add
up locally in the buckets, no collective), `dp.sync = True` before the last one, then finish().
"""
import torch
import torch.distributed as dist


class GradientAllReducer(object):
    def __init__(self, model, bucket_mb=64.0, process_group=None, average=True, wire_dtype=None):
        # wire_dtype = torch.bfloat16: the buckets travel as bf16 (half the xGMI bytes: 125 MB instead of 250 MB per step for
        # Darknet-53); gradients stay fp32 locally -- one cast pass each way per bucket.  The cross-rank SUM itself is done by
        # RCCL in bf16 (8-bit mantissa over `world` addends); to keep the addends in range the bucket is pre-scaled by 1/world
        # before the cast when the reducer averages, so the wire carries the mean, not the sum.
        # Default (None): fp32 on the wire, bit-compatible with a single-process mean of per-rank gradients.
        self.wire_dtype = wire_dtype
        self.model = model
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.collective = dist.is_initialized()          # a one-rank group still runs the collectives (tests)
        self.average = average
        self.sync = True                # False: accumulate locally (micro-batches before the last of an accumulation group)
        self.params = [p for p in model.parameters() if p.requires_grad]
        if self.collective:
            with torch.no_grad():
                for t in list(model.parameters()) + list(model.buffers()):
                    dist.broadcast(t.data, src=0, group=process_group)
        # buckets in REVERSE registration order: the last layers' gradients are ready first
        cap = int(bucket_mb * 1024 * 1024)
        self.buckets = []        # dict(flat, params, pending, handle)
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nbytes = p.numel() * p.element_size()
            if cur and (cur_bytes + nbytes > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self._close(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nbytes
        if cur:
            self._close(cur)
        # the HIP TrainEngine cuts its backward launch list at the bucket boundaries and runs the hooks itself
        core = model.module if hasattr(model, 'module') and hasattr(model.module, 'module_list') else model
        core._dp_buckets = [list(b["params"]) for b in self.buckets]
        core._dp_grad_views = {p: p.grad for p in self.params}     # the engine's kernels accumulate straight into these
        for eng in getattr(core, '_engines', {}).values():
            if hasattr(eng, '_reset_grad_sink'):
                eng._reset_grad_sink()
        self._hooks = []
        if self.collective:
            for bi, b in enumerate(self.buckets):
                for p in b["params"]:
                    self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    def _close(self, params):
        n = sum(p.numel() for p in params)
        flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        off = 0
        for p in params:
            p.grad = flat[off:off + p.numel()].view_as(p)      # gradient accumulates straight into the bucket
            off += p.numel()
        self.buckets.append(dict(flat=flat, params=params, pending=len(params), handle=None))

    def _make_hook(self, bi):
        def hook(param):
            if not self.sync:
                return
            b = self.buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0:
                b["handle"] = self._launch(b)
        return hook

    def _launch(self, b):
        if self.wire_dtype is None:
            return dist.all_reduce(b["flat"], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        if b.get("wire") is None:
            b["wire"] = torch.empty_like(b["flat"], dtype=self.wire_dtype)
        if self._wire_prescale():
            torch.mul(b["flat"], 1.0 / self.world, out=b["wire"])      # fused scale + cast: the wire carries each rank's share of the mean
        else:
            b["wire"].copy_(b["flat"])
        return dist.all_reduce(b["wire"], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def _wire_prescale(self):
        return self.wire_dtype is not None and self.average and self.world > 1

    def finish(self):
        """Block until every bucket launched during this backward has been reduced; average; re-arm.  With sync = False
        (a micro-batch that only accumulates) nothing is reduced."""
        for b in self.buckets:
            if self.collective and self.sync:
                if b["handle"] is None:        # a parameter got no gradient this step: reduce what there is
                    b["handle"] = self._launch(b)
                b["handle"].wait()
                in_opt = getattr(self, 'scale_in_optimizer', False)
                if self.wire_dtype is not None:
                    if self._wire_prescale() and in_opt:       # the optimizer applies 1/world itself: hand it the sum (same pass as the cast back)
                        torch.mul(b["wire"], float(self.world), out=b["flat"])
                    else:
                        b["flat"].copy_(b["wire"])
                elif self.average and self.world > 1 and not in_opt:
                    b["flat"].div_(self.world)
            b["handle"] = None
            b["pending"] = len(b["params"])

    def zero_grad(self):
        for b in self.buckets:
            b["flat"].zero_()

    def detach(self):
        """Give the model back: remove the hooks and the engine-facing views (the buckets' storage is no longer kept alive by the
        model; param.grad keeps pointing into it until the caller replaces it)."""
        for h in self._hooks:
            h.remove()
        self._hooks = []
        core = self.model.module if hasattr(self.model, 'module') and hasattr(self.model.module, 'module_list') else self.model
        core._dp_buckets = None
        core._dp_grad_views = None
        for eng in getattr(core, '_engines', {}).values():
            if hasattr(eng, '_reset_grad_sink'):
                eng._reset_grad_sink()

    def grad_bytes(self):
        return sum(b["flat"].numel() * b["flat"].element_size() for b in self.buckets)

    def wire_bytes(self):
        es = torch.empty((), dtype=self.wire_dtype).element_size() if self.wire_dtype is not None else None
        return sum(b["flat"].numel() * (es or b["flat"].element_size()) for b in self.buckets)

    def time_collectives(self, reps=5):
        """Stand-alone time of one step's bucket all-reduces (no backward to hide under), ms; gradients are restored."""
        if not self.collective:
            return 0.0
        keep = [b["flat"].clone() for b in self.buckets]
        dev = self.buckets[0]["flat"].device
        for _ in range(2):
            for b in self.buckets:
                self._launch(b).wait()
        torch.cuda.synchronize(dev)
        dist.barrier(group=self.pg)
        import time
        t0 = time.perf_counter()
        for _ in range(reps):
            hs = [self._launch(b) for b in self.buckets]
            for h in hs:
                h.wait()
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t0) / reps * 1e3
        for b, k in zip(self.buckets, keep):
            b["flat"].copy_(k)
        return dt


class RankMonitor(object):
    """Failure surfacing for a one-process-per-GPU run (bench.py --gpus N, train.py under torchrun).

    A rank that dies inside a step leaves its peers blocked in an RCCL collective; under torchrun the agent then tears the job
    down and the driver sees rc != 0 and an EMPTY record.  The monitor is a side channel that does not depend on the data-path
    process group: a small TCP key/value store served by rank 0 (MASTER_PORT + 17) and one daemon thread per rank.
      * every rank publishes its phase (`phase('train warm-up')`) and, from its exception handler, its failure (`fail(msg)`);
        a failing rank then lingers (up to `linger_s`) until rank 0 has acknowledged, so that rank 0 gets to print before the
        launcher's tear-down reaches it;
      * rank 0's thread polls for failures and for the no-progress deadline (`timeout_s` since the last phase change of any rank); on either it calls `on_abort(report)` -- bench.py prints a
        valid JSON line with value null and the report (which rank, which phase, what error) -- and ends the process;
      * the other ranks' threads end their process when rank 0 has published 'abort' (no stragglers holding GPUs).
    Without a failure the monitor costs one store round trip per second per rank.  `close()` stops it."""

    def __init__(self, rank, world, on_abort=None, timeout_s=1800.0, host=None, port=None, linger_s=20.0, poll_s=1.0):
        import datetime
        import os
        import threading
        import time
        self.rank, self.world, self.on_abort = int(rank), int(world), on_abort
        self.linger_s, self.poll_s = float(linger_s), float(poll_s)
        self._time, self._os = time, os
        self.timeout_s = float(timeout_s)           # a NO-PROGRESS limit: every phase change of any rank starts it again (one wall-clock
        self.deadline = time.time() + self.timeout_s   # limit from construction killed healthy long runs -- many steps, cold kernel caches)
        self._seen = None
        host = host or os.environ.get("MASTER_ADDR", "127.0.0.1")
        port = int(port or (int(os.environ.get("MASTER_PORT", "29531")) + 17))
        self.store = dist.TCPStore(host, port, self.world, is_master=(self.rank == 0), timeout=datetime.timedelta(seconds=60),
                                   wait_for_workers=False)
        self._stop = threading.Event()
        self._lock = threading.Lock()
        self._thread = threading.Thread(target=self._run, name="ryolo-rank-monitor", daemon=True)
        self._thread.start()

    def phase(self, name):
        try:
            with self._lock:
                self.store.set("phase_%d" % self.rank, str(name))
        except Exception:       # noqa: BLE001  the side channel must never take the run down
            pass

    def fail(self, msg):
        """Publish this rank's failure; returns after rank 0 acknowledged it (or after linger_s)."""
        try:
            with self._lock:
                self.store.set("fail_%d" % self.rank, str(msg)[:500])
        except Exception:       # noqa: BLE001
            return
        if self.rank == 0:
            return
        t_end = self._time.time() + self.linger_s
        while self._time.time() < t_end:
            try:
                with self._lock:
                    if self.store.check(["abort"]):
                        return
            except Exception:   # noqa: BLE001
                return
            self._time.sleep(0.2)

    def report(self):
        rep = {"failed": {}, "phase": {}}
        for r in range(self.world):
            try:
                with self._lock:
                    if self.store.check(["fail_%d" % r]):
                        rep["failed"][str(r)] = self.store.get("fail_%d" % r).decode("utf-8", "replace")
                    if self.store.check(["phase_%d" % r]):
                        rep["phase"][str(r)] = self.store.get("phase_%d" % r).decode("utf-8", "replace")
            except Exception:   # noqa: BLE001
                pass
        return rep

    def _run(self):
        while not self._stop.wait(self.poll_s):
            try:
                if self.rank == 0:
                    with self._lock:
                        failed = any(self.store.check(["fail_%d" % r]) for r in range(self.world))
                        seen = tuple(self.store.get("phase_%d" % r) if self.store.check(["phase_%d" % r]) else None for r in range(self.world))
                    if seen != self._seen:                 # some rank moved on: the run is alive
                        self._seen = seen
                        self.deadline = self._time.time() + self.timeout_s
                    timed_out = self._time.time() > self.deadline
                    if failed or timed_out:
                        rep = self.report()
                        if timed_out and not failed:
                            rep["timeout"] = "no result after the rank timeout; ranks were in the phases listed"
                        try:
                            if self.on_abort is not None:
                                self.on_abort(rep)         # the line is out BEFORE any peer is released (a released peer exits, and
                        finally:                           # its closed connection may raise in this rank's main thread)
                            with self._lock:
                                self.store.set("abort", "1")
                            self._time.sleep(1.0)          # let the lingering ranks see 'abort'
                            self._os._exit(3)
                else:
                    with self._lock:
                        if self.store.check(["abort"]):
                            self._os._exit(3)
            except Exception:   # noqa: BLE001  (store gone: rank 0 has exited)
                if self.rank != 0:
                    return

    def close(self):
        self._stop.set()
