"""Ablation / schedule-variant timing of the multi-phase conv tile (csrc/conv_mp.hip) on the ABLATION build of the library
(__graft_entry__.build_ablation(): -DRYOLO_MP_ABLATION; several variants compute wrong results on purpose -- the shipped
library does not contain them).  Run on the GPU box:
    python tools/mp_ablate.py [--exp variants,cap,skew,trace] [--bs 32]
Experiments:
  variants  product vs no-stores / no-epilogue / store cache policies, with and without the fused shortcut
  cap       the same tiles-per-workgroup depth on a grid capped to 64 / 128 workgroups (smaller batch): do the workgroups
            slow each other down (synchronised store bursts, L2 / fabric contention)?
  skew      workgroups of an XCD start (loc & 3) * D cycles apart
  trace     s_memtime stamps around the K loop and the epilogue passes of the first tiles of three workgroups
  data      the same launches on random / sparse / constant / all-zero operands: the instruction stream is identical, only the
            bits that toggle in the MFMA datapath (and with them the chip's power and sustained clock) differ
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("RYOLO_HIP_LIB", os.path.join(ROOT, "rotate-yolov3_amd", "libryolo_hip_ablation.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import rotate_yolov3_amd  # noqa: E402,F401
from rotate_yolov3_amd import _lib  # noqa: E402
from rotate_yolov3_amd.model import hip_ops as ops  # noqa: E402

L = _lib.lib()
L.ryolo_debug_conv_trace.argtypes = [C.c_void_p]
L.ryolo_debug_conv_trace.restype = None
L.ryolo_debug_conv_set.argtypes = [C.c_int, C.c_int]
L.ryolo_debug_conv_set.restype = None
L.ryolo_debug_conv_variant.argtypes = [C.c_int, C.c_int]
L.ryolo_debug_conv_variant.restype = None
L.ryolo_debug_convq_trace.argtypes = [C.c_void_p]
L.ryolo_debug_convq_trace.restype = None
L.ryolo_debug_convq_set.argtypes = [C.c_int, C.c_int]
L.ryolo_debug_convq_set.restype = None
dev = torch.device("cuda:0")

# VAR bits (conv_mp.hip): 8 no stores, 16 no epilogue, 32 start skew, 64 stores nt, 512 stores sc1, 1024 epilogue trace
VARS = {"prod": 0, "nostore": 8, "noepi": 16, "skew": 32, "skew_nostore": 40, "nt": 64, "sc1": 512, "trace": 1024, "trace_skew": 1056,
        "trace_nostore": 1032, "skew_noepi": 48, "prio_half": 2, "prio_toggle": 4, "trace_prio_half": 1026, "trace_prio_toggle": 1028,
        "noepi_prio_toggle": 20, "prio_burst": 256, "noepi_prio_burst": 272}
SLOT = {}      # variant name -> one of the library's 16 variant slots, assigned on first use (an experiment uses fewer than 16)


def slot_of(name):
    if name not in SLOT:
        if len(SLOT) >= 16:
            SLOT.clear()
        SLOT[name] = len(SLOT)
        L.ryolo_debug_conv_variant(SLOT[name], VARS[name])
    return SLOT[name]


def qtile(name):
    """tile code of conv_mq.hip (two 4-wave workgroups per CU) with ablation variant `name`"""
    return 9 if name == "prod" else 64 + slot_of(name)


def tile_of(name, bm=256):
    if name == "prod":
        return 8 if bm == 256 else 11
    return 32 + slot_of(name) + (16 if bm == 192 else 0)


def make(bs, cin, cout, hw, k=3, residual=False, data="random"):
    x = torch.randn(bs, hw, hw, cin, device=dev).clamp_(-3, 3).to(torch.bfloat16)
    x = torch.where(x > 0, x, x * 0.1)
    w = torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5
    if data == "zeros":
        x.zero_(); w.zero_()
    elif data == "ones":
        x.fill_(1.0); w.fill_(1.0 / (cin * k * k))
    elif data == "sparse90":                   # 90 % of the activations exactly zero (what a ReLU net would feed), weights random
        x = torch.where(torch.rand(x.shape, device=dev) < 0.9, torch.zeros_like(x), x)
    elif data == "zero_x":
        x.zero_()
    packed = ops.pack_weights(w, cin_pad=cin)
    sc = torch.ones(ops.cpad(cout), device=dev)
    sh = torch.zeros(ops.cpad(cout), device=dev)
    out = torch.empty(bs, hw, hw, cout, device=dev, dtype=torch.bfloat16)
    res = torch.randn(bs, hw, hw, cout, device=dev).to(torch.bfloat16) if residual else None
    flop = 2.0 * k * k * cin * cout * hw * hw * bs

    def run(tile):
        ops.conv2d_bn_act(x, packed, sc, sh, cout, k, act=1, out=out, residual=res, tile=tile)
    run.out = out
    return run, flop


def time_tiles(run, tiles, rounds=7, reps=10):
    times = {t: [] for t in tiles}
    for t in tiles:
        run(t)
    torch.cuda.synchronize()
    for _ in range(rounds):
        for t in tiles:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run(t)
            e1.record()
            torch.cuda.synchronize()
            times[t].append(e0.elapsed_time(e1) / reps * 1e3)
    return {t: (sorted(v)[len(v) // 2], min(v)) for t, v in times.items()}


def report(label, names, res, flop, bm=256):
    line = "%-34s|" % label
    for n in names:
        med, mn = res[tile_of(n, bm)]
        line += " %s %6.1f us %5.0f TF |" % (n, med, flop / med / 1e6)
    print(line, flush=True)


SHAPES = [(128, 256, 76), (256, 512, 38), (512, 1024, 19)]


def exp_variants(bs):
    names = ["prod", "nostore", "noepi", "nt", "sc1"]
    for cin, cout, hw in SHAPES:
        for resid in (False, True):
            for bm in (256, 192):
                run, flop = make(bs, cin, cout, hw, residual=resid)
                res = time_tiles(run, [tile_of(n, bm) for n in names])
                report("%d->%d@%d bs%d BM%d %s" % (cin, cout, hw, bs, bm, "res" if resid else "   "), names, res, flop, bm)


def exp_data(bs):
    for cin, cout, hw in SHAPES:
        line = "%d->%d@%d bs%d |" % (cin, cout, hw, bs)
        for data in ("random", "sparse90", "ones", "zero_x", "zeros"):
            run, flop = make(bs, cin, cout, hw, data=data)
            tiles = [qtile("prod"), qtile("noepi"), tile_of("prod", 256), tile_of("noepi", 256)]
            res = time_tiles(run, tiles, rounds=5, reps=20)
            line += " %s: mq %5.1f us %4.0f TF, K loop only %4.0f TF; mp256 %4.0f TF, K loop only %4.0f TF |" % (
                data, res[tiles[0]][0], flop / res[tiles[0]][0] / 1e6, flop / res[tiles[1]][0] / 1e6, flop / res[tiles[2]][0] / 1e6,
                flop / res[tiles[3]][0] / 1e6)
        print(line, flush=True)


def exp_cap(bs):
    # same depth of the tile list per workgroup on a smaller grid: cap C workgroups, batch bs * C / 256
    names = ["prod", "nostore", "noepi"]
    for cin, cout, hw in SHAPES[:2]:
        for cap in (256, 128, 64, 32):
            b = max(1, bs * cap // 256)
            L.ryolo_debug_conv_set(1, cap if cap < 256 else 0)
            run, flop = make(b, cin, cout, hw, residual=True)
            res = time_tiles(run, [tile_of(n) for n in names])
            report("%d->%d@%d bs%d grid<=%d res" % (cin, cout, hw, b, cap), names, res, flop)
        L.ryolo_debug_conv_set(1, 0)


def exp_skew(bs):
    names = ["prod", "skew", "nostore", "skew_nostore"]
    cin, cout, hw = SHAPES[0]
    run, flop = make(bs, cin, cout, hw, residual=True)
    for d in (0, 5000, 10000, 20000, 40000):
        L.ryolo_debug_conv_set(0, d)
        res = time_tiles(run, [tile_of(n) for n in names])
        report("%d->%d@%d bs%d skew unit %d cyc" % (cin, cout, hw, bs, d), names, res, flop)
    L.ryolo_debug_conv_set(0, 0)


def exp_trace(bs):
    buf = torch.zeros(3 * 8 * 32, dtype=torch.int32, device=dev)
    L.ryolo_debug_conv_trace(buf.data_ptr())
    cin, cout, hw = SHAPES[0]
    for resid in (False, True):
        for name, d in (("trace", 0), ("trace_nostore", 0), ("trace_skew", 20000)):
            L.ryolo_debug_conv_set(0, d)
            run, flop = make(bs, cin, cout, hw, residual=resid)
            for _ in range(3):
                buf.zero_()
                run(tile_of(name))
            torch.cuda.synchronize()
            tr = buf.cpu().numpy().astype(np.int64).reshape(3, 8, 32) & 0xffffffff
            print("trace %s residual=%s: per tile [K loop | residual requests | pass 0 | pass 1 | tail], cycles; start = first stamp relative to workgroup 0 wave 0"
                  % (name, resid))
            t00 = tr[0, 0, 0]
            for wg in range(3):
                for wv in (0, 3, 4, 7):
                    s = tr[wg, wv]
                    parts = []
                    for t in range(4):
                        e = s[t * 8:t * 8 + 6]
                        if e[5] == 0:
                            break
                        parts.append("%6d|%5d|%5d|%5d|%4d" % (e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3], e[5] - e[4]))
                        if t + 1 < 4 and s[(t + 1) * 8] != 0:
                            parts[-1] += " gap %5d" % (s[(t + 1) * 8] - e[5])
                    print("  wg %d wave %d start %7d : %s" % (wg, wv, (s[0] - t00) & 0xffffffff, "  ||  ".join(parts)))
    L.ryolo_debug_conv_set(0, 0)


def mq_trace(run_maker, names, bs):
    buf = torch.zeros(3 * 8 * 32, dtype=torch.int32, device=dev)
    L.ryolo_debug_convq_trace(buf.data_ptr())
    cin, cout, hw = SHAPES[0]
    for resid in (False, True):
        for name in names:
            run, flop = make(bs, cin, cout, hw, residual=resid)
            for _ in range(3):
                buf.zero_()
                run(qtile(name))
            torch.cuda.synchronize()
            tr = buf.cpu().numpy().astype(np.int64).reshape(3, 8, 32) & 0xffffffff
            print("mq trace %s residual=%s: per tile [K loop | residual requests | pass 0 | pass 1 | tail] gap" % (name, resid))
            for wg in range(3):
                for wv in (0, 3):
                    s = tr[wg, wv]
                    parts = []
                    for t in range(4):
                        e = s[t * 8:t * 8 + 6]
                        if e[5] == 0:
                            break
                        parts.append("%6d|%5d|%5d|%5d|%4d" % (e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3], e[5] - e[4]))
                        if t + 1 < 4 and s[(t + 1) * 8] != 0:
                            parts[-1] += " gap %5d" % (s[(t + 1) * 8] - e[5])
                    print("  wg %d wave %d start %10d end %10d : %s" % (wg, wv, s[0], max(s), "  ||  ".join(parts)))


def exp_mqprio(bs):
    for cin, cout, hw in SHAPES:
        for resid in (False, True):
            run, flop = make(bs, cin, cout, hw, residual=resid)
            names = ["mp256", "mp192", "mq", "prio_half", "prio_toggle", "prio_burst", "noepi", "noepi_prio_toggle", "noepi_prio_burst"]
            tiles = [8, 11, 9, qtile("prio_half"), qtile("prio_toggle"), qtile("prio_burst"), qtile("noepi"), qtile("noepi_prio_toggle"),
                     qtile("noepi_prio_burst")]
            res = time_tiles(run, tiles)
            line = "%d->%d@%d bs%d %s |" % (cin, cout, hw, bs, "res" if resid else "   ")
            for nm, t in zip(names, tiles):
                line += " %s %6.1f us %5.0f TF |" % (nm, res[t][0], flop / res[t][0] / 1e6)
            print(line, flush=True)
    if "--trace" in sys.argv:
        mq_trace(make, ["trace_prio_half", "trace_prio_toggle"], bs)


def exp_traffic(bs):
    """K loop with and without operand traffic (tile bit 0x400 of the ablation build: every load out of range = zeros, no L2 / HBM
    traffic, same instruction stream)"""
    for cin, cout, hw in SHAPES:
        run, flop = make(bs, cin, cout, hw, residual=False)
        names = ["mp_noepi", "mp_noepi_notraffic", "mq_noepi", "mq_noepi_notraffic", "mp", "mp_notraffic", "mq", "mq_notraffic"]
        tiles = [tile_of("noepi"), tile_of("noepi") | 0x400, qtile("noepi"), qtile("noepi") | 0x400, 8, 8 | 0x400, 9, 9 | 0x400]
        res = time_tiles(run, tiles)
        line = "%d->%d@%d bs%d |" % (cin, cout, hw, bs)
        for nm, t in zip(names, tiles):
            line += " %s %6.1f us %5.0f TF |" % (nm, res[t][0], flop / res[t][0] / 1e6)
        print(line, flush=True)


def exp_mq(bs):
    """conv_mq.hip against conv_mp.hip: bit-equality of the outputs, then timing with the second-half workgroups delayed"""
    for cin, cout, hw in SHAPES:
        for resid in (False, True):
            run, flop = make(bs, cin, cout, hw, residual=resid)
            run(8)
            ref = run.out.clone()
            same = []
            for rep_ in range(3):
                run.out.zero_()
                run(9)
                torch.cuda.synchronize()
                same.append(bool(torch.equal(run.out, ref)))
            bad = int((run.out != ref).sum())
            tiles = [8, 11, 9, qtile("nostore"), qtile("noepi")]
            res = time_tiles(run, tiles)
            line = "%d->%d@%d bs%d %s | equal to conv_mp: %s (%d differ) |" % (cin, cout, hw, bs, "res" if resid else "   ", same, bad)
            for nm, t in zip(["mp256", "mp192", "mq", "mq_nostore", "mq_noepi"], tiles):
                line += " %s %6.1f us %5.0f TF |" % (nm, res[t][0], flop / res[t][0] / 1e6)
            print(line, flush=True)
    cin, cout, hw = SHAPES[0]
    for resid in (False, True):
        run, flop = make(bs, cin, cout, hw, residual=resid)
        for d in (0, 8000, 16000, 24000, 32000):
            L.ryolo_debug_convq_set(0, d)
            tiles = [9, qtile("skew"), qtile("skew_nostore"), qtile("skew_noepi")]
            res = time_tiles(run, tiles)
            line = "%d->%d@%d bs%d %s second-half workgroups %5d cycles late |" % (cin, cout, hw, bs, "res" if resid else "   ", d)
            for nm, t in zip(["mq", "skew", "skew_nostore", "skew_noepi"], tiles):
                line += " %s %6.1f us %5.0f TF |" % (nm, res[t][0], flop / res[t][0] / 1e6)
            print(line, flush=True)
    L.ryolo_debug_convq_set(0, 0)
    buf = torch.zeros(3 * 8 * 32, dtype=torch.int32, device=dev)
    L.ryolo_debug_convq_trace(buf.data_ptr())
    for resid in (False, True):
        for name, d in (("trace", 0), ("trace_skew", 16000)):
            L.ryolo_debug_convq_set(0, d)
            run, flop = make(bs, cin, cout, hw, residual=resid)
            for _ in range(3):
                buf.zero_()
                run(qtile(name))
            torch.cuda.synchronize()
            tr = buf.cpu().numpy().astype(np.int64).reshape(3, 8, 32) & 0xffffffff
            print("mq trace %s residual=%s: per tile [K loop | residual requests | pass 0 | pass 1 | tail] gap" % (name, resid))
            for wg in range(3):
                for wv in (0, 3):
                    s = tr[wg, wv]
                    parts = []
                    for t in range(4):
                        e = s[t * 8:t * 8 + 6]
                        if e[5] == 0:
                            break
                        parts.append("%6d|%5d|%5d|%5d|%4d" % (e[1] - e[0], e[2] - e[1], e[3] - e[2], e[4] - e[3], e[5] - e[4]))
                        if t + 1 < 4 and s[(t + 1) * 8] != 0:
                            parts[-1] += " gap %5d" % (s[(t + 1) * 8] - e[5])
                    print("  wg %d wave %d start %10d : %s" % (wg, wv, s[0], "  ||  ".join(parts)))
    L.ryolo_debug_convq_set(0, 0)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--exp", default="variants,cap,skew,trace")
    ap.add_argument("--bs", type=int, default=32)
    ap.add_argument("--trace", action="store_true")
    a = ap.parse_args()
    for e in a.exp.split(","):
        print("==== %s" % e, flush=True)
        SLOT.clear()
        {"variants": exp_variants, "cap": exp_cap, "skew": exp_skew, "trace": exp_trace, "mq": exp_mq, "mqprio": exp_mqprio, "traffic": exp_traffic, "data": exp_data}[e](a.bs)
