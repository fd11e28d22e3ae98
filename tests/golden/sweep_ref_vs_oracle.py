"""Index-exactness evidence: keep lists of the REFERENCE's own IoU arithmetic (oracle/_ref: kernel.cu:19-260
compiled from /root/reference, with libm's sincos) against this repo's restatement (oracle/riou_oracle.c, correctly rounded sincos --
2 % of its IoU values differ from the reference's in the last bits) over many box sets and thresholds, hunting for a pair whose IoU
straddles the threshold between the two definitions.

Runs only in the build container (needs /root/reference):   python tests/golden/sweep_ref_vs_oracle.py [--sets 72]
Writes tests/golden/rnms_sweep_ref_vs_oracle.json: per (distribution, threshold) the number of sets, boxes, and differing keep
lists; and, for every differing list, the first differing index with the offending pair's IoU under both definitions.
Distributions: uniform (the SURVEY 8(d) config-3 one at three densities), clustered (Gaussian clumps: many overlapping pairs),
near-duplicates (jittered copies of a few boxes: IoUs close to 1 and to every threshold), aligned (axis-parallel boxes on a grid:
collinear edges, the arithmetic's degenerate branch), thin (aspect ratios up to 64:1)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import riou  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rnms_sweep_ref_vs_oracle.json")
THRS = (0.1, 0.3, 0.5, 0.7)


def scores(rng, n):
    return ((rng.permutation(n) + 0.5) / n).astype(np.float32)


def gen(kind, seed):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        n = int(rng.choice([300, 1000, 3000]))
        d = riou.random_boxes(n, seed=seed, extent=float(rng.choice([120.0, 300.0, 608.0])))
    elif kind == "clustered":
        n, k = 1500, 12
        c = rng.uniform(50, 550, (k, 2))
        which = rng.integers(0, k, n)
        d = np.empty((n, 6), np.float32)
        d[:, :2] = c[which] + rng.normal(0, 14, (n, 2))
        d[:, 2] = 8.0 * 12.0 ** rng.uniform(0, 1, n)
        d[:, 3] = 8.0 * 12.0 ** rng.uniform(0, 1, n)
        d[:, 4] = rng.uniform(-np.pi / 2, np.pi / 2, n)
        d[:, 5] = scores(rng, n)
    elif kind == "near_duplicates":
        base = riou.random_boxes(40, seed=seed, extent=400.0)
        rep = 30
        d = np.repeat(base, rep, 0)
        n = len(d)
        amp = 10.0 ** rng.uniform(-3, 0.5, (n, 1))            # jitters from 1e-3 px to 3 px: IoUs from 1 - 1e-5 down to ~0.5
        d[:, :4] += (rng.normal(0, 1, (n, 4)) * amp * np.array([1, 1, 1, 1])).astype(np.float32)
        d[:, 2:4] = np.maximum(d[:, 2:4], 1.0)
        d[:, 4] += (rng.normal(0, 1, n) * amp[:, 0] * 0.01).astype(np.float32)
        d[:, 5] = scores(rng, n)
    elif kind == "aligned":
        n = 900
        d = np.empty((n, 6), np.float32)
        g = rng.integers(0, 40, (n, 2)).astype(np.float32)
        d[:, :2] = g * 8.0 + 20.0
        d[:, 2] = rng.choice([8.0, 16.0, 24.0, 32.0], n)
        d[:, 3] = rng.choice([8.0, 16.0, 24.0, 32.0], n)
        d[:, 4] = rng.choice([0.0, np.pi / 2, -np.pi / 2, np.pi / 4], n)
        d[:, 5] = scores(rng, n)
    elif kind == "thin":
        n = 1200
        d = riou.random_boxes(n, seed=seed, extent=260.0)
        d[:, 2] = 4.0 * 64.0 ** rng.uniform(0, 1, n)
        d[:, 3] = 4.0 * 4.0 ** rng.uniform(0, 1, n)
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(d, np.float32)


def first_difference(d, thr, a, b):
    """the first original index kept by one and not by the other, and the kept box (earlier in score order) whose IoU with it
    straddles the threshold"""
    sa, sb = set(a.tolist()), set(b.tolist())
    order = np.argsort(-d[:, 5], kind="stable")
    for pos, j in enumerate(order):
        if (j in sa) != (j in sb):
            kept_before = [i for i in order[:pos] if i in sa and i in sb]
            if not kept_before:
                return {"index": int(j)}
            rows = d[kept_before]
            i_ref = riou.riou_matrix(rows, d[j:j + 1], use_ref=True)[:, 0]
            i_orc = riou.riou_matrix(rows, d[j:j + 1])[:, 0]
            k = int(np.argmax((i_ref > thr) != (i_orc > thr)))
            return {"index": int(j), "suppressor": int(kept_before[k]), "iou_reference": float(i_ref[k]), "iou_oracle": float(i_orc[k]),
                    "box": d[j].tolist(), "suppressor_box": d[kept_before[k]].tolist()}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=72, help="box sets per distribution")
    a = ap.parse_args()
    assert riou.have_ref(), "needs oracle/_ref (built from /root/reference by oracle/Makefile)"
    rows, offenders, total_sets, total_lists, total_diff = [], [], 0, 0, 0
    for kind in ("uniform", "clustered", "near_duplicates", "aligned", "thin"):
        per_thr = {t: dict(lists=0, differing=0, boxes=0, kept=0) for t in THRS}
        for s in range(a.sets):
            d = gen(kind, 1000 * (1 + ["uniform", "clustered", "near_duplicates", "aligned", "thin"].index(kind)) + s)
            total_sets += 1
            for t in THRS:
                kr = riou.rnms(d, t, use_ref=True)
                ko = riou.rnms(d, t)
                r = per_thr[t]
                r["lists"] += 1
                r["boxes"] += len(d)
                r["kept"] += len(kr)
                if not np.array_equal(kr, ko):
                    r["differing"] += 1
                    info = first_difference(d, t, kr, ko) or {}
                    info.update(distribution=kind, seed=s, thr=t, n=len(d), kept_reference=len(kr), kept_oracle=len(ko))
                    offenders.append(info)
        for t in THRS:
            r = per_thr[t]
            rows.append(dict(distribution=kind, thr=t, **r))
            total_lists += r["lists"]
            total_diff += r["differing"]
            print("%-16s thr %.1f  %3d lists  %7d boxes  %6d kept  differing %d" % (kind, t, r["lists"], r["boxes"], r["kept"], r["differing"]),
                  flush=True)
    out = {"what": "keep lists of oracle/_ref (reference arithmetic, libm sincos) vs oracle/riou_oracle.c (correctly rounded sincos)",
           "sets": total_sets, "keep_lists": total_lists, "differing_keep_lists": total_diff, "per_distribution_and_threshold": rows,
           "offenders": offenders}
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1)
    print("%d sets, %d keep lists, %d differ -> %s" % (total_sets, total_lists, total_diff, OUT))


if __name__ == "__main__":
    main()
