"""Attribute the dispatches of the target kernel in rocprofv3 CSV traces to the phases of tools/inchain.py and print, per phase, the
average duration and (when the counter passes exist) effective clock and HBM bytes per launch.
    python tools/inchain_summary.py gpurun_out/inchain            (directories trace/ GRBM_GUI_ACTIVE/ FETCH_SIZE/ WRITE_SIZE/ below it)"""
import csv
import glob
import json
import os
import sys

KERNEL_OF = {"conv_mq<k3,128x256>": "conv_mq_kernel<0, 0, 64, 8, false", "conv_mp<k3,192x256>": "conv_mp_kernel<192, 0, 0>",
             "conv_mp<k3,256x256>": "conv_mp_kernel<256, 0, 0>"}


def rows_of(d, pattern):
    f = glob.glob(os.path.join(d, "**", pattern), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []


def phases(rows, info, key):
    """rows: dispatches of the target kernel in launch order -> {phase: [rows]}"""
    per_fwd, ti = info["launches_of_kernel_per_forward"], info["target_index_in_forward"]
    k = info["warm_forwards"] * per_fwd
    out = {"A isolated": rows[k:k + info["reps"]]}
    k += info["reps"]
    n_b = info["reps"] * (2 if info["pred_same_kernel"] else 1)
    b = rows[k:k + n_b]
    out["B alternating with its predecessor"] = b[1::2] if info["pred_same_kernel"] else b
    k += n_b
    c = rows[k:k + info["forwards"] * per_fwd]
    out["C in the forward"] = [c[i * per_fwd + ti] for i in range(info["forwards"]) if i * per_fwd + ti < len(c)]
    out["(all launches of this kernel in the forward)"] = c
    return out


def main():
    base = sys.argv[1]
    info = json.load(open(os.path.join(base, "inchain_phases.json")))
    kname = KERNEL_OF.get(info["target_kernel"], info["target_kernel"])
    print("# target: layer %d, %s (%s); predecessor: layer %d %s" % (info["target_layer"], info["target_kernel"], kname, info["predecessor_layer"], info["predecessor"]))
    res = {}
    tr = [r for r in rows_of(os.path.join(base, "trace"), "*kernel_trace.csv") if kname in r["Kernel_Name"]]
    tr.sort(key=lambda r: int(r["Start_Timestamp"]))
    for ph, rs in phases(tr, info, None).items():
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rs]
        if d:
            res.setdefault(ph, {})["us"] = sum(d) / len(d)
            res[ph]["n"] = len(d)
    for cnt in ("GRBM_GUI_ACTIVE", "FETCH_SIZE", "WRITE_SIZE"):
        rows = [r for r in rows_of(os.path.join(base, cnt), "*counter_collection.csv") if kname in r["Kernel_Name"] and r["Counter_Name"] == cnt]
        if not rows:
            continue
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        kt = {r["Dispatch_Id"]: r for r in rows_of(os.path.join(base, cnt), "*kernel_trace.csv")}
        for ph, rs in phases(rows, info, None).items():
            if not rs:
                continue
            v = [float(r["Counter_Value"]) for r in rs]
            res.setdefault(ph, {})[cnt] = sum(v) / len(v)
            if cnt == "GRBM_GUI_ACTIVE":
                if rs and rs[0].get("Start_Timestamp") and rs[0].get("End_Timestamp"):      # (the counter CSV carries the dispatch's own timestamps)
                    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs]
                else:
                    d = [(int(kt[r["Dispatch_Id"]]["End_Timestamp"]) - int(kt[r["Dispatch_Id"]]["Start_Timestamp"])) for r in rs if r["Dispatch_Id"] in kt]
                if d:
                    res[ph]["us_in_counter_pass"] = sum(d) / len(d) / 1e3
    print("%-46s %5s %9s %12s %14s %12s %12s" % ("phase", "n", "us", "us (pmc run)", "GUI_ACTIVE", "fetch MB x2", "write MB"))
    for ph, r in res.items():
        ga = r.get("GRBM_GUI_ACTIVE")
        print("%-46s %5d %9.2f %12s %14s %12s %12s" % (
            ph, r.get("n", 0), r.get("us", 0.0), "%.2f" % r["us_in_counter_pass"] if "us_in_counter_pass" in r else "-",
            "%.4g" % ga if ga else "-",
            "%.1f" % (2 * r["FETCH_SIZE"] / 1e3) if "FETCH_SIZE" in r else "-", "%.1f" % (r["WRITE_SIZE"] / 1e3) if "WRITE_SIZE" in r else "-"))
        if ga and "us_in_counter_pass" in r:
            print("%-46s       effective clock = GUI_ACTIVE / duration = %.3f GHz (per-instance sum / 8: %.3f)" % (
                "", ga / (r["us_in_counter_pass"] * 1e3), ga / 8 / (r["us_in_counter_pass"] * 1e3)))
    json.dump({"info": info, "phases": res}, open(os.path.join(base, "inchain_summary.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
