"""Per-kernel HBM traffic of the bs-64 train step: join the FETCH_SIZE / WRITE_SIZE counter passes (rocprofv3 --pmc, CSV output, one pass each)
of `bench.py --mode train` by kernel name and print, per kernel, launches per step, average microseconds (from the counter runs' own
timestamps: profiled clocks), fetched (x2, MI355X_MICROARCH.md) and written MB per launch and the HBM rate they amount to.
    python tools/step_traffic.py gpurun_out/step_traffic <steps traced>"""
import collections
import csv
import glob
import os
import sys


def load(d, counter):
    f = glob.glob(os.path.join(d, counter, "**", "*counter_collection.csv"), recursive=True)
    rows = collections.OrderedDict()
    if not f:
        return rows
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        e = rows.setdefault(k, [0, 0.0, 0.0])
        e[0] += 1
        e[1] += float(r["Counter_Value"])
        if r.get("Start_Timestamp") and r.get("End_Timestamp"):
            e[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return rows


def main():
    d, steps = sys.argv[1], float(sys.argv[2])
    fe, wr = load(d, "FETCH_SIZE"), load(d, "WRITE_SIZE")
    print("# per-kernel HBM traffic of the bs-64 train step (%g traced steps; FETCH_SIZE x 2 KB, WRITE_SIZE KB; durations under the counter passes)" % steps)
    print("%-100s %8s %9s %10s %10s %8s" % ("kernel", "n/step", "avg us", "fetch MB", "write MB", "TB/s"))
    tot_f = tot_w = tot_us = 0.0
    out = []
    for k in fe:
        n, fv, us = fe[k]
        wv = wr.get(k, [0, 0.0, 0.0])[1]
        fmb, wmb = 2 * fv * 1024 / 1e6 / n, wv * 1024 / 1e6 / max(wr.get(k, [1])[0], 1)
        avg = us / n if us else 0.0
        out.append((n / steps * (fmb + wmb), k, n / steps, avg, fmb, wmb))
        tot_f += 2 * fv * 1024 / 1e6 / steps
        tot_w += wv * 1024 / 1e6 / steps
        tot_us += us / steps
    for _, k, n, avg, fmb, wmb in sorted(out, reverse=True):
        if n * (fmb + wmb) < 20:
            continue
        print("%-100s %8.1f %9.1f %10.1f %10.1f %8.2f" % (k[:100], n, avg, fmb, wmb, (fmb + wmb) / avg / 1e0 / 1e0 * 1e-0 / 1e0 if avg else 0.0))
    print("total per step: fetched %.1f GB, written %.1f GB, kernel time %.1f ms -> %.2f TB/s averaged over the step" % (
        tot_f / 1e3, tot_w / 1e3, tot_us / 1e3, (tot_f + tot_w) / tot_us if tot_us else 0.0))


if __name__ == "__main__":
    main()
