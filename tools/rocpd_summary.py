"""Print the per-kernel summary (calls, total, average, share) of a rocprofv3 rocpd sqlite database.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--pmc]"""
import sqlite3
import sys


def as_json(db, steps, out):
    """--json <steps> <out.json>: per-kernel table per STEP (calls and time divided by the number of traced steps)"""
    import json
    c = sqlite3.connect(db)
    rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"))
    tot = sum(r[2] for r in rows)
    ks = [{"kernel": r[0][:110], "launches_per_step": round(r[1] / steps, 2), "avg_us": round(r[3], 2),
           "ms_per_step": round(r[2] / 1e3 / steps, 3), "share": round(r[2] / tot, 4)} for r in rows if r[2] / tot >= 0.002]
    json.dump({"traced_steps": steps, "kernel_ms_per_step": round(tot / 1e3 / steps, 2), "kernels": ks}, open(out, "w"), indent=1)


def main():
    db = sys.argv[1]
    if "--json" in sys.argv:
        i = sys.argv.index("--json")
        return as_json(db, int(sys.argv[i + 1]), sys.argv[i + 2])
    c = sqlite3.connect(db)
    print("# rocprofv3 --kernel-trace --stats summary of %s (durations in us)" % db.split("/")[-1])
    print("%-96s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "share"))
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
        if r[4] < 0.05:
            continue
        print("%-96s %8d %14.1f %12.2f %6.1f%%" % (r[0][:96], r[1], r[2], r[3], r[4]))
    if "--pmc" in sys.argv:
        # counters_collection: one row per (dispatch, counter) with the value summed over the counter's instances
        q = "select kernel_name, counter_name, count(*), avg(value), sum(value) from counters_collection group by kernel_name, counter_name"
        try:
            for r in c.execute(q):
                print("PMC %-80s %-28s n=%5d avg=%.6g sum=%.6g" % (r[0][:80], r[1], r[2], r[3], r[4]))
        except sqlite3.Error as e:
            print("pmc query failed:", e)


if __name__ == "__main__":
    main()
