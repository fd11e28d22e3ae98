"""Print the per-kernel summary (calls, total, average, share) of a rocprofv3 rocpd sqlite database.
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db [--pmc]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    print("# rocprofv3 --kernel-trace --stats summary of %s (durations in us)" % db.split("/")[-1])
    print("%-96s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "share"))
    for r in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"):
        if r[4] < 0.05:
            continue
        print("%-96s %8d %14.1f %12.2f %6.1f%%" % (r[0][:96], r[1], r[2], r[3], r[4]))
    if "--pmc" in sys.argv:
        q = ("select k.name, p.name, count(*), avg(e.value), sum(e.value) from pmc_events e "
             "join pmc_info p on e.pmc_id = p.id join kernels k on e.event_id = k.id group by k.name, p.name")
        try:
            for r in c.execute(q):
                print("PMC %-80s %-28s n=%5d avg=%.4g sum=%.6g" % (r[0][:80], r[1], r[2], r[3], r[4]))
        except sqlite3.Error as e:
            print("pmc query failed:", e)
            for t in ("pmc_events", "pmc_info", "counters_collection"):
                try:
                    print(t, [x[1] for x in c.execute("pragma table_info('%s')" % t)])
                except sqlite3.Error:
                    pass


if __name__ == "__main__":
    main()
