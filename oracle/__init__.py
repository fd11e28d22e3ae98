"""oracle/ -- CPU restatements of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package, and
only as the checker / the reported CPU baseline.  The product (rotate-yolov3_amd/) never does.
"""


def host_cores(cap=32):
    """Usable host cores: min(affinity, cgroup cpu quota, cap).  A container may report 256 logical CPUs while its
    cgroup grants a handful; spinning OpenMP/oneDNN teams sized by os.cpu_count() then crawl."""
    import math
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(math.ceil(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(math.ceil(q / per))))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, cap))
