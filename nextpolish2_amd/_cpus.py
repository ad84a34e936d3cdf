"""CPUs this process may actually use: affinity mask and the cgroup's CFS quota (a container that sees 256 hardware
threads may be allowed 16 CPUs' worth of time; threads beyond that only get throttled)."""
import math
import os


def usable_cpus():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if q != "max" and int(per) > 0:
            n = min(n, max(1, math.ceil(int(q) / int(per))))
    except (OSError, ValueError):
        try:  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                n = min(n, max(1, math.ceil(q / per)))
        except (OSError, ValueError):
            pass
    return max(1, n)
