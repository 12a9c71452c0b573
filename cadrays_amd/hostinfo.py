"""What the host gives this process (pure Python, no torch / HIP import)."""
import os


def usable_cpus():
    """CPUs this process can actually run on at once: the smaller of os.cpu_count(), the scheduler affinity mask and the cgroup CPU
    quota (cpu.max of cgroup v2, cfs_quota of v1).  On the pool's GPU boxes os.cpu_count() says 256 hardware threads while the container
    is throttled to 16 CPUs of time: 256 OpenMP threads there are SLOWER than 16 (8.7 vs 11.5 Mrays/s for the CPU baseline on C3)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n
