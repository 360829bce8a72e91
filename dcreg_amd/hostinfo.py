"""What the host side of a rank may use: CPUs granted by the cgroup (a container shows the machine's hardware threads in os.cpu_count()),
the share of one rank, NUMA-friendly pinning of the rank's threads."""
import os


def cgroup_cpu_quota():
    """CPUs the container may actually use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / per if q > 0 else None
    except Exception:
        return None


def usable_cpus():
    """min(affinity mask, cgroup quota): the CPUs this process can keep busy."""
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        n = os.cpu_count() or 1
    q = cgroup_cpu_quota()
    if q:
        n = min(n, max(1, int(q)))
    return max(1, n)


def threads_per_rank(local_world=1, cap=32):
    """Host threads one of `local_world` ranks sharing this box should use: its share of the usable CPUs, at least 1."""
    return max(1, min(cap, usable_cpus() // max(1, int(local_world))))


def pin_rank(local_rank, local_world):
    """Give rank `local_rank` a contiguous slice of the CPUs in this process's affinity mask (the spinning host thread and the
    OpenMP team of a rank stay on one NUMA node when the slices follow the CPU numbering).  No-op when the mask cannot be split."""
    try:
        cpus = sorted(os.sched_getaffinity(0))
    except Exception:
        return None
    if local_world <= 1 or len(cpus) < local_world:
        return None
    per = len(cpus) // local_world
    mine = cpus[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except Exception:
        return None
    return mine


def setup_rank(local_rank, local_world, cap=32):
    """Thread count and CPU slice of one rank, in the only order that works: the share is computed from the mask the process was
    STARTED with, then the mask is cut down to the rank's slice (usable_cpus() reads the affinity mask: asked after pin_rank it
    sees the slice and would divide by local_world a second time).  Returns (threads, pinned CPUs or None)."""
    want = threads_per_rank(local_world, cap)
    mine = pin_rank(local_rank, local_world)
    if mine:
        want = max(1, min(want, len(mine)))
    return want, mine
