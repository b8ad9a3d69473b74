"""How many CPUs this process may actually use: the smaller of its affinity mask and its cgroup's CPU bandwidth quota.

A container can see every CPU of the host (os.cpu_count() = 256 on the MI355X boxes) while its cgroup grants a fraction
(cpu.max "1600000 100000" = 16 CPUs there).  A thread pool sized by cpu_count() then burns the quota in a few
milliseconds of spinning and the kernel parks the WHOLE process — including the thread that launches kernels — until the
100 ms period ends: seen as sporadic 15-70 ms gaps between steps (tools/diag_level_jitter3.py, profiles/r3_notes.md)."""
import math
import os


def usable_cpus():
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    quota = None
    try:                                            # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:                                        # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota is not None:
        n = min(n, max(1, math.floor(quota)))
    return max(1, n)


def limit_thread_pools(reserve=2):
    """Set OMP / MKL / torch intra-op pools to the usable CPUs minus `reserve` (the launching thread and the autograd
    thread), unless the environment already chose.  Call before the first parallel region; returns the pool size."""
    n = max(1, usable_cpus() - reserve)
    for var in ("OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(var, str(n))
    n = int(os.environ["OMP_NUM_THREADS"])
    try:
        import torch
        if torch.get_num_threads() > n:
            torch.set_num_threads(n)
    except ImportError:
        pass
    return n


def single_thread_backward():
    """Run autograd's backward on the calling thread (torch.autograd.set_multithreading_enabled(False), process-wide).

    With one GPU per process there is exactly one device thread to hand the graph to, so nothing runs in parallel anyway; the
    hand-off itself costs 0.10-0.14 ms per step on small scenes (C1 0.55 -> 0.41 ms, C2 0.57 -> 0.46 ms) and the device
    thread's wake-up occasionally takes a scheduler tick (2.6-4.8 ms, ~1 % of the steps on the shared MI355X boxes:
    profiles/r3_notes.md).  A caller-side setting: one line at the top of train.py for a drop-in user."""
    import torch
    torch.autograd.set_multithreading_enabled(False)
