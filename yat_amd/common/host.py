"""Host-side housekeeping of the trainer process (no reference counterpart: Accelerate leaves the thread count to torch).

torch sizes its intra-op (OpenMP) pool from ``os.cpu_count()`` -- the whole host, 256 on the MI355X boxes -- while a job is
usually confined to a slice of it (a cgroup CPU quota, an affinity mask; 16 CPUs per GPU on this pool).  With 128 - 256 OpenMP
threads on a 16-CPU quota every small CPU tensor op of the step's host thread (the staging copies, the bf16 noise draw, the
sampler's decode) fans out to threads that are throttled or descheduled: measured here, a 0.5 MB ``copy_`` takes 0.02 ms in a
tight loop and 30 - 64 ms after a few milliseconds of Python in between, the CPU oracle runs 5 - 25 x slower (tests/conftest.py),
and ``bench.py --data shards`` goes from 80 ms per step to 115 - 190 ms on a busy box -- the trainer becomes host-bound."""
import os


def usable_cores() -> int:
    """CPUs this process may actually use: min(affinity mask, cgroup v2 CPU quota)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cap_host_threads() -> int:
    """Cap torch's intra-op threads at this rank's share of the usable CPUs (usable / LOCAL_WORLD_SIZE, at least 1; at most 16:
    the host thread's ops are small).  An explicit ``OMP_NUM_THREADS`` (torchrun sets 1 for multi-process jobs) is left alone.
    -> the thread count in force."""
    import torch
    if "OMP_NUM_THREADS" not in os.environ:
        share = max(1, usable_cores() // max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1"))))
        n = min(torch.get_num_threads(), share, 16)
        if n != torch.get_num_threads():
            torch.set_num_threads(n)
    return torch.get_num_threads()
