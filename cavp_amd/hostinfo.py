"""Host-side housekeeping shared by bench.py, the tests and the tools: how many CPUs this process may really use.

The MI355X boxes expose 256 logical CPUs to a container that runs under a 16-CPU cgroup quota.  torch sizes its intra-op pool by
the logical count (128 threads), so every CPU-side parallel region - the oracle in the tests, `torch.randperm` / numpy in
`ContrastLoss`, the DropPath draws - leaves ~128 spinning workers sharing 16 CPUs with the Python thread that issues the HIP
launches: eager steps ran 2 .. 4 x slower (config #5: 50 .. 95 ms per step, 23 .. 26 ms with the pool capped)."""
import os
from typing import Optional, Tuple


def cpu_quota() -> Optional[int]:
    """CPUs available to this process: min(cgroup quota (v2 cpu.max, v1 cfs_quota_us), affinity mask); None = no limit known."""
    lim = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
        if q != "max":
            lim = max(1, int(int(q) / int(per)))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = int(f.read())
            if q > 0:
                lim = max(1, q // per)
        except (OSError, ValueError):
            pass
    try:
        aff = len(os.sched_getaffinity(0))
        lim = aff if lim is None else min(lim, aff)
    except (AttributeError, OSError):
        pass
    return lim


def cap_torch_threads() -> Tuple[int, Optional[int]]:
    """Shrink torch's intra-op pool to the quota (never grows it); returns (threads now, quota)."""
    import torch
    q = cpu_quota()
    if q is not None and torch.get_num_threads() > q:
        torch.set_num_threads(q)
    return torch.get_num_threads(), q
