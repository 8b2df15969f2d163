"""Host-thread hygiene for the launching process.

The training loop is launch-bound on ONE host thread (about 20 us of submission work per mini-batch), and the small CPU tensor
ops around it (shuffling, concatenating, gathering index rows) go through torch's intra-op pool.  By default that pool has one
thread per VISIBLE core (256 on the MI355X hosts), its threads spin for a while after every parallel region, and a container
whose cgroup grants fewer CPUs than it shows (cpu.max "1600000 100000" = 16 CPUs on the measured boxes) then runs out of quota
within each 100 ms CFS period: every thread of the process -- the launching one included -- is frozen until the next period.
Measured (tools/history/throttle_probe.sh, profiles/r03_throttle_probe.txt): 1024-mini-batch epochs 38 - 45 us/batch with 13 throttled
periods (10.1 s of throttled thread time) at the default pool, 21.1 / 20.5 us/batch and no throttling with a 4-thread pool.

`fit_torch_threads()` sizes the pool to what the cgroup really grants (half the quota, shared between the ranks of the node),
once per process; NeuralNet and FusedTrainer call it.  An explicit OMP_NUM_THREADS, or DRGNN_KEEP_TORCH_THREADS=1, leaves the
pool alone."""
import os

import torch

_done = False


def cpu_quota():
    """CPUs the cgroup of this process may use per scheduling period (float), or None when it is unlimited / unreadable."""
    try:                                                            # cgroup v2
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()[:2]
        if quota != "max" and float(period) > 0:
            return float(quota) / float(period)
        return None
    except (OSError, ValueError):
        pass
    try:                                                            # cgroup v1
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fq, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fp:
            quota, period = float(fq.read()), float(fp.read())
        return quota / period if quota > 0 and period > 0 else None
    except (OSError, ValueError):
        return None


def granted_cpus():
    """min(cores in the affinity mask, cgroup quota), at least 1"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = cpu_quota()
    if quota is not None:
        n = min(n, max(1, int(quota)))
    return max(1, n)


def fit_torch_threads(force=False):
    """Shrink torch's intra-op pool to half the granted CPUs per local rank (never grows it).  Returns the pool size in use."""
    global _done
    if _done and not force:
        return torch.get_num_threads()
    _done = True
    if os.environ.get("OMP_NUM_THREADS") or os.environ.get("DRGNN_KEEP_TORCH_THREADS") == "1":
        return torch.get_num_threads()
    local_world = local_world_size()
    want = max(1, granted_cpus() // (2 * local_world))
    have = torch.get_num_threads()
    if have > want:
        torch.set_num_threads(want)
        # a process-wide side effect: say so once
        import logging
        logging.getLogger("deeprank_gnn_amd").info(
            "torch intra-op pool %d -> %d threads (%d CPUs granted, %d rank(s) on this node); OMP_NUM_THREADS or "
            "DRGNN_KEEP_TORCH_THREADS=1 keeps the pool", have, want, granted_cpus(), local_world)
    return torch.get_num_threads()


def local_world_size():
    """Ranks of this job on this node: torchrun's LOCAL_WORLD_SIZE, else what Open MPI / Slurm launchers export."""
    for name in ("LOCAL_WORLD_SIZE", "OMPI_COMM_WORLD_LOCAL_SIZE", "MPI_LOCALNRANKS", "SLURM_NTASKS_PER_NODE"):
        v = os.environ.get(name, "")
        try:
            if int(v.split("(")[0] or 0) > 0:
                return int(v.split("(")[0])
        except ValueError:
            continue
    return 1
