"""Data-parallel training over the GPUs of one node: one process per GPU, graphs sharded
across ranks, ONE all-reduce of a flat fp32 gradient buffer per step (RCCL over xGMI when
the backend is "nccl"; gloo on CPU in the tests).

The reference has no distributed code at all; graphs of a mini-batch are
independent (block-diagonal adjacency), so the only exchange step is the gradient sum.
GINet(F=32,out=1) has 10 697 parameters = 42.8 KB: the collective is latency-bound, hence a
single bucket and no overlap machinery.
"""
import torch
import torch.distributed as dist

__all__ = ["FlatGradBucket", "shard_range"]


def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of ``n_items`` owned by ``rank`` (sizes differ by <= 1)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class FlatGradBucket(object):
    """Makes every parameter's ``.grad`` a view into one contiguous buffer.

    ``zero()`` clears all gradients with one fill; ``all_reduce(weight)`` sums the buffer over
    the ranks and rescales it so that the result is the gradient of the GLOBAL mean loss:
    each rank's loss is a mean over its own ``n_local`` graphs, so its gradient is weighted
    by ``n_local / n_global`` (equal shards -> 1/world)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        ref = self.params[0]
        self.flat = torch.zeros(total, dtype=ref.dtype, device=ref.device)
        off = 0
        for p in self.params:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            off += n

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, n_local=None, n_global=None, group=None):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        if n_local is not None and n_global:
            self.flat.mul_(float(n_local) / float(n_global))
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(dist.get_world_size(group))
