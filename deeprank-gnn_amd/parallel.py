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

from . import _lib

__all__ = ["FlatGradBucket", "shard_range", "OneShotAllReduce"]


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


class OneShotAllReduce(object):
    """One-shot all-reduce of a flat fp32 buffer over the GPUs of one node (csrc/drgnn_p2p.h): every rank publishes
    its weighted vector in a fine-grained exchange buffer that the peers have mapped through hipIpc, reads all W
    vectors over the point-to-point xGMI links and adds them in rank order -- one launch, one xGMI round trip instead
    of a ring's 2 (W-1) latency-bound hops for a 43 KB message, bit-identical sums on all ranks, hipGraph-capturable.

    ``OneShotAllReduce(n_floats, device)`` inside an initialised ``torch.distributed`` job exchanges the IPC handles
    with ``all_gather_object``; ``handles=[...]`` / ``rank`` / ``world`` set the peers explicitly (tests).  Opt-in
    (``DRGNN_DP_ONESHOT=1`` in bench.py): the default exchange stays RCCL's all-reduce."""

    def __init__(self, n_floats, device, api=None, group=None, rank=None, world=None, own=None, handles=None):
        self.api = api or _lib.get()
        self.n = int(n_floats)
        self.device = torch.device(device)
        nbytes = self.api.p2p_bytes(self.n)
        self.own_ptr, self.own_handle, self._opened, self.peers = None, None, [], []
        gathered = handles is None and dist.is_available() and dist.is_initialized()
        # Every rank makes the SAME sequence of collectives whatever fails locally: a rank whose allocation
        # was refused still takes part in the all_gather_object, with a failure marker instead of a handle; the error is
        # raised only after the gather, on every rank (so none is left waiting inside a collective).
        alloc_error = None
        try:
            if own is None:
                own = self.api.p2p_alloc(nbytes)
            self.own_ptr, self.own_handle = own
        except Exception as exc:
            if not gathered:
                raise
            alloc_error = exc
        if handles is None:
            if not gathered:
                rank, world, handles = 0, 1, [self.own_handle]
            else:
                rank, world = dist.get_rank(group), dist.get_world_size(group)
                handles = [None] * world
                dist.all_gather_object(handles, ("failed", repr(alloc_error)) if alloc_error is not None else self.own_handle,
                                       group=group)
                bad = [r for r, h in enumerate(handles) if isinstance(h, tuple) and h and h[0] == "failed"]
                if bad:
                    self.close()
                    raise _lib.DrgnnError("one-shot all-reduce: exchange buffer allocation failed on rank(s) %s: %s" %
                                          (bad, handles[bad[0]][1]))
        self.rank, self.world = int(rank), int(world)
        if self.world > 16:
            self.close()
            raise ValueError("at most 16 ranks")
        try:
            for r, h in enumerate(handles):
                if r == self.rank:
                    self.peers.append(self.own_ptr)
                elif isinstance(h, int):               # already a device pointer valid here (ranks of one process: tests)
                    self.peers.append(h)
                else:
                    ptr = self.api.p2p_open(h)
                    self.peers.append(ptr)
                    self._opened.append(ptr)
        except Exception:
            self.close()                               # no leaked mappings / exchange buffer on a refused hipIpc open
            raise
        self.seq = torch.zeros(16, dtype=torch.int32, device=self.device)
        self.status = torch.zeros(1, dtype=torch.int32, device=self.device)

    def __call__(self, flat, weight=None, part=0):
        """In place: flat <- sum_r weight_r * flat_r (``weight`` default 1 / world).  Enqueued on torch's current stream."""
        assert flat.dtype == torch.float32 and flat.is_contiguous() and flat.numel() == self.n
        w = (1.0 / self.world) if weight is None else float(weight)
        self.api.allreduce_oneshot(flat, self.n, self.peers, self.world, self.rank, w, self.seq, self.status,
                                   _lib.current_stream(flat), part=part)
        return flat

    def check(self):
        """Synchronising: raises when a wait expired (a peer never published its vector)."""
        st = int(self.status.item())
        if st:
            raise _lib.DrgnnError("one-shot all-reduce: the wait for rank %d expired" % (st - 1))

    def close(self):
        for ptr in self._opened:
            self.api.p2p_close(ptr)
        self._opened = []
        if self.own_ptr:
            self.api.p2p_free(self.own_ptr)
            self.own_ptr = None
