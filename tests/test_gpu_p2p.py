"""One-shot all-reduce (csrc/drgnn_p2p.h) on the MI355X.  On ONE device the launches of several "ranks" are not
guaranteed to be co-resident (measured: the queues of one process were served one after the other, the waiting
workgroups expired), so the multi-rank protocol is driven in its two halves -- every rank publishes, then every rank
consumes -- which exercises the same code: slots, sequence counters, system-scope flags, fine-grained exchange buffers,
and (second test) their hipIpc mapping into another process.  The single-launch exchange runs with world 1 here and
with world N only on a multi-GPU node (bench.py with DRGNN_DP_ONESHOT=1)."""
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,n", [(1, 10697), (4, 10697), (8, 4273)])
def test_oneshot_allreduce_ranks_of_one_process(world, n):
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.parallel import OneShotAllReduce
    api = _lib.get()
    dev = torch.device("cuda:0")
    owns = [api.p2p_alloc(api.p2p_bytes(n)) for _ in range(world)]
    ptrs = [p for p, _ in owns]
    ranks = [OneShotAllReduce(n, dev, rank=r, world=world, own=owns[r], handles=ptrs) for r in range(world)]
    streams = [torch.cuda.Stream() for _ in range(world)]
    gen = torch.Generator().manual_seed(world)
    for step in range(4):
        host = [torch.randn(n, generator=gen) for _ in range(world)]
        vecs = [h.to(dev) for h in host]
        want = torch.zeros(n)
        for r in range(world):
            want += host[r] * (1.0 / world)
        torch.cuda.synchronize()
        if world == 1:
            ranks[0](vecs[0])                       # the single-launch exchange
        else:
            for part in (1, 2):
                for r in range(world):
                    with torch.cuda.stream(streams[r]):
                        ranks[r](vecs[r], part=part)
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        for r in range(world):
            ranks[r].check()
            assert torch.equal(vecs[r].cpu(), vecs[0].cpu())                       # bit-identical on every rank
            np.testing.assert_allclose(vecs[r].cpu().numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
    for r in ranks:
        r.own_ptr = None
    for p in ptrs:
        api.p2p_free(p)


def _ipc_worker(rank, world, init_file, out_dir):
    import torch.distributed as dist
    from deeprank_gnn_amd.parallel import OneShotAllReduce
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method="file://" + init_file, rank=rank, world_size=world)
    n = 10697
    ar = OneShotAllReduce(n, "cuda:0")              # exchanges the hipIpc handles through the process group
    gen = torch.Generator().manual_seed(100 + rank)
    outs = []
    for step in range(3):
        v = torch.randn(n, generator=gen).to("cuda:0")
        dist.barrier()
        ar(v, part=1)                               # publish into the own buffer (mapped in the other process) ...
        torch.cuda.synchronize()
        dist.barrier()
        ar(v, part=2)                               # ... read both buffers, the peer's through the IPC mapping
        torch.cuda.synchronize()
        ar.check()
        outs.append(v.cpu().numpy())
    np.save(os.path.join(out_dir, "o%d.npy" % rank), np.stack(outs))
    dist.barrier()
    ar.close()
    dist.destroy_process_group()


def test_oneshot_allreduce_over_hipipc_between_two_processes():
    """Two processes, one GPU: exchange buffers mapped through hipIpcGetMemHandle / hipIpcOpenMemHandle."""
    import torch.multiprocessing as mp
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_ipc_worker, args=(2, os.path.join(tmp, "rdv"), tmp), nprocs=2, join=True)
        o = [np.load(os.path.join(tmp, "o%d.npy" % r)) for r in range(2)]
    np.testing.assert_array_equal(o[0], o[1])
    gens = [torch.Generator().manual_seed(100 + r) for r in range(2)]
    for step in range(3):
        want = sum(torch.randn(10697, generator=g) * 0.5 for g in gens)
        np.testing.assert_allclose(o[0][step], want.numpy(), rtol=1e-6, atol=1e-6)
