"""One-shot all-reduce (csrc/drgnn_p2p.h), protocol emulated on the CPU: W ranks of ONE process publish, then consume."""
import numpy as np
import pytest
import torch

from emu_api import emu
from deeprank_gnn_amd.parallel import OneShotAllReduce


@pytest.mark.parametrize("world,n", [(1, 100), (3, 10697), (8, 4273), (4, 5)])
def test_oneshot_allreduce_protocol(world, n):
    api = emu()
    owns = [api.p2p_alloc(api.p2p_bytes(n)) for _ in range(world)]
    handles = [h for _, h in owns]
    ranks = [OneShotAllReduce(n, "cpu", api=api, rank=r, world=world, own=owns[r], handles=handles) for r in range(world)]
    rng = np.random.default_rng(world)
    weights = rng.uniform(0.1, 1.0, size=world).astype(np.float32)
    for step in range(3):                        # three exchanges: both slots and the sequence counters
        vecs = [torch.from_numpy(rng.normal(size=n).astype(np.float32)) for _ in range(world)]
        want = torch.zeros(n)
        for r in range(world):                   # rank order, like the kernel
            want += vecs[r] * float(weights[r])
        if world == 1:
            ranks[0](vecs[0], weight=weights[0])
        else:
            for r in range(world):
                ranks[r](vecs[r], weight=weights[r], part=1)
            for r in range(world):
                ranks[r](vecs[r], weight=weights[r], part=2)
        for r in range(world):
            ranks[r].check()
            assert torch.equal(vecs[r], want), (step, r)
    # a rank that never published: the consumers flag it instead of hanging
    if world > 1:
        vecs = [torch.zeros(n) for _ in range(world)]
        for r in range(world - 1):
            ranks[r](vecs[r], part=1)
        ranks[0](vecs[0], part=2)
        with pytest.raises(Exception):
            ranks[0].check()
