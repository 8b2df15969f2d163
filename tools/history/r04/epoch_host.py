"""Host enqueue time vs device time per mini-batch of the native epoch loop (drgnn_train_epoch)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd.resident import ResidentGraphSet        # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer             # noqa: E402
from deeprank_gnn_amd.ginet import GINet                      # noqa: E402

dev = torch.device("cuda:0")
n_graphs, B = 4096, 64
graphs = [synth.make_graph(64 + i) for i in range(n_graphs)]
torch.manual_seed(0)
tr = FusedTrainer(GINet(32, 1, 1).to(dev), lr=1e-3, task="reg")
rs = ResidentGraphSet(graphs, dev)
gen = torch.Generator().manual_seed(0)
for cached in (False, True):
    for passes in (1, 16):
        def order():
            return torch.cat([torch.randperm(n_graphs, generator=gen) for _ in range(passes)])
        tr.train_epoch(rs, order(), B, cached=cached)[0].sum().item()
        nb = passes * n_graphs // B
        for rep in range(2):
            orders = [order() for _ in range(4)]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pend = [tr.train_epoch(rs, o, B, cached=cached) for o in orders]
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print("cached=%d passes=%2d: host enqueue %.2f us per batch, until device done %.2f us per batch" % (
                cached, passes, (t1 - t0) / (4 * nb) * 1e6, (t2 - t0) / (4 * nb) * 1e6), flush=True)
            del pend
