"""Does the native epoch loop slow down when its mini-batches come from HBM instead of the Infinity Cache?  Same number of
mini-batches per epoch (long orders, per-epoch host work amortised), resident sets of different sizes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.resident import ResidentGraphSet
from deeprank_gnn_amd.trainer import FusedTrainer

dev = torch.device("cuda:0")
GMAX = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
all_graphs = [synth.make_graph(64 + i) for i in range(GMAX)]
gen = torch.Generator().manual_seed(0)
TOTAL = 16384
for cached in (True, False):
    for G in (128, 1024, GMAX):
        rs = ResidentGraphSet(all_graphs[:G], dev)
        torch.manual_seed(0)
        tr = FusedTrainer(GINet(32, 1, 1).to(dev), lr=1e-3, task="reg")
        nb = TOTAL // 64

        def epoch():
            order = torch.cat([torch.randperm(G, generator=gen) for _ in range(TOTAL // G)])
            losses, pred = tr.train_epoch(rs, order, 64, cached=cached)
            return float(losses.sum())
        epoch()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            epoch()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("cached=%s resident graphs=%d (%.0f MB of x + edges): %.2f us per mini-batch" %
              (cached, G, G * (200 * 32 * 4 + 1000 * 2 * 8 + 1000 * 4) / 1e6, dt / (3 * nb) * 1e6))
        del rs, tr
