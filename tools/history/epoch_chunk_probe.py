"""us per mini-batch of the native epoch loop vs epoch length and FusedTrainer.EPOCH_CHUNK (measurement tool)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd.resident import ResidentGraphSet
from deeprank_gnn_amd.trainer import FusedTrainer
from deeprank_gnn_amd.ginet import GINet
dev = torch.device("cuda:0")
n = 4096
graphs = [synth.make_graph(64 + i) for i in range(n)]
torch.manual_seed(0)
tr = FusedTrainer(GINet(32, 1, 1).to(dev), lr=1e-3, task="reg")
rs = ResidentGraphSet(graphs, dev)
gen = torch.Generator().manual_seed(0)
for cached in (True, False):
    for passes in (1, 4, 16):
        for chunk in (32, 128, 100000):
            FusedTrainer.EPOCH_CHUNK = chunk
            order = torch.cat([torch.randperm(n, generator=gen) for _ in range(passes)])
            tr.train_epoch(rs, order, 64, cached=cached)[0].sum().item()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            reps = max(1, 8 // passes)
            keep = [tr.train_epoch(rs, order, 64, cached=cached) for _ in range(reps)]
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            nb = reps * passes * 64
            print("cached=%s passes=%2d chunk=%6d: %.2f us/batch (host enqueue %.2f us/batch)" % (cached, passes, chunk, (t2 - t0) / nb * 1e6, (t1 - t0) / nb * 1e6), flush=True)
