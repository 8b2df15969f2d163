"""1024-mini-batch epochs with / without a host synchronisation per epoch (measurement tool)."""
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
FusedTrainer.EPOCH_CHUNK = int(os.environ.get("CHUNK", "128"))
mk = lambda: torch.cat([torch.randperm(n, generator=gen) for _ in range(16)])
for cached in (False, True):
    tr.train_epoch(rs, mk(), 64, cached=cached)[0].sum().item()
    for sync in (True, False, True, False):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        keep = []
        for _ in range(4):
            done = tr.train_epoch(rs, mk(), 64, cached=cached)
            if sync:
                done[0].sum().item()
            else:
                keep.append(done)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print("chunk=%d cached=%s sync_per_epoch=%s: %.2f us/batch" % (FusedTrainer.EPOCH_CHUNK, cached, sync, (t2 - t0) / 4096 * 1e6), flush=True)
        del keep
