"""Host time of every drgnn_train_epoch call vs device time, long epochs (measurement tool)."""
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
orig = tr.api.train_epoch
calls = []
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); calls.append(time.perf_counter() - t0); return r
tr.api.train_epoch = timed
FusedTrainer.EPOCH_CHUNK = int(os.environ.get("CHUNK", "128"))
print("EPOCH_CHUNK", FusedTrainer.EPOCH_CHUNK)
for cached in (False, True):
    for fresh in (True,):
        fixed = torch.cat([torch.randperm(n, generator=gen) for _ in range(16)])
        mk = (lambda: torch.cat([torch.randperm(n, generator=gen) for _ in range(16)])) if fresh else (lambda: fixed)
        tr.train_epoch(rs, mk(), 64, cached=cached)[0].sum().item()
        for rep in range(3):
            del calls[:]
            torch.cuda.synchronize(); t0 = time.perf_counter()
            keep = [tr.train_epoch(rs, mk(), 64, cached=cached) for _ in range(4)]
            t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            nb = 4 * 1024
            print("cached=%s fresh_order=%s: %.2f us/batch total, host loop %.2f us/batch, C calls %.2f us/batch (max call %.1f ms of %d)" %
                  (cached, fresh, (t2 - t0) / nb * 1e6, (t1 - t0) / nb * 1e6, sum(calls) / nb * 1e6, max(calls) * 1e3, len(calls)), flush=True)
            del keep
