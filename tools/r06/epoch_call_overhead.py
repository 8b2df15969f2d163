"""What one FusedTrainer.train_epoch call costs the DEVICE beyond its mini-batches (cached topology, batch 128): epochs of 32
mini-batches against the same number of mini-batches in fewer calls, and with the id upload taken out (ids already on the device).
    python tools/r06/epoch_call_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import deeprank_gnn_amd.synthetic as synth                      # noqa: E402
from deeprank_gnn_amd.resident import ResidentGraphSet          # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer               # noqa: E402
from deeprank_gnn_amd.ginet import GINet                        # noqa: E402

dev = torch.device("cuda:0")
G, B = 4096, 128
rs = ResidentGraphSet([synth.make_graph(i) for i in range(G)], dev)
torch.manual_seed(0)
tr = FusedTrainer(GINet(32, 1, 1).to(dev), lr=1e-3, task="reg")
gen = torch.Generator().manual_seed(0)


def run(calls, per_call, patch_upload=False):
    orders = [torch.cat([torch.randperm(G, generator=gen) for _ in range(per_call)]) for _ in range(calls)]
    real = rs.upload_ids
    if patch_upload:
        pre = {id(o): real(o.numpy()) for o in orders}
        torch.cuda.synchronize()
        it = iter(orders)
        rs.upload_ids = lambda ids: pre[id(cur[0])]
    cur = [None]
    try:
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for o in orders:
                cur[0] = o
                tr.train_epoch(rs, o, B, cached=True)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / (calls * per_call * (G // B)) * 1e6)
        return best
    finally:
        rs.upload_ids = real


tr.train_epoch(rs, torch.randperm(G, generator=gen), B, cached=True)
for calls, per_call in ((32, 1), (8, 4), (2, 16)):
    a = run(calls, per_call)
    b = run(calls, per_call, patch_upload=True)
    print("%2d calls x %2d passes (%4d mini-batches per call): %.2f us per mini-batch; ids already on the device: %.2f" % (
        calls, per_call, per_call * (G // B), a, b))
