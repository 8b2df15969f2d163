"""Host time of ONE FusedTrainer.train_epoch call (cached topology), the device kept out of the picture: the calls are enqueued
back to back and the host clock is read around each; then cProfile.   python tools/r06/epoch_host_profile.py [batch] [graphs]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import deeprank_gnn_amd.synthetic as synth                      # noqa: E402
from deeprank_gnn_amd.resident import ResidentGraphSet          # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer               # noqa: E402
from deeprank_gnn_amd.ginet import GINet                        # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
G = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
rs = ResidentGraphSet([synth.make_graph(i) for i in range(G)], dev)
torch.manual_seed(0)
tr = FusedTrainer(GINet(32, 1, 1).to(dev), lr=1e-3, task="reg")
gen = torch.Generator().manual_seed(0)
nb = (G + B - 1) // B
for cached in (True, False):
    tr.train_epoch(rs, torch.randperm(G, generator=gen), B, cached=cached)
    torch.cuda.synchronize()
    host = []
    t_all = time.perf_counter()
    for _ in range(20):
        order = torch.randperm(G, generator=gen)
        t0 = time.perf_counter()
        tr.train_epoch(rs, order, B, cached=cached)
        host.append(time.perf_counter() - t0)
    t_enq = time.perf_counter() - t_all
    torch.cuda.synchronize()
    t_dev = time.perf_counter() - t_all
    host.sort()
    print("cached=%s batch %d, %d mini-batches per epoch: host %.0f us per train_epoch call (median; %.2f us per mini-batch), "
          "20 epochs enqueued in %.2f ms, done in %.2f ms = %.2f us per mini-batch" % (
              cached, B, nb, host[10] * 1e6, host[10] * 1e6 / nb, t_enq * 1e3, t_dev * 1e3, t_dev * 1e6 / (20 * nb)))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    tr.train_epoch(rs, torch.randperm(G, generator=gen), B, cached=True)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
