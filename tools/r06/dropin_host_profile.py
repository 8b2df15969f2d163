"""Host-side cost of the eager drop-in loop body (reference NeuralNet.py:489-506) by part: wall time per call of model(batch), the
loss, backward and optimizer.step with the device kept busy (no synchronisation inside the loop), then cProfile of model(batch) +
backward.   python tools/r06/dropin_host_profile.py [GINet|sGAT|FoutNet]"""
import cProfile
import os
import pstats
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import deeprank_gnn_amd.synthetic as synth                      # noqa: E402
from deeprank_gnn_amd.ginet import GINet                        # noqa: E402
from deeprank_gnn_amd.sGAT import sGAT                          # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                    # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "GINet"
dev = torch.device("cuda:0")
batches = [synth.make_batch(64 * (i + 1), 64).to(dev) for i in range(32)]
torch.manual_seed(0)
net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[name](32, 1, 1).to(dev)
net.train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3)
acc = {"zero_grad": 0.0, "model": 0.0, "loss": 0.0, "backward": 0.0, "step": 0.0}


def body(b, timed):
    t0 = time.perf_counter()
    opt.zero_grad()
    t1 = time.perf_counter()
    pred = net(b)
    t2 = time.perf_counter()
    loss = F.mse_loss(pred.reshape(-1), b.y)
    t3 = time.perf_counter()
    loss.backward()
    t4 = time.perf_counter()
    opt.step()
    t5 = time.perf_counter()
    if timed:
        for k, d in zip(acc, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4)):
            acc[k] += d


for b in batches:
    body(b, False)
torch.cuda.synchronize()
n = 0
for _ in range(20):
    for b in batches:
        body(b, True)
        n += 1
torch.cuda.synchronize()
print(name, "host us per call:", {k: round(v / n * 1e6, 1) for k, v in acc.items()}, "sum", round(sum(acc.values()) / n * 1e6, 1))
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    for b in batches:
        opt.zero_grad()
        F.mse_loss(net(b).reshape(-1), b.y).backward()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
