"""One variant of bench.py's dropin_loop on its own, for rocprofv3 --kernel-trace --stats: the reference trainer's loop body
(NeuralNet.py:489-506) over a cycle of 32 distinct SYN mini-batches, recorded in one hipGraph and replayed.
    python tools/r06/dropin_trace.py GINet 64 kept|rebuilt fused|foreach [replays]
Prints us per step (HIP events) and the step count, so that a kernel's calls / steps = launches per step."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import deeprank_gnn_amd.synthetic as synth                      # noqa: E402
from deeprank_gnn_amd.fused_autograd import engine_for          # noqa: E402
from deeprank_gnn_amd.ginet import GINet                        # noqa: E402
from deeprank_gnn_amd.sGAT import sGAT                          # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                    # noqa: E402

net_name, B, topo_mode, adam = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 100
dev = torch.device("cuda:0")
batches = [synth.make_batch(B * (i + 1), B).to(dev) for i in range(32)]
torch.manual_seed(0)
net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[net_name](32, 1, 1).to(dev)
net.train()
opt = torch.optim.Adam(net.parameters(), lr=1e-3, capturable=True, **({"fused": True} if adam == "fused" else {}))
eng = engine_for(net)
eng.cache_topology = topo_mode == "kept"


def body(b):
    opt.zero_grad()
    loss = F.mse_loss(net(b).reshape(-1), b.y)
    loss.backward()
    opt.step()
    return loss


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for b in batches:          # (every batch once: with `kept` its workspace is built here, outside the recording)
        body(b)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
opt.zero_grad(set_to_none=True)
with torch.cuda.graph(g):
    for b in batches:
        loss = body(b)
for _ in range(3):
    g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    g.replay()
e1.record()
torch.cuda.synchronize()
print("%s batch %d %s %s: %.2f us per step; steps replayed %d (+ %d warm-up/eager); path %s; loss %.4f" % (
    net_name, B, topo_mode, adam, e0.elapsed_time(e1) * 1e3 / (reps * 32), reps * 32, 3 * 32 + 32, eng.last_path, float(loss)))
