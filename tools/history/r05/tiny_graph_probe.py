"""What does replaying a TINY recorded graph (one step + one update = 2 kernel nodes) per training step cost against a graph of
32 steps?  (Would an epoch loop made of per-step replays of static-argument launches keep the argument blocks' L2 residency
without paying for graph boundaries?)  GINet cached, the same 64 graphs every step.
usage: python tools/r05/tiny_graph_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np                                            # noqa: E402
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd.resident import ResidentGraphSet        # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer             # noqa: E402
from deeprank_gnn_amd.ginet import GINet                      # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = FusedTrainer(GINet(32, 1, 1).to(dev), lr=1e-3, task="reg")
rs = ResidentGraphSet([synth.make_graph(i) for i in range(256)], dev)
cache = rs.topology_cache(need_weights=False)
ids = np.arange(64)
c = tr._cached_prepare(cache, ids, rs.upload_ids(ids), True, None)


def chunk(n):
    for _ in range(n):
        c["stream"] = torch.cuda.current_stream().cuda_stream
        tr._cached_launch_step(c, True)
        tr._fused_launch_update(c, True, lr=0.0)


side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    chunk(2)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
for steps in (32, 8, 2, 1):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chunk(steps)
    reps = 4096 // steps
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    print("graph of %2d steps: %.2f us per step on the device, host enqueue %.2f us per step" % (
        steps, e0.elapsed_time(e1) * 1e3 / (reps * steps), (t1 - t0) * 1e6 / (reps * steps)), flush=True)
