"""Does the builder's output reach the next launch's step workgroups through the L2?  Pipelined GINet steps over 32 topology
WORKSPACES: (a) one mini-batch's tensors behind all of them (warm raw inputs, every workspace rewritten every 32 steps),
(b) 32 different mini-batches (cold raw inputs)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd.topology import Topology                # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer             # noqa: E402
from deeprank_gnn_amd.ginet import GINet                      # noqa: E402

dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = FusedTrainer(GINet(32, 1, 1).to(dev), lr=1e-3, task="reg")
STEPS = 32


def timed(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (100 * STEPS)


one = synth.make_batch(0, 64).to(dev)
for tag, bs in (("one mini-batch, 32 workspaces", [one] * 32), ("32 mini-batches, 32 workspaces", [synth.make_batch(64 * i, 64).to(dev) for i in range(32)])):
    tp = [Topology.from_batch(b, need_weights=False, build=(i == 0)) for i, b in enumerate(bs)]

    def piped():
        for k in range(STEPS):
            tr.train_step(bs[k % 32], topo=tp[k % 32], next_topo=tp[(k + 1) % 32])
    print("%-34s %.2f us per step" % (tag, timed(piped)), flush=True)
