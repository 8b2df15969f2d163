"""Rebuilt-topology steps replayed from a hipGraph: (a) the same mini-batch every step, (b) a cycle of different mini-batches
(host-collated device tensors; every batch has a topology workspace of its own, built by the previous step's launch)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd.topology import Topology                # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer             # noqa: E402
from deeprank_gnn_amd.ginet import GINet                      # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "GINet"
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = FusedTrainer({"GINet": GINet, "FoutNet": FoutNet}[name](32, 1, 1).to(dev), lr=1e-3, task="reg")
STEPS = 32


def timed(batches):
    n = len(batches)
    topos = [Topology.from_batch(b, need_weights=False, build=(i == 0)) for i, b in enumerate(batches)]
    if n == 1:
        topos.append(Topology.from_batch(batches[0], need_weights=False, build=False))
        batches = batches * 2
        n = 2

    def chunk():
        for k in range(STEPS):
            tr.train_step(batches[k % n], topo=topos[k % n], next_topo=topos[(k + 1) % n])
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chunk()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chunk()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (20 * STEPS)


one = [synth.make_batch(0, 64).to(dev)]
many = [synth.make_batch(64 * i, 64).to(dev) for i in range(32)]
print("%s rebuilt, the same mini-batch every step      %.2f us per step" % (name, timed(one)), flush=True)
print("%s rebuilt, a cycle of 2 mini-batches           %.2f us per step" % (name, timed(many[:2])), flush=True)
print("%s rebuilt, a cycle of 32 different mini-batches %.2f us per step" % (name, timed(many)), flush=True)
