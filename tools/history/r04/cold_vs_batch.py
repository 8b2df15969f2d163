"""Cold-input penalty of the step (prebuilt topologies, no builder in the launch) at batch 16 / 32 / 64: a latency or a
bandwidth effect?  GINet, hipGraph replays, same mini-batch vs a cycle of 64 / 32 / 32 different ones."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd import _lib                             # noqa: E402
from deeprank_gnn_amd.topology import Topology                # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer             # noqa: E402
from deeprank_gnn_amd.ginet import GINet                      # noqa: E402

dev = torch.device("cuda:0")
FL = _lib.TOPO_HIER | _lib.TOPO_LEAN | _lib.TOPO_TILES
torch.manual_seed(0)
tr = FusedTrainer(GINet(32, 1, 1).to(dev), lr=1e-3, task="reg")
STEPS = 64


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
    for _ in range(60):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (60 * STEPS)


for B in (16, 32, 64):
    n = 2048 // B                      # every cycle covers 2048 graphs: the same total working set
    bs = [synth.make_batch(B * i, B).to(dev) for i in range(n)]
    tq = [Topology.from_batch(b, need_weights=False, flags=FL) for b in bs]

    def same():
        for k in range(STEPS):
            tr.train_step(bs[0], topo=tq[0])

    def cycle():
        for k in range(STEPS):
            tr.train_step(bs[k % n], topo=tq[k % n])
    a, b = timed(same), timed(cycle)
    print("B = %2d: same mini-batch %.2f us per step, cycle of %3d mini-batches %.2f  (+%.2f)" % (B, a, n, b, b - a), flush=True)
