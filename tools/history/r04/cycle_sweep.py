"""us per step of GINet replayed from a hipGraph over a cycle of N different mini-batches, pipelined (builder co-launched)
and on prebuilt topologies (no builder): where does the L2 stop helping?"""
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
batches = [synth.make_batch(64 * i, 64).to(dev) for i in range(32)]
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


for n in (1, 2, 4, 8, 16, 32):
    bs = batches[:n] if n > 1 else [batches[0], batches[0]]
    m = len(bs)
    tp = [Topology.from_batch(b, need_weights=False, build=(i == 0)) for i, b in enumerate(bs)]

    def piped():
        for k in range(STEPS):
            tr.train_step(bs[k % m], topo=tp[k % m], next_topo=tp[(k + 1) % m])
    a = timed(piped)
    tq = [Topology.from_batch(b, need_weights=False, flags=FL) for b in bs]

    def prebuilt():
        for k in range(STEPS):
            tr.train_step(bs[k % m], topo=tq[k % m])
    b = timed(prebuilt)
    print("cycle of %2d: pipelined %.2f us per step, prebuilt topologies %.2f" % (n, a, b), flush=True)
