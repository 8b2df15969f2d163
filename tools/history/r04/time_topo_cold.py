"""k_topo (lean + tiles, own launch) replayed from a hipGraph: the same mini-batch every launch vs a cycle of 32 different ones,
and the step + update pair alone on warm / cold topologies (which side of the launch pays for distinct mini-batches?)."""
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
topos = [Topology.from_batch(b, need_weights=False, flags=FL) for b in batches]
torch.manual_seed(0)
tr = FusedTrainer(GINet(32, 1, 1).to(dev), lr=1e-3, task="reg")


def timed(fn, per):
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
    return e0.elapsed_time(e1) * 1e3 / (100 * per)


def builds(n):
    def f():
        for k in range(32):
            topos[k % n].rebuild(FL)
    return f


def steps(n):        # step + update on prebuilt topologies, no builder in the launch
    def f():
        for k in range(32):
            tr.train_step(batches[k % n], topo=topos[k % n])
    return f


print("k_topo lean+tiles, same mini-batch        %.2f us" % timed(builds(1), 32))
print("k_topo lean+tiles, cycle of 32            %.2f us" % timed(builds(32), 32))
print("step + update, same mini-batch/topology   %.2f us" % timed(steps(1), 32))
print("step + update, cycle of 32                %.2f us" % timed(steps(32), 32))
