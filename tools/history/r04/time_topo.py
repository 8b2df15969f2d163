"""GPU time per launch of the topology builder inside a hipGraph (20 launches per replay): full build vs TOPO_LEAN,
with / without edge weights, at a given batch size.  usage: python tools/r04/time_topo.py [B ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd import _lib                             # noqa: E402
from deeprank_gnn_amd.topology import Topology                # noqa: E402

dev = torch.device("cuda:0")
N = 20


def graph_of(fn):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(N):
            fn()
    return g


def timed(g, iters=50):
    for _ in range(5):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (iters * N)


for B in [int(a) for a in sys.argv[1:]] or [64]:
    batch = synth.make_batch(0, B).to(dev)
    for w in (False, True):
        topo = Topology.from_batch(batch, need_weights=w)
        out = []
        for name, flags in (("full", _lib.TOPO_HIER), ("full+tiles", _lib.TOPO_HIER | _lib.TOPO_TILES), ("no-hier", 0),
                            ("lean", _lib.TOPO_HIER | _lib.TOPO_LEAN), ("lean+tiles", _lib.TOPO_HIER | _lib.TOPO_LEAN | _lib.TOPO_TILES)):
            out.append("%s %.2f us" % (name, timed(graph_of(lambda: topo.rebuild(flags)))))
        print("k_topo B=%d weights=%d: %s" % (B, w, "   ".join(out)), flush=True)
