"""What would a degree-sorted work assignment buy?  Times the fused step launch on the SYN64 batch as it is and on the same
graphs with their nodes RELABELLED by descending degree (x rows, edge ends and cluster0 permuted alike): the kernel then
sees lanes of similar trip counts side by side in its aggregation phases, which is what a degree-sorted permutation from
the builder would give it without relabelling.  Measurement tool only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd.data import Batch
from deeprank_gnn_amd.topology import Topology
from deeprank_gnn_amd.trainer import FusedTrainer
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet
from deeprank_gnn_amd import _lib

name = sys.argv[1] if len(sys.argv) > 1 else "GINet"
dev = torch.device("cuda:0")


def relabel(g):
    n = g.x.shape[0]
    deg = np.bincount(g.edge_index[0].numpy(), minlength=n)
    order = np.argsort(-deg, kind="stable")           # new position p holds old node order[p]
    inv = np.empty(n, dtype=np.int64); inv[order] = np.arange(n)
    g.x = g.x[torch.from_numpy(order)]
    g.pos = g.pos[torch.from_numpy(order)]
    g.cluster0 = g.cluster0[torch.from_numpy(order)]
    g.edge_index = torch.from_numpy(inv)[g.edge_index]
    g.internal_edge_index = torch.from_numpy(inv)[g.internal_edge_index]
    return g


def timed(batch):
    torch.manual_seed(0)
    net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[name](32, 1, 1).to(dev)
    tr = FusedTrainer(net, lr=1e-3, seed=1)
    topo = Topology.from_batch(batch, need_weights=(name == "sGAT"))
    assert topo.status()[0] == 0
    c = tr._fused_prepare(batch, topo)

    def fn():
        c["stream"] = _lib.current_stream(c["x"])
        tr._fused_launch_step(c, None)
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20):
            fn()
    for _ in range(5):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(50):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 1000, float(tr.loss)


plain = Batch.from_data_list([synth.make_graph(i) for i in range(64)]).to(dev)
sortd = Batch.from_data_list([relabel(synth.make_graph(i)) for i in range(64)]).to(dev)
for rep in range(2):
    a, la = timed(plain)
    b, lb = timed(sortd)
    print("%s step kernel: as generated %.2f us   nodes relabelled by descending degree %.2f us   (loss %.5f / %.5f)" % (name, a, b, la, lb))
