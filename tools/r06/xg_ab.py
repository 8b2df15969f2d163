"""64-feature sGAT / FoutNet training steps at batch 64: 200-node graphs through the x-from-memory form of the 64-wide kernels
(round 6) against the launch pair that stepped them before (plan override no_aggregate), and 120-node graphs through the staged
form; 20 pipelined steps per hipGraph replay, us per step.    python tools/r06/xg_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import deeprank_gnn_amd.synthetic as synth                      # noqa: E402
from deeprank_gnn_amd.data import Batch                          # noqa: E402
from deeprank_gnn_amd.topology import Topology                   # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer                # noqa: E402
from deeprank_gnn_amd.sGAT import sGAT                           # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                     # noqa: E402

dev = torch.device("cuda:0")


def timed(Net, n_nodes, ov):
    shape = dict(n_nodes=n_nodes, n_pairs=(5 * n_nodes) // 2, n_c1=max(4, n_nodes // 12), n_internal=(7 * n_nodes) // 4)
    batch = Batch.from_data_list([synth.make_graph(i, n_feat=64, **shape) for i in range(64)]).to(dev)
    torch.manual_seed(0)
    tr = FusedTrainer(Net(64, 1, 1).to(dev), lr=1e-3, task="reg")
    tr.plan_overrides = dict(ov)
    need_w = Net is sGAT
    topos = [Topology.from_batch(batch, need_weights=need_w), Topology.from_batch(batch, need_weights=need_w)]
    plan = tr._plan_for(topos[0], 64, topos[1], True, batch.x)

    def chunk():
        for k in range(20):
            tr.train_step(batch, topo=topos[k & 1], next_topo=topos[1 - (k & 1)])
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
    for _ in range(100):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 2000, plan, float(tr.loss)


for Net in (sGAT, FoutNet):
    for n_nodes, ov, what in ((200, {}, "x from memory (XG)"), (200, {"no_aggregate": 1}, "launch pair"), (120, {}, "staged tiles")):
        us, plan, loss = timed(Net, n_nodes, ov)
        print("%-8s 64 features, %3d nodes, %-20s family %d wgs %d lds %6d B: %7.2f us per step   loss %.4f" % (
            Net.__name__, n_nodes, what, plan.family, plan.wgs_per_graph, plan.lds_bytes, us, loss))
