"""Training steps on graphs beyond the staged kernels' LDS budget (round 6: the from-memory instances, plan.from_memory) against
the launch pair that stepped them before (plan override no_aggregate), with a staged shape of the same width for scale;
20 pipelined steps per hipGraph replay, us per step.    python tools/r06/big_graph_ab.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import deeprank_gnn_amd.synthetic as synth                      # noqa: E402
from deeprank_gnn_amd.data import Batch                          # noqa: E402
from deeprank_gnn_amd.topology import Topology                   # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer                # noqa: E402
from deeprank_gnn_amd.ginet import GINet                         # noqa: E402
from deeprank_gnn_amd.sGAT import sGAT                           # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                     # noqa: E402

dev = torch.device("cuda:0")


def timed(Net, n_feat, n_nodes, B, ov):
    shape = dict(n_nodes=n_nodes, n_pairs=(5 * n_nodes) // 2, n_c1=max(4, n_nodes // 12), n_internal=(7 * n_nodes) // 4)
    batch = Batch.from_data_list([synth.make_graph(i, n_feat=n_feat, **shape) for i in range(B)]).to(dev)
    torch.manual_seed(0)
    tr = FusedTrainer(Net(n_feat, 1, 1).to(dev), lr=1e-3, task="reg")
    tr.plan_overrides = dict(ov)
    need_w = Net is sGAT
    topos = [Topology.from_batch(batch, need_weights=need_w), Topology.from_batch(batch, need_weights=need_w)]
    plan = tr._plan_for(topos[0], n_feat, topos[1], True, batch.x)

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
    for _ in range(50):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 1000, plan, float(tr.loss)


def timed_cached(Net, n_feat, n_nodes, B):
    """the same step out of a resident set's cached topology (tiles formed by the stand-alone launch where the builder stages
    none): what NeuralNet.train runs by default"""
    from deeprank_gnn_amd.resident import ResidentGraphSet
    shape = dict(n_nodes=n_nodes, n_pairs=(5 * n_nodes) // 2, n_c1=max(4, n_nodes // 12), n_internal=(7 * n_nodes) // 4)
    rs = ResidentGraphSet([synth.make_graph(i, n_feat=n_feat, **shape) for i in range(B)], dev)
    cache = rs.topology_cache(need_weights=(Net is sGAT))
    torch.manual_seed(0)
    tr = FusedTrainer(Net(n_feat, 1, 1).to(dev), lr=1e-3, task="reg")
    ids = list(range(B))
    ids_dev = rs.upload_ids(ids)
    plan = tr._cached_prepare(cache, ids, ids_dev)["plan"]

    def chunk():
        for k in range(20):
            tr.train_step_cached(cache, ids, ids_dev)
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
    for _ in range(50):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / 1000, plan


if len(sys.argv) > 1 and sys.argv[1] == "cached":
    for Net, F, N, B in [(GINet, 48, 390, 128), (sGAT, 48, 380, 64), (FoutNet, 48, 380, 128), (sGAT, 64, 390, 64), (FoutNet, 64, 390, 64)]:
        us, plan = timed_cached(Net, F, N, B)
        print("%-8s F=%d, %3d nodes, batch %3d, cached topology          family %d wgs %d from_memory %d lds %6d B: %8.2f us per step = %5.2f M graphs/s" % (
            Net.__name__, F, N, B, plan.family, plan.wgs_per_graph, plan.from_memory, plan.lds_bytes, us, B / us), flush=True)
        us, plan, loss = timed(Net, F, N, B, {"no_aggregate": 1})
        print("%-8s F=%d, %3d nodes, batch %3d, launch pair (before)     family %d: %8.2f us per step = %5.2f M graphs/s" % (
            Net.__name__, F, N, B, plan.family, us, B / us), flush=True)
    sys.exit(0)

CASES = [(GINet, 32, 340, 64), (GINet, 32, 340, 128), (GINet, 48, 300, 128), (sGAT, 32, 340, 64), (FoutNet, 32, 340, 64),
         (sGAT, 48, 300, 128), (FoutNet, 48, 300, 128)]
for Net, F, N, B in CASES:
    rows = [({}, "default plan"), ({"no_aggregate": 1}, "launch pair (before)")]
    if Net is GINet:
        rows.insert(1, ({"force_wgs": 1}, "one workgroup per graph"))
    for ov, what in rows:
        us, plan, loss = timed(Net, F, N, B, ov)
        print("%-8s F=%d, %3d nodes, batch %3d, %-24s family %d wgs %d from_memory %d lds %6d B: %8.2f us per step = %5.2f M graphs/s   loss %.4f" % (
            Net.__name__, F, N, B, what, plan.family, plan.wgs_per_graph, plan.from_memory, plan.lds_bytes, us, B / us, loss), flush=True)
    us, plan, loss = timed(Net, F, 200, B, {})
    print("%-8s F=%d, 200 nodes, batch %3d, %-24s family %d wgs %d from_memory %d lds %6d B: %8.2f us per step" % (
        Net.__name__, F, B, "staged (for scale)", plan.family, plan.wgs_per_graph, plan.from_memory, plan.lds_bytes, us), flush=True)
