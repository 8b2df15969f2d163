"""GPU time per launch of the fused step kernel inside a hipGraph (no host launch cost): 20 launches per replay."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd.topology import Topology
from deeprank_gnn_amd.trainer import FusedTrainer
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet
from deeprank_gnn_amd import _lib

tag = sys.argv[1] if len(sys.argv) > 1 else "-"
name = sys.argv[2] if len(sys.argv) > 2 else "GINet"
dev = torch.device("cuda:0")
batch = synth.make_batch(0, 64).to(dev)
torch.manual_seed(0)
net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[name](32, 1, 1).to(dev)
# (layout A/Bs: DRGNN_STEP_PLAN=product|nosplit|one|two|noclass|seq, read by the library at its first plan query)
tr = FusedTrainer(net, lr=1e-3, seed=1)
topo = Topology.from_batch(batch, need_weights=(name == "sGAT"))
nxt = Topology.from_batch(batch, need_weights=(name == "sGAT"), build=False)
c = tr._fused_prepare(batch, topo, True, nxt)
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


def on_current(fn):
    def run():
        c["stream"] = _lib.current_stream(c["x"])      # the capture stream while capturing
        fn()
    return run


t_step = timed(graph_of(on_current(lambda: tr._fused_launch_step(c, None))))
t_co = timed(graph_of(on_current(lambda: tr._fused_launch_step(c, nxt))))
t_upd = timed(graph_of(on_current(lambda: tr._fused_launch_update(c, True, lr=0.0))))
t_red = timed(graph_of(on_current(lambda: tr._fused_launch_update(c, False))))
print("graph %6s %s  step %.2f us   step+topo %.2f us   update %.2f us   update w/o Adam %.2f us"
      % (tag, name, t_step, t_co, t_upd, t_red), flush=True)
