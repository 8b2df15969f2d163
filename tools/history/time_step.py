"""HIP-event time of the fused step launch (no co-launched topology) and of the update launch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd.topology import Topology
from deeprank_gnn_amd.trainer import FusedTrainer
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet

tag = sys.argv[1] if len(sys.argv) > 1 else "-"
name = sys.argv[2] if len(sys.argv) > 2 else "GINet"
dev = torch.device("cuda:0")
batch = synth.make_batch(0, 64).to(dev)
torch.manual_seed(0)
net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[name](32, 1, 1).to(dev)
tr = FusedTrainer(net, lr=1e-3, seed=1)
topo = Topology.from_batch(batch, need_weights=(name == "sGAT"))
nxt = Topology.from_batch(batch, need_weights=(name == "sGAT"), build=False)
assert tr._can_fuse(topo, 32)
c = tr._fused_prepare(batch, topo)


def timed(fn, iters=300):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


t_step = timed(lambda: tr._fused_launch_step(c, None))
t_co = timed(lambda: tr._fused_launch_step(c, nxt))
t_upd = timed(lambda: tr._fused_launch_update(c, True, lr=0.0))
print("skip %4s  %s  step %.2f us   step+topo %.2f us   update %.2f us" % (tag, name, t_step, t_co, t_upd), flush=True)
