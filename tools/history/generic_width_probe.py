"""Step time of the GENERIC-width fused step kernels (feature widths outside {16, 32, 48, 64}: XF = 0 instances), graph
replays of 20 steps, SYN graphs with `n_feat` features (measurement tool).  usage: generic_width_probe.py <n_feat> [nets]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd.topology import Topology
from deeprank_gnn_amd.trainer import FusedTrainer
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet

F = int(sys.argv[1]) if len(sys.argv) > 1 else 40
names = sys.argv[2:] or ["GINet", "sGAT", "FoutNet"]
dev = torch.device("cuda:0")
small = os.environ.get("SMALL", "0") != "0"       # 80-node graphs: room in LDS for wide features
batch = synth.make_batch(0, 64, n_feat=F, **(dict(n_nodes=80, n_pairs=200, n_internal=140) if small else {})).to(dev)
for name in names:
    torch.manual_seed(0)
    net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[name](F, 1, 1).to(dev)
    tr = FusedTrainer(net, lr=1e-3, seed=1)
    topos = [Topology.from_batch(batch, need_weights=(name == "sGAT")) for _ in range(2)]
    if not tr._can_fuse(topos[0], F):
        print("%s n_feat=%d: does not fit the fused step" % (name, F))
        continue
    variant = tr.api.step_is_specialised(tr.kind, batch.x, F, topos[0].max_nodes, topos[0].max_edges, topos[0].max_c0, tr.H, tr.O)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for k in range(4):
            tr.train_step(batch, topo=topos[k & 1], next_topo=topos[1 - (k & 1)])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for k in range(20):
            tr.train_step(batch, topo=topos[k & 1], next_topo=topos[1 - (k & 1)])
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    print("%s n_feat=%d kernel variant XF=%d: %.2f us per training step (loss %.4g, faults %d)"
          % (name, F, variant, e0.elapsed_time(e1) * 1e3 / 1000, float(tr.loss), tr.faults()), flush=True)
