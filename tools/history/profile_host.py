"""cProfile of the eager native training step's HOST side (launch overhead of FusedTrainer.train_step)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd.topology import Topology
from deeprank_gnn_amd.trainer import FusedTrainer
from deeprank_gnn_amd.ginet import GINet

dev = torch.device("cuda:0")
batch = synth.make_batch(0, 64).to(dev)
net = GINet(32, 1, 1).to(dev)
tr = FusedTrainer(net, lr=1e-3, seed=1)
topos = [Topology.from_batch(batch, need_weights=False), Topology.from_batch(batch, need_weights=False)]
for i in range(50):
    tr.train_step(batch, topo=topos[i & 1], next_topo=topos[1 - (i & 1)])
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
N = 2000
for i in range(N):
    tr.train_step(batch, topo=topos[i & 1], next_topo=topos[1 - (i & 1)])
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(22)
