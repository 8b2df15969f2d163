"""Quick GPU probe: eager GINet/sGAT/FoutNet train steps on SYN64, wall time per step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd.topology import Topology
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet

dev = torch.device("cuda:0")
batch = synth.make_batch(0, 64).to(dev)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for Net in (GINet, sGAT, FoutNet):
    torch.manual_seed(0)
    net = Net(32, 1, 1).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)
    def step():
        opt.zero_grad(set_to_none=True)
        topo = Topology.from_batch(batch)
        out = net(batch, topo=topo)
        loss = F.mse_loss(out.reshape(-1), batch.y)
        loss.backward()
        opt.step()
        return loss
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): l = step()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print("%s eager: %.1f us/step  %.0f graphs/s  loss %.4f" % (Net.__name__, dt * 1e6, 64 / dt, float(l)))
