"""cProfile of the DROP-IN path's host side: the nets as torch modules inside a plain PyTorch loop (model(batch), MSE loss,
loss.backward(), torch.optim.Adam) -- what a user of the reference's own trainer gets."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd.ginet import GINet

dev = torch.device("cuda:0")
batch = synth.make_batch(0, 64).to(dev)
net = GINet(32, 1, 1).to(dev)
opt = torch.optim.Adam(net.parameters(), lr=1e-3)


def step():
    opt.zero_grad()
    pred = net(batch)
    loss = F.mse_loss(pred.reshape(-1), batch.y)
    loss.backward()
    opt.step()
    return loss


for i in range(30):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
N = 300
for i in range(N):
    step()
torch.cuda.synchronize()
print("%.1f us per step (wall)" % ((time.perf_counter() - t0) / N * 1e6))
pr = cProfile.Profile()
pr.enable()
for i in range(N):
    step()
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(35)
