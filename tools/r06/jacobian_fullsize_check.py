"""Where does the drop-in (jacobian) step of a full-size mini-batch (64 x 200 nodes) leave the oracle's gradients?
Compares, per parameter tensor, against the CPU oracle: the drop-in call on the engine's own workspace, on an explicit full
workspace, under plan overrides, the launch pair, and FusedTrainer.compute_gradients.   usage: python tools/r06/jacobian_fullsize_check.py [net]"""
import os
import sys
import numpy as np
import torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import importlib
importlib.import_module("deeprank_gnn_amd")
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd.topology import Topology
from deeprank_gnn_amd.fused_autograd import engine_for
from deeprank_gnn_amd.trainer import FusedTrainer
from test_gpu_parity import build, cpu_ref

net_name = sys.argv[1] if len(sys.argv) > 1 else "GINet"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
batch_cpu = synth.make_batch(0, B)
params = cpu_ref.init_params(net_name, 32, 1, 1, seed=5)
kw = {"looped": False} if net_name == "FoutNet" else {}
ref_pred, ref_loss, ref_grads = cpu_ref.loss_and_grads(net_name, params, batch_cpu, batch_cpu.y, **kw)
batch = batch_cpu.clone().to(dev)
need_w = net_name == "sGAT"


def report(tag, pred, grads):
    worst = []
    for k, r in ref_grads.items():
        g = grads[k].detach().cpu().numpy()
        r = r.numpy()
        bad = np.abs(g - r) > 1e-4 + 1e-4 * np.abs(r)
        worst.append((int(bad.sum()), float(np.abs(g - r).max() / (np.abs(r).max() + 1e-30)), k))
    nb = sum(w[0] for w in worst)
    w = max(worst)
    dp = float(np.abs(pred.detach().cpu().numpy().reshape(-1) - ref_pred.numpy().reshape(-1)).max())
    print("%-44s pred max diff %.2e | elements outside 1e-4: %5d | worst tensor %s (%d, max diff / max ref %.2e)" % (
        tag, dp, nb, w[2], w[0], w[1]), flush=True)


def dropin(tag, topo_fn=None, overrides=None, order=None):
    net = build(net_name, params, 1)
    eng = engine_for(net)
    if overrides:
        eng.plan_overrides = dict(overrides)
    net.zero_grad(set_to_none=True)
    out = net(batch, topo=topo_fn()) if topo_fn else net(batch)
    path = eng.last_path
    F.mse_loss(out.reshape(-1), batch.y).backward()
    torch.cuda.synchronize()
    p = eng.last_plan
    report("%s [%s wgs %d cls %d]" % (tag, path, p.wgs_per_graph if p else -1, p.cls if p else -1), out,
           {k: v.grad for k, v in net.named_parameters()})


dropin("drop-in, engine's workspace")
dropin("drop-in, explicit full workspace", lambda: Topology.from_batch(batch, need_weights=need_w))
dropin("drop-in, no_class", None, {"no_class": 1})
dropin("drop-in, one workgroup per graph", None, {"force_wgs": 1} if net_name == "GINet" else {"no_split": 1})
dropin("drop-in, launch pair", None, {"no_aggregate": 1})
net = build(net_name, params, 1)
net.train()
tr = FusedTrainer(net, lr=1e-3, task="reg", seed=1)
tr.compute_gradients(batch, topo=Topology.from_batch(batch, need_weights=need_w))
torch.cuda.synchronize()
report("FusedTrainer.compute_gradients", tr.last_pred, {k: v.grad for k, v in net.named_parameters()})
