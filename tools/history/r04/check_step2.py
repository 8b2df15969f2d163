"""sGAT / FoutNet aggregation-first kernels (drgnn_step2.h) vs the oracle at SYN64, every layout, + timing A/B.
usage: python tools/r04/check_step2.py [nets...]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import torch
from oracle import cpu_ref
from elementwise import Lazy64, check_step, new_stats
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd import _lib
from deeprank_gnn_amd.topology import Topology
from deeprank_gnn_amd.trainer import FusedTrainer
from test_gpu_parity import build

dev = torch.device("cuda:0")
api = _lib.get()
nets = sys.argv[1:] or ["FoutNet", "sGAT"]
LAYOUTS_SINGLE = [("old", (8,)), ("af1", (7, 10)), ("af2", (7, 9))]
LAYOUTS_GINET = [("old", (12,)), ("af", (11,))]      # GINet: drgnn_step.h vs drgnn_step3.h
for net_name in nets:
    LAYOUTS = LAYOUTS_GINET if net_name == "GINet" else LAYOUTS_SINGLE
    kw = {"looped": False} if net_name == "FoutNet" else {}
    batch_cpu = synth.make_batch(0, 64)
    params = cpu_ref.init_params(net_name, 32, 1, 1, seed=11)
    ref_pred, ref_loss, ref_grads = cpu_ref.loss_and_grads(net_name, params, batch_cpu, batch_cpu.y, **kw)
    lazy = Lazy64(net_name, params, batch_cpu, **kw)
    batch = batch_cpu.clone().to(dev)
    need_w = net_name == "sGAT"
    for lname, modes in LAYOUTS:
        for m in modes:
            api.set_step_layout(m)
        net = build(net_name, params, 1)
        tr = FusedTrainer(net, lr=0.01, task="reg")
        topo = Topology.from_batch(batch, need_weights=need_w)
        nxt = Topology.from_batch(batch, need_weights=need_w, build=False)
        c = tr._fused_prepare(batch, topo)
        try:
            loss = tr.compute_gradients(batch, topo=topo, next_topo=nxt)
            torch.cuda.synchronize()
            grads = {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}
            stats = new_stats()
            check_step("%s %s" % (net_name, lname), lazy, loss, tr.last_pred.cpu().numpy(), grads, ref_loss, ref_pred.numpy(),
                       {k: v.numpy() for k, v in ref_grads.items()}, stats)
            tr.check_faults()
            print("%-8s %-4s slabs=%d split=%d PARITY OK loss %.6f (ref %.6f) arbiter %d/%d" % (
                net_name, lname, c["slabs"], c["hints"][0].split, float(loss), float(ref_loss), stats["arbiter"], stats["elements"]), flush=True)
        except Exception as e:
            print("%-8s %-4s FAILED: %s" % (net_name, lname, str(e)[:400]), flush=True)
            continue
        # timing: 200 eager steps (step + update), topologies ping-ponging
        topos = [topo, nxt]
        for it in range(20):
            tr.train_step(batch, topo=topos[it & 1], next_topo=topos[1 - (it & 1)])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 400
        for it in range(n):
            tr.train_step(batch, topo=topos[it & 1], next_topo=topos[1 - (it & 1)])
        torch.cuda.synchronize()
        print("%-8s %-4s eager %.2f us/step, loss after %d steps %.6f" % (net_name, lname, (time.perf_counter() - t0) / n * 1e6, n + 20, float(tr.loss)), flush=True)
    for m in (7, 9, 11):
        api.set_step_layout(m)

# ---- several Adam steps per layout vs the oracle + torch.optim.Adam (the benchmarked ping-pong schedule) ------------------
import torch.nn.functional as F
for net_name in nets:
    LAYOUTS = LAYOUTS_GINET if net_name == "GINet" else LAYOUTS_SINGLE
    kw = {"looped": False} if net_name == "FoutNet" else {}
    batch_cpu = synth.make_batch(0, 64)
    batch = batch_cpu.clone().to(dev)
    need_w = net_name == "sGAT"
    params = cpu_ref.init_params(net_name, 32, 1, 1, seed=12)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=0.01)
    ref_losses = []
    for it in range(6):
        opt.zero_grad()
        pred = cpu_ref.FORWARD[net_name](leaves, batch_cpu, **kw)
        loss = F.mse_loss(pred.reshape(-1), batch_cpu.y)
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    for lname, modes in LAYOUTS:
        for m in modes:
            api.set_step_layout(m)
        net = build(net_name, params, 1)
        tr = FusedTrainer(net, lr=0.01, task="reg")
        topos = [Topology.from_batch(batch, need_weights=need_w), Topology.from_batch(batch, need_weights=need_w)]
        got = []
        for it in range(6):
            got.append(float(tr.train_step(batch, topo=topos[it & 1], next_topo=topos[1 - (it & 1)])))
        sd = net.state_dict()
        err = max(float((sd[k].cpu() - v.detach()).abs().max()) for k, v in leaves.items())
        print("%-8s %-4s 6 Adam steps: losses %s vs ref %s  max |param - ref| %.2e" % (
            net_name, lname, ["%.4f" % v for v in got], ["%.4f" % v for v in ref_losses], err), flush=True)
    for m in (7, 9, 11):
        api.set_step_layout(m)
