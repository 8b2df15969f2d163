"""Does the topology builder as a launch of its OWN, on a parallel branch of the recorded hipGraph, beat the co-launch?
(feasibility probe for a co-resident small-block builder).  SYN64, three schedules, 20 steps per replay:
  co    step launch carrying the next mini-batch's builder workgroups ; update        (what bench.py times)
  par   [step launch alone ; update]  ||  side stream: lean + tiles build of the next mini-batch's workspace (k_topo, own launch),
        forked behind the previous update, joined in front of the next step
  none  step launch alone ; update (no builder at all: the floor)
usage: python tools/r05/parallel_builder_probe.py [net] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd import _lib                             # noqa: E402
from deeprank_gnn_amd.topology import Topology                # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer             # noqa: E402
from deeprank_gnn_amd.ginet import GINet                      # noqa: E402
from deeprank_gnn_amd.sGAT import sGAT                        # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "GINet"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
torch.manual_seed(0)
batch = synth.make_batch(0, B).to(dev)
need_w = name == "sGAT"
net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[name](32, 1, 1).to(dev)


def timed(record, reps=300):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        record(warm=True)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        record(warm=False)
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * 20)


def run(mode):
    tr = FusedTrainer(net, lr=1e-3, task="reg", seed=1)
    topos = [Topology.from_batch(batch, need_weights=need_w) for _ in range(2)]
    co = mode == "co"
    cs = [tr._fused_prepare(batch, topos[i], True, topos[1 - i] if co else None) for i in range(2)]
    flags = tr._flags_for(topos[0], 32)
    if mode == "par":
        for t in topos:
            t.rebuild(flags)
        cs = [tr._fused_prepare(batch, topos[i], True, None) for i in range(2)]
    print(mode, "plan: wgs", cs[0]["plan"].wgs_per_graph, "family", cs[0]["plan"].family, "lean_ok", cs[0]["plan"].lean_ok,
          "flags", flags)
    builder = torch.cuda.Stream()

    def record(warm):
        main = torch.cuda.current_stream()
        for i in range(20):
            c = cs[i & 1]
            c["stream"] = _lib.current_stream(c["x"])
            if mode == "par":
                # the next step's workspace is rebuilt on the side stream while this step and its update run
                builder.wait_stream(main)
                with torch.cuda.stream(builder):
                    topos[1 - (i & 1)].rebuild(flags)
            tr._fused_launch_step(c, topos[1 - (i & 1)] if co else None)
            tr._fused_launch_update(c, True)
            if mode == "par":
                main.wait_stream(builder)
    us = timed(record)
    torch.cuda.synchronize()
    assert tr.faults() == 0
    print("%-5s %s B=%d  %.2f us per step   loss %.6g" % (mode, name, B, us, float(tr.loss)))


for m in ("co", "par", "none", "co", "par"):
    run(m)
