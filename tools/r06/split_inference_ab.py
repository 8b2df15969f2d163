"""Same-box A/B of the single-branch nets' inference launches: node split (two workgroups per graph, round 6) against one workgroup
per graph (plan override no_split), FusedTrainer.predict_epoch over a resident set, topology cached and rebuilt, batch 64.
    python tools/r06/split_inference_ab.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import deeprank_gnn_amd.synthetic as synth                      # noqa: E402
from deeprank_gnn_amd.resident import ResidentGraphSet          # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer               # noqa: E402
from deeprank_gnn_amd.sGAT import sGAT                          # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                    # noqa: E402

dev = torch.device("cuda:0")
graphs = [synth.make_graph(64 + i) for i in range(4096)]
rs = ResidentGraphSet(graphs, dev)
order = torch.arange(4096)
for Net in (sGAT, FoutNet):
    torch.manual_seed(0)
    tr = FusedTrainer(Net(32, 1, 1).to(dev), lr=1e-3, task="reg")
    ref = None
    for label, ov in (("split", {}), ("whole", {"no_split": 1}), ("split", {}), ("whole", {"no_split": 1})):
        tr.plan_overrides = dict(ov)
        for cached in (True, False):
            tr.predict_epoch(rs, order, 64, cached=cached)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(4):
                    pred = tr.predict_epoch(rs, order, 64, cached=cached)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / (4 * 64) * 1e6)
            if ref is None:
                ref = pred.clone()
            err = float((pred - ref).abs().max())
            print("%-8s %-6s %-8s %.2f us per mini-batch of 64   max |diff to first| %.2e" % (Net.__name__, label, "cached" if cached else "rebuilt", best, err))
