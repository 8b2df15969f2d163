"""Whole-epoch throughput of the trainer loop on SYN graphs (BASELINE size): host collate per mini-batch (what the
reference's DataLoader does, NeuralNet.py:153-154,489-506) against the resident graph set + device collate.

usage: python tools/epoch_bench.py [--graphs 4096] [--batch 64] [--epochs 3] [--net GINet]
Prints one JSON line per mode: graphs/s over whole epochs (shuffle, collate, topology, step, update, outputs kept).
Modes: host-collate (Batch.from_data_list + .to(device) per mini-batch), resident (device collate, Python loop),
native-epoch (drgnn_train_epoch: the loop itself in the library)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import deeprank_gnn_amd.synthetic as synth                     # noqa: E402
from deeprank_gnn_amd import _lib                              # noqa: E402
from deeprank_gnn_amd.data import Batch                        # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                   # noqa: E402
from deeprank_gnn_amd.ginet import GINet                       # noqa: E402
from deeprank_gnn_amd.resident import ResidentGraphSet         # noqa: E402
from deeprank_gnn_amd.sGAT import sGAT                         # noqa: E402
from deeprank_gnn_amd.topology import Topology                 # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer              # noqa: E402


def epoch(trainer, batches, need_w):
    """The loop of deeprank_gnn_amd.NeuralNet._epoch: one-batch look-ahead, outputs kept, one sync at the end."""
    dev = trainer.flat_p.device
    running = torch.zeros((), device=dev)
    preds = []
    it = iter(batches)
    batch = next(it, None)
    topo = None if batch is None else Topology.from_batch(batch, need_weights=need_w)
    while batch is not None:
        nxt = next(it, None)
        nxt_topo = None if nxt is None else Topology.from_batch(nxt, need_weights=need_w, build=False)
        running += trainer.train_step(batch, topo=topo, next_topo=nxt_topo).reshape(())
        preds.append(trainer.last_pred.detach().clone())
        batch, topo = nxt, nxt_topo
    out = torch.cat(preds).cpu()
    return float(running), out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--net", choices=["GINet", "sGAT", "FoutNet"], default="GINet")
    ap.add_argument("--only", default=None, help="run a single mode (for profiling)")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    graphs = [synth.make_graph(i) for i in range(args.graphs)]
    Net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[args.net]
    need_w = args.net == "sGAT"
    results = {}
    modes = ("host-collate", "resident", "native-epoch") if args.only is None else (
        (args.only,) if args.only != "native-predict" else ())
    for mode in modes:
        torch.manual_seed(0)
        net = Net(32, 1, 1).to(dev)
        tr = FusedTrainer(net, lr=1e-3, task="reg")
        rs = ResidentGraphSet(graphs, dev) if mode != "host-collate" else None
        gen = torch.Generator().manual_seed(0)

        def batches():
            order = torch.randperm(args.graphs, generator=gen).tolist()
            if rs is not None:
                ids_dev = rs.upload_ids(order)
                for lo in range(0, len(order), args.batch):
                    yield rs.batch(order[lo:lo + args.batch], ids_dev[lo:lo + args.batch])
            else:
                for lo in range(0, len(order), args.batch):
                    yield Batch.from_data_list([graphs[i] for i in order[lo:lo + args.batch]]).to(dev)
        def native_epoch():
            order = torch.randperm(args.graphs, generator=gen).tolist()
            losses, pred = tr.train_epoch(rs, order, args.batch)
            out = pred.cpu()
            return float(losses.sum()), out
        run = native_epoch if mode == "native-epoch" else (lambda: epoch(tr, batches(), need_w))
        run()                                                  # warm-up epoch (allocator, descriptor caches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        losses = [run()[0] for _ in range(args.epochs)]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_batches = (args.graphs + args.batch - 1) // args.batch
        results[mode] = {"mode": mode, "net": args.net, "graphs_per_s": args.graphs * args.epochs / dt,
                         "us_per_batch": dt / (args.epochs * n_batches) * 1e6, "epochs": args.epochs,
                         "graphs": args.graphs, "batch": args.batch, "epoch_losses": losses}
        print(json.dumps(results[mode]))
    if args.only is None or args.only == "native-predict":
        # inference over the whole set in a fixed order (validation / test pass), native loop
        torch.manual_seed(0)
        tr = FusedTrainer(Net(32, 1, 1).to(dev), lr=1e-3, task="reg")
        rs = ResidentGraphSet(graphs, dev)
        order = list(range(args.graphs))
        tr.predict_epoch(rs, order, args.batch).cpu()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.epochs):
            out = tr.predict_epoch(rs, order, args.batch).cpu()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        n_batches = (args.graphs + args.batch - 1) // args.batch
        print(json.dumps({"mode": "native-predict", "net": args.net, "graphs_per_s": args.graphs * args.epochs / dt,
                          "us_per_batch": dt / (args.epochs * n_batches) * 1e6, "graphs": args.graphs,
                          "batch": args.batch, "pred_checksum": float(out.double().sum())}))
    if args.only is not None:
        return
    a, b, c = results["host-collate"], results["resident"], results["native-epoch"]
    print(json.dumps({"speedup_resident_over_host_collate": b["graphs_per_s"] / a["graphs_per_s"],
                      "speedup_native_epoch_over_host_collate": c["graphs_per_s"] / a["graphs_per_s"],
                      "same_losses": a["epoch_losses"] == b["epoch_losses"],
                      "native_epoch_losses_rel_diff": max(abs(x - y) / abs(x) for x, y in zip(a["epoch_losses"], c["epoch_losses"]))}))


if __name__ == "__main__":
    main()
