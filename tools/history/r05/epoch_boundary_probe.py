"""What an epoch boundary of the native epoch loop costs on the device: runs pipelined shuffled epochs (as bench.py's
epoch_loop does) and, when run under `rocprofv3 --kernel-trace --memory-copy-trace`, tools/r05/epoch_boundary_report.py lists
everything between the last update launch of epoch e and the first step launch of epoch e+1.
usage: python tools/r05/epoch_boundary_probe.py [--cached] [--epochs 6] [--graphs 4096]"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cached", action="store_true")
    ap.add_argument("--epochs", type=int, default=6)
    ap.add_argument("--graphs", type=int, default=4096)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--net", default="GINet")
    args = ap.parse_args()
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.foutnet import FoutNet
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.sGAT import sGAT
    from deeprank_gnn_amd.trainer import FusedTrainer
    dev = torch.device("cuda:0")
    Net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[args.net]
    graphs = [synth.make_graph(64 + i) for i in range(args.graphs)]
    torch.manual_seed(0)
    tr = FusedTrainer(Net(32, 1, 1).to(dev), lr=1e-3, task="reg")
    rs = ResidentGraphSet(graphs, dev)
    gen = torch.Generator().manual_seed(0)

    # host time by part (wrapped callables; the device runs behind)
    acc = {}

    def wrap(obj, name, key):
        f = getattr(obj, name)

        def g(*a, **k):
            t = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                acc[key] = acc.get(key, 0.0) + time.perf_counter() - t
        setattr(obj, name, g)
    wrap(tr.api, "train_epoch_scratch_bytes", "scratch_bytes (C: carve)")
    wrap(tr.api, "train_epoch", "train_epoch (C: carve + launches)")
    wrap(rs, "upload_ids", "upload_ids")

    def enqueue():
        t = time.perf_counter()
        order = torch.randperm(args.graphs, generator=gen)
        acc["randperm"] = acc.get("randperm", 0.0) + time.perf_counter() - t
        t = time.perf_counter()
        out = tr.train_epoch(rs, order, args.batch, cached=args.cached)
        acc["train_epoch (python, all)"] = acc.get("train_epoch (python, all)", 0.0) + time.perf_counter() - t
        return out
    enqueue()[0].sum().item()
    nb = (args.graphs + args.batch - 1) // args.batch
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pending = [enqueue() for _ in range(args.epochs)]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("rep %d: %.2f us per mini-batch (%d epochs x %d mini-batches, %s)" % (
            rep, dt / (args.epochs * nb) * 1e6, args.epochs, nb, "cached" if args.cached else "rebuilt"), flush=True)
        del pending
        print("   host time per epoch, us: " + ", ".join("%s %.0f" % (k, v / args.epochs * 1e6) for k, v in acc.items()), flush=True)
        acc.clear()


if __name__ == "__main__":
    main()
