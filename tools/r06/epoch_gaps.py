"""Where the device waits inside the native epoch loop: kernel start / end stamps of a rocprofv3 --kernel-trace run of a few
cached-topology epochs, gaps between consecutive kernels listed when they exceed 4 us.
    rocprofv3 --kernel-trace -d /tmp/eg -o run --output-format csv -- python tools/r06/epoch_gaps.py run 128
    python tools/r06/epoch_gaps.py parse /tmp/eg"""
import csv
import glob
import os
import sys

if sys.argv[1] == "run":
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.trainer import FusedTrainer
    from deeprank_gnn_amd.ginet import GINet
    B = int(sys.argv[2])
    dev = torch.device("cuda:0")
    rs = ResidentGraphSet([synth.make_graph(i) for i in range(4096)], dev)
    tr = FusedTrainer(GINet(32, 1, 1).to(dev), lr=1e-3, task="reg")
    gen = torch.Generator().manual_seed(0)
    for _ in range(3):
        tr.train_epoch(rs, torch.randperm(4096, generator=gen), B, cached=True)
    torch.cuda.synchronize()
    for _ in range(6):
        tr.train_epoch(rs, torch.randperm(4096, generator=gen), B, cached=True)
    torch.cuda.synchronize()
else:
    f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-(6 * 2 * 32 + 40):]
    prev = None
    total_gap = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if prev is not None:
            gap = (s - prev[1]) / 1e3
            if gap > 4.0:
                print("gap %7.2f us   after %-28s (%.2f us)   before %-28s (%.2f us)" % (
                    gap, prev[2][:28], (prev[1] - prev[0]) / 1e3, r["Kernel_Name"][:28], (e - s) / 1e3))
        prev = (s, e, r["Kernel_Name"])
    span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / 1e3
    print("last %d kernels: span %.1f us, busy %.1f us" % (len(rows), span, busy))
    names = {}
    for r in rows:
        d = names.setdefault(r["Kernel_Name"][:40], [0, 0.0])
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    for k, (n, t) in sorted(names.items(), key=lambda kv: -kv[1][1]):
        print("  %-42s x %4d  avg %.2f us" % (k, n, t / n))
