"""Is the epoch loop's extra time per mini-batch cold data?  Cached-topology steps replayed from a hipGraph with (a) the same 64
graphs every step, (b) 64 random graphs of a small pool, (c) 64 random graphs of the whole 4096-graph set."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np                                            # noqa: E402
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd.resident import ResidentGraphSet        # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer             # noqa: E402
from deeprank_gnn_amd.ginet import GINet                      # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "GINet"
dev = torch.device("cuda:0")
n_graphs, B, STEPS = 4096, 64, 64
rs = ResidentGraphSet([synth.make_graph(64 + i) for i in range(n_graphs)], dev)
cache = rs.topology_cache(need_weights=False)
torch.manual_seed(0)
tr = FusedTrainer({"GINet": GINet, "FoutNet": FoutNet}[name](32, 1, 1).to(dev), lr=1e-3, task="reg")
rng = np.random.default_rng(0)


def timed(id_lists):
    devs = [rs.upload_ids(np.asarray(i, dtype=np.int32)) for i in id_lists]

    def chunk():
        for ids, d in zip(id_lists, devs):
            tr.train_step_cached(cache, ids, d)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chunk()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        chunk()
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (20 * len(id_lists))


same = [list(range(64))] * STEPS
small = [rng.choice(256, size=B, replace=False).tolist() for _ in range(STEPS)]
sorted_all = [sorted(rng.choice(n_graphs, size=B, replace=False).tolist()) for _ in range(STEPS)]
full = [rng.choice(n_graphs, size=B, replace=False).tolist() for _ in range(STEPS)]
contig = [list(range(s, s + 64)) for s in rng.choice(n_graphs - 64, size=STEPS).tolist()]
for tag, lists in (("same 64 graphs", same), ("random of 256", small), ("contiguous 64 at a random place", contig),
                   ("random of 4096, sorted", sorted_all), ("random of 4096", full)):
    print("%s cached, %-34s %.2f us per step" % (name, tag, timed(lists)), flush=True)
