"""Cached-topology steps replayed from a hipGraph, a different list of 64 graphs per step, with and without the L2 prefetch of
the NEXT step's graphs by spare workgroups of the launch (drgnn_step_hints.next_ids).
usage: python tools/r05/cached_prefetch_probe.py [net] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np                                            # noqa: E402
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd.resident import ResidentGraphSet        # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer             # noqa: E402
from deeprank_gnn_amd.ginet import GINet                      # noqa: E402
from deeprank_gnn_amd.sGAT import sGAT                        # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "GINet"
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = FusedTrainer({"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[name](32, 1, 1).to(dev), lr=1e-3, task="reg")
G = 2048
rs = ResidentGraphSet([synth.make_graph(i) for i in range(G)], dev)
cache = rs.topology_cache(need_weights=(name == "sGAT"))
rng = np.random.default_rng(0)
SHARE = os.environ.get("PROBE_SHARE", "0") == "1"
print("contexts shared by repetitions of a list:", SHARE)


def timed(lists, prefetch, n_steps=0):
    n = len(lists)
    if SHARE:      # one id buffer / one prepared context per DISTINCT list object (repetitions of a list share them)
        devmap = {id(l): rs.upload_ids(l) for l in lists}
        devs = [devmap[id(l)] for l in lists]
        cmap = {}
        cs = []
        for k in range(n):
            key = (id(lists[k]), id(lists[(k + 1) % n]) if prefetch else 0)
            if key not in cmap:
                cmap[key] = tr._cached_prepare(cache, lists[k], devs[k], True, devs[(k + 1) % n] if prefetch else None)
            cs.append(cmap[key])
    else:
        devs = [rs.upload_ids(l) for l in lists]
        cs = [tr._cached_prepare(cache, lists[k], devs[k], True, devs[(k + 1) % n] if prefetch else None) for k in range(n)]

    def chunk():
        for k in range(n_steps or STEPS):
            c = cs[k % n]
            c["stream"] = torch.cuda.current_stream().cuda_stream
            tr._cached_launch_step(c, True)
            tr._fused_launch_update(c, True, lr=0.0)
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
    for _ in range(40):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (40 * (n_steps or STEPS))


same = [np.arange(64)]
rand = [rng.choice(G, size=64, replace=False) for _ in range(STEPS)]
print("%s cached: same 64 graphs every step %.2f us | + prefetch (of itself) %.2f us" % (name, timed(same, False), timed(same, True)), flush=True)
print("%s cached: 64 random graphs of %d per step %.2f us | + prefetch of the next step's graphs %.2f us" % (
    name, G, timed(rand, False), timed(rand, True)), flush=True)
# the same SCATTERED list every step (64 random graphs instead of graphs 0 .. 63): warmth or contiguity?
print("%s cached: the same 64 RANDOM graphs every step %.2f us" % (name, timed([rand[0]], False)), flush=True)
contig = [np.arange(64) + 64 * k for k in range(STEPS)]
print("%s cached: 64 CONTIGUOUS graphs, a different run of 64 per step (cycle of %d) %.2f us" % (name, STEPS, timed(contig, False)), flush=True)
# every list stepped R times in a row: steps 2 .. R of a run find what the first one's workgroups (same slots, same XCDs) left in
# the L2 -- the upper bound of what any prefetch on the right XCD can deliver.  warmed = (R * mean - cold) / (R - 1)
cold = timed(rand, False)
for R in (2, 4, 8):
    rep = [l for l in rand for _ in range(R)]          # (the same 32 distinct lists: the cycle's working set stays 80 MB)
    m = timed(rep, False, n_steps=R * STEPS)
    print("%s cached: every random list %d times in a row %.2f us per step (cold %.2f -> steps 2.. of a run %.2f)" % (
        name, R, m, cold, (R * m - cold) / (R - 1)), flush=True)
