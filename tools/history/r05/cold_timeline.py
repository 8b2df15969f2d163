"""Where a step launch on data that is not cache-resident loses its time: the step kernel alone (no update launch), on prebuilt
topologies, replayed from a hipGraph over (a) the same mini-batch, (b) a cycle of N different mini-batches.  Run it under
libraries truncated with -DDRGNN_EXIT_AFTER=k (tools/r05/build_af_variant.sh) to get the cumulative timeline warm and cold.
usage: DRGNN_LIB=... python tools/r05/cold_timeline.py <tag> [net] [n_batches] [with_update]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd.topology import Topology                # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer             # noqa: E402
from deeprank_gnn_amd.ginet import GINet                      # noqa: E402
from deeprank_gnn_amd.sGAT import sGAT                        # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                  # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else "-"
name = sys.argv[2] if len(sys.argv) > 2 else "GINet"
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 32
with_update = len(sys.argv) > 4 and sys.argv[4] == "1"
dev = torch.device("cuda:0")
torch.manual_seed(0)
tr = FusedTrainer({"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[name](32, 1, 1).to(dev), lr=1e-3, task="reg")
need_w = name == "sGAT"
STEPS = 64


ARENA = int(os.environ.get("COLD_ARENA", "0"))
# COLD_FLUSH=<MB>: a kernel that reads that many MB runs between the steps (its own time is measured and subtracted): does a
# REPLAYED mini-batch stay fast when the L2s (8 x 4 MB) have been read over?  when the Infinity Cache (256 MB) has?
PRETOUCH = int(os.environ.get("COLD_PRETOUCH", "0"))
FLUSH_MB = int(os.environ.get("COLD_FLUSH", "0"))
flush_buf = torch.zeros(max(FLUSH_MB, 1) * (1 << 18), dtype=torch.float32, device="cuda:0")
flush_out = torch.zeros(1, dtype=torch.float32, device="cuda:0")
MIX = os.environ.get("COLD_MIX", "all")      # all | tiles (only the aggregation tiles differ per step) | ws (only the workspace)


def timed(batches):
    n = len(batches)
    topos = [Topology.from_batch(b, need_weights=need_w, build=(ARENA == 0)) for b in batches]
    if ARENA:
        # every workspace and every tiles buffer of the cycle out of ONE allocation, each on a 2 MB boundary (is the cold
        # penalty a matter of where the allocator puts the buffers -- address translation -- or of the bytes themselves?)
        two_mb = 2 << 20
        sizes = []
        for t in topos:
            sizes += [t.ws_i32.numel() * 4, t.tiles.numel() * 4]
        offs, o = [], 0
        for sz in sizes:
            offs.append(o)
            o += (sz + two_mb - 1) // two_mb * two_mb
        arena = torch.empty(o + two_mb, dtype=torch.uint8, device=dev)
        base = (-arena.data_ptr()) % two_mb
        for i, t in enumerate(topos):
            a0, a1 = base + offs[2 * i], base + offs[2 * i + 1]
            t.ws_i32 = arena[a0:a0 + sizes[2 * i]].view(torch.int32)
            t.tiles = arena[a1:a1 + sizes[2 * i + 1]].view(torch.float32)
            t.rebuild()
    cs = [tr._fused_prepare(b, t) for b, t in zip(batches, topos)]
    if MIX != "all" and n > 1:
        import ctypes
        from deeprank_gnn_amd import _lib
        mixed = []
        for k in range(n):
            # the launch of step k: everything of mini-batch 0 except ...
            c = dict(cs[0])
            if MIX == "tiles":          # ... the tiles (same shapes: every synthetic graph has 200 nodes)
                h, keep = _lib.step_hints(node_ptr=batches[0].__dict__["_host_node_ptr"], edge_ptr=batches[0].__dict__["_host_edge_ptr"],
                                          topo_flags=cs[0]["hints"][0].topo_flags, tiles=topos[k].tiles, plan=cs[0]["plan"])
                c["hints"] = (h, keep)
            elif MIX == "ws":           # ... the topology workspace (the tiles stay mini-batch 0's)
                c = dict(cs[k])
                h, keep = _lib.step_hints(node_ptr=batches[k].__dict__["_host_node_ptr"], edge_ptr=batches[k].__dict__["_host_edge_ptr"],
                                          topo_flags=cs[k]["hints"][0].topo_flags, tiles=topos[0].tiles, plan=cs[k]["plan"])
                c["hints"] = (h, keep)
                c["y"] = cs[0]["y"]
            mixed.append(c)
        cs = mixed

    def chunk():
        for k in range(STEPS):
            c = cs[k % n]
            c["stream"] = torch.cuda.current_stream().cuda_stream
            tr._fused_launch_step(c, None)
            if with_update:
                tr._fused_launch_update(c, True, lr=0.0)
            if FLUSH_MB:
                torch.sum(flush_buf, dim=0, keepdim=True, out=flush_out)
            if PRETOUCH:
                # a plain torch kernel (any CU, any XCD) reads the tiles of the mini-batch that is stepped NEXT
                torch.sum(topos[(k + 1) % n].tiles, dim=0, keepdim=True, out=flush_out)
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
    return e0.elapsed_time(e1) * 1e3 / (40 * STEPS)


def flush_us():
    if not FLUSH_MB:
        return 0.0
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        torch.sum(flush_buf, dim=0, keepdim=True, out=flush_out)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(STEPS):
            torch.sum(flush_buf, dim=0, keepdim=True, out=flush_out)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (10 * STEPS)


one = [synth.make_batch(0, 64).to(dev)]
many = [synth.make_batch(64 * i, 64).to(dev) for i in range(NB)]
ONLY = os.environ.get("COLD_ONLY", "")
w = timed(one) if ONLY != "cold" else 0.0
c = timed(many) if ONLY != "warm" else 0.0
fu = flush_us()
if PRETOUCH:
    # cost of the extra kernel: measured on the same-batch replay, whose step it does not change
    PRETOUCH = 0
    base = timed(one)
    print("pretouch kernel costs %.2f us per step (same-batch replay with it %.2f, without %.2f); subtracted below" % (w - base, w, base))
    c -= (w - base)
    w = base
if FLUSH_MB:
    print("flush kernel over %d MB alone: %.2f us; figures below have it subtracted" % (FLUSH_MB, fu))
    w, c = w - fu, c - fu
print("timeline %-8s %-8s mix=%-5s arena=%d update=%d  same mini-batch %6.2f us   cycle of %d %6.2f us   (+%.2f)" % (tag, name, MIX, ARENA, with_update, w, NB, c, c - w), flush=True)
