"""Per-phase clock stamps of workgroup 0 (profiling build libdrgnn_prof.so)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd import _lib, functional
from deeprank_gnn_amd.topology import Topology
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT

kind_name = sys.argv[1] if len(sys.argv) > 1 else "GINet"
api = _lib.Api(os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("DRGNN_PROF_LIB", "libdrgnn_prof.so")))
api.lib.drgnn_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
buf = torch.zeros(4100, dtype=torch.int64, device=dev)
assert api.lib.drgnn_debug_set_phase_buffer(buf.data_ptr()) == 0
batch = synth.make_batch(0, 64).to(dev)
net = {"GINet": GINet, "sGAT": sGAT}[kind_name](32, 1, 1).to(dev)

def dump(title):
    torch.cuda.synchronize()
    b = buf.cpu().tolist()
    k = b[0]
    print("== %s: %d marks, total %d cycles" % (title, k, (b[3 + 2 * (k - 1)] - b[3]) if k > 1 else 0))
    for i in range(1, k):
        print("   line %5d  +%7d cycles" % (b[2 + 2 * i], b[3 + 2 * i] - b[3 + 2 * (i - 1)]))
    buf.zero_()
    torch.cuda.synchronize()

for rep in range(2):
    buf.zero_()
    topo = Topology.from_batch(batch, api=api, need_weights=(kind_name == "sGAT"))   # GINet / FoutNet: no edge weights
    if rep: dump("k_topo")
    buf.zero_(); torch.cuda.synchronize()
    x = batch.x
    out = net.body(batch, topo)
    if rep: dump("k_net fwd")
    buf.zero_(); torch.cuda.synchronize()
    out.sum().backward()
    if rep: dump("k_net bwd")

# ---- native training step (head / update kernels) -------------------------------------
from deeprank_gnn_amd.trainer import FusedTrainer
net2 = GINet(32, 1, 1).to(dev)
tr = FusedTrainer(net2, lr=1e-3, api=api)
for rep in range(2):
    topo = Topology.from_batch(batch, api=api, need_weights=False)
    stream = _lib.current_stream(batch.x)
    x, desc, xp, arg0, arg1, readout, scratch = tr._body_forward(batch, topo, stream)
    B = topo.n_graphs
    pred = torch.empty((B, 1), device=dev); gr = torch.empty_like(readout)
    hp = torch.empty((api.head_num_slabs(B), api.head_partial_elems(tr.R, tr.H, tr.O)), device=dev)
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    api.head_step(tr._head_desc(True), readout, batch.y.contiguous(), B, tr.step, pred, gr, hp, stream)
    if rep: dump("k_head")

# ---- the pipelined step's launches: forward, backward with fused head (+ next topology) ---------
for rep in range(2):
    topo = Topology.from_batch(batch, api=api, need_weights=False)
    nxt = Topology.from_batch(batch, api=api, need_weights=False, build=False)
    stream = _lib.current_stream(batch.x)
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    tr._body_forward(batch, topo, stream)
    if rep: dump("k_net fwd (trainer)")
    torch.cuda.synchronize(); buf.zero_(); torch.cuda.synchronize()
    tr.step.zero_()
    tr.train_step(batch, topo=topo, next_topo=nxt)
    if rep: dump("k_net_co_topo bwd + fused head (train_step)")
