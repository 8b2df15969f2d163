"""Per-phase clock stamps of workgroup 0 of the fused step kernel (profiling build libdrgnn_prof.so:
make -C deeprank-gnn_amd/csrc libdrgnn_prof.so).  Prints source line of every barrier and the cycles since the last."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd import _lib                             # noqa: E402
from deeprank_gnn_amd.foutnet import FoutNet                  # noqa: E402
from deeprank_gnn_amd.ginet import GINet                      # noqa: E402
from deeprank_gnn_amd.sGAT import sGAT                        # noqa: E402
from deeprank_gnn_amd.topology import Topology                # noqa: E402
from deeprank_gnn_amd.trainer import FusedTrainer             # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "GINet"
api = _lib.Api(os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("PROF_LIB", "libdrgnn_prof.so")))
NB = int(os.environ.get("PROF_BATCH", "64"))      # PROF_LIB=libdrgnn_prof300.so PROF_BATCH=512: a workgroup of the SECOND round (warm I-cache)
api.lib.drgnn_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
buf = torch.zeros(4100, dtype=torch.int64, device=dev)
assert api.lib.drgnn_debug_set_phase_buffer(buf.data_ptr()) == 0
batch = synth.make_batch(0, NB).to(dev)
torch.manual_seed(0)
net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[kind](32, 1, 1).to(dev)
tr = FusedTrainer(net, lr=1e-3, task="reg", api=api)
if os.environ.get("PROF_NOCLASS"):      # the run-time layout (GINet one workgroup per graph: branch after branch)
    tr.plan_overrides = {"no_class": 1}
need_w = kind == "sGAT"
topo = Topology.from_batch(batch, api=api, need_weights=need_w)
for rep in range(3):
    buf.zero_()
    torch.cuda.synchronize()
    c = tr._fused_prepare(batch, topo)
    tr._fused_launch_step(c, None)
    torch.cuda.synchronize()
    if rep == 2:
        b = buf.cpu().tolist()
        k = b[0]
        print("== fused step %s: %d marks, total %d cycles" % (kind, k, (b[3 + 2 * (k - 1)] - b[3]) if k > 1 else 0))
        for i in range(1, k):
            print("   line %5d  +%7d cycles" % (b[2 + 2 * i], b[3 + 2 * i] - b[3 + 2 * (i - 1)]))
    tr._fused_launch_update(c, True)
