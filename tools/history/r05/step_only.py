"""N eager launches of the fused step kernel alone (no co-launched builder, no update) on the SYN64 batch: the workload of the
per-phase counter passes (tools/r05/lds_phase_counters.sh).  usage: python tools/r05/step_only.py [net] [launches]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "GINet"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
torch.manual_seed(0)
batch = synth.make_batch(0, 64).to(dev)
tr = FusedTrainer({"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[name](32, 1, 1).to(dev), lr=1e-3, task="reg", seed=1)
topo = Topology.from_batch(batch, need_weights=(name == "sGAT"))
c = tr._fused_prepare(batch, topo)
for _ in range(n):
    tr._fused_launch_step(c, None)
torch.cuda.synchronize()
print("done", name, n)
