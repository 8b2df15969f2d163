"""Per-barrier clock stamps of ONE builder workgroup (profiling build of the library: -DDRGNN_PHASE_TIMING; the stamping
workgroup is word 1 of the stamp buffer: PHASE_BLOCK=0 = the "pool" role of graph 0, 8 = its "structure" role).
usage: PROF_LIB=variants/libdrgnn_prof.so PHASE_BLOCK=8 python tools/r04/topo_phases.py [weights] [flags]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch                                                  # noqa: E402
import deeprank_gnn_amd.synthetic as synth                    # noqa: E402
from deeprank_gnn_amd import _lib                             # noqa: E402
from deeprank_gnn_amd.topology import Topology                # noqa: E402

need_w = len(sys.argv) > 1 and sys.argv[1] == "1"
flags = int(sys.argv[2]) if len(sys.argv) > 2 else None
api = _lib.Api(os.path.join(os.path.dirname(_lib.LIB_PATH), os.environ.get("PROF_LIB", "libdrgnn_prof.so")))
api.lib.drgnn_debug_set_phase_buffer.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
buf = torch.zeros(4100, dtype=torch.int64, device=dev)
assert api.lib.drgnn_debug_set_phase_buffer(buf.data_ptr()) == 0
batch = synth.make_batch(0, 64).to(dev)
for rep in range(3):
    buf.zero_()
    buf[1] = int(os.environ.get("PHASE_BLOCK", "0"))
    torch.cuda.synchronize()
    topo = Topology.from_batch(batch, api=api, need_weights=need_w, flags=flags)
    torch.cuda.synchronize()
    if rep == 2:
        b = buf.cpu().tolist()
        k = b[0]
        print("== k_topo %s weights=%d flags=%s: %d marks, total %d ticks" % (
            os.environ.get("PROF_LIB", "libdrgnn_prof.so"), need_w, flags, k, (b[3 + 2 * (k - 1)] - b[3]) if k > 1 else 0))
        for i in range(1, k):
            print("   line %5d  +%7d" % (b[2 + 2 * i], b[3 + 2 * i] - b[3 + 2 * (i - 1)]))
