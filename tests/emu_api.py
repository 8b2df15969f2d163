"""Loads the HOST-EMULATION build of the kernels (tests only).

The emulation library is the product source (deeprank-gnn_amd/csrc/drgnn_capi.hip)
compiled by g++ with -DDRGNN_EMU: every kernel launch becomes a loop over workgroups and
work items on host memory (see csrc/drgnn_rt.h).  It lets the CPU suite check the kernels'
index logic and arithmetic against the oracle without a GPU.  The package never loads it.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "emu", "build", "libdrgnn_emu.so")
SRC_DIR = os.path.join(HERE, "..", "deeprank-gnn_amd", "csrc")
_api = None


def _stale():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    srcs = [os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR) if f.endswith((".h", ".hip"))]
    srcs.append(os.path.join(HERE, "..", "include", "drgnn.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def emu():
    global _api
    if _api is None:
        if _stale():
            subprocess.check_call([os.path.join(HERE, "emu", "build_emu.sh")])
        from deeprank_gnn_amd._lib import Api
        _api = Api(SO)
    return _api
