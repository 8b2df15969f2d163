"""The HIP library loads (no GPU needed) and exports every entry point include/drgnn.h declares;
the host-emulation build exports the same set.  No compute calls."""
import ctypes
import os
import re
import subprocess

import pytest

from helpers import ROOT

HEADER = os.path.join(ROOT, "include", "drgnn.h")
LIB = os.path.join(ROOT, "deeprank-gnn_amd", "csrc", "libdrgnn.so")


def declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(drgnn_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = declared()
    assert len(names) >= 25
    for must in ("drgnn_topology_build", "drgnn_net_forward", "drgnn_net_backward", "drgnn_net_backward_fused_head",
                 "drgnn_train_update", "drgnn_head_step", "drgnn_adam_step", "drgnn_conv_layer_forward",
                 "drgnn_segpool_forward", "drgnn_pooled_edges_export", "drgnn_cluster_offset"):
        assert must in names


def test_hip_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-C", os.path.dirname(LIB), "libdrgnn.so"])
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared() if not hasattr(lib, n)]
    assert not missing, missing
    lib.drgnn_abi_version.restype = ctypes.c_int
    assert lib.drgnn_abi_version() == 4
    # pure host-side helpers may be called without a GPU
    off_i = (ctypes.c_int64 * 64)()
    off_f = (ctypes.c_int64 * 8)()
    lib.drgnn_topology_layout.argtypes = [ctypes.c_int64] * 3 + [ctypes.POINTER(ctypes.c_int64)] * 2
    assert lib.drgnn_topology_layout(12800, 64000, 64, off_i, off_f) == 0
    assert off_i[0] == 0 and off_f[2] >= 2 * 64000
    lib.drgnn_net_lds_bytes.restype = ctypes.c_int64
    lib.drgnn_net_lds_bytes.argtypes = [ctypes.c_int32] * 6
    assert 0 < lib.drgnn_net_lds_bytes(0, 32, 200, 1005, 50, 0) <= 160 * 1024      # GINet SYN graph fits LDS
    assert 0 < lib.drgnn_net_lds_bytes(1, 32, 200, 1005, 50, 1) <= 160 * 1024      # sGAT backward too


def test_emulation_build_exports_the_same_surface():
    from emu_api import emu
    lib = emu().lib
    missing = [n for n in declared() if not hasattr(lib, n)]
    assert not missing, missing


def test_package_binding_covers_the_header():
    from deeprank_gnn_amd import _lib
    src = open(_lib.__file__).read()
    unbound = [n for n in declared() if "lib." + n not in src]
    assert not unbound, unbound
