"""ctypes binding of the C ABI in include/drgnn.h.

``get()`` loads ``csrc/libdrgnn.so`` (built by ``__graft_entry__.build()`` /
``make -C deeprank-gnn_amd/csrc``) and raises if it is missing: there is no CPU fallback.
``Api`` itself is device-agnostic pointer plumbing (it only reads ``data_ptr()``), which
is what lets the CPU test-suite drive the host-emulation build of the same kernels.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DRGNN_LIB: use another build of the same library (profiling / ablation builds under tools/)
LIB_PATH = os.environ.get("DRGNN_LIB") or os.path.join(_HERE, "csrc", "libdrgnn.so")

GINET, SGAT, FOUT = 0, 1, 2
MAX_BRANCH = 2

# enum drgnn_topo_i32 / drgnn_topo_f32 (include/drgnn.h)
TI = {name: i for i, name in enumerate([
    "NPTR", "EPTR", "ROWPTR0", "COL0", "EID0", "COLPTR0", "ROWIDX0", "TSLOT0", "CL0", "NC0",
    "MPTR0", "MEM0", "ROWPTR1", "COL1", "NE1", "COLPTR1", "ROWIDX1", "TSLOT1", "CL1", "NC1",
    "MPTR1", "MEM1", "CPTR0", "E1PTR", "CPTR1", "ERR", "GSTAT", "HORD", "HMP0", "HSPLIT", "IHORD"])}
TI_COUNT = len(TI)
TF = {"W0": 0, "W1": 1}
TF_COUNT = 2

FAULT_EXCHANGE = 1      # DRGNN_FAULT_EXCHANGE (step2[2])
STATUS_BITS = {1: "edge endpoint outside its graph's node range",
               2: "batch vector / edge list not grouped by graph",
               4: "cluster ids of one graph span too large a range",
               8: "len(cluster1) != number of depth-0 clusters"}

_c_i64 = ctypes.c_int64
_c_i32 = ctypes.c_int32
_vp = ctypes.c_void_p


class ConvParams(ctypes.Structure):
    _fields_ = [("w_nbr", _vp), ("nbr_sk", _c_i64), ("nbr_sh", _c_i64),
                ("w_self", _vp), ("self_sk", _c_i64), ("self_sh", _c_i64),
                ("bias", _vp)]


class ConvGrads(ctypes.Structure):
    _fields_ = [("w_nbr", _vp), ("w_self", _vp), ("bias", _vp)]


class NetDesc(ctypes.Structure):
    _fields_ = [("kind", _c_i32), ("n_branch", _c_i32), ("n_feat", _c_i32), ("reserved", _c_i32),
                ("conv1", ConvParams * MAX_BRANCH), ("conv2", ConvParams * MAX_BRANCH)]


class TopologyRequest(ctypes.Structure):
    _fields_ = [("edge_index", _vp), ("edge_attr", _vp), ("batch", _vp), ("cluster0", _vp), ("cluster1", _vp),
                ("node_ptr", _vp), ("edge_ptr", _vp), ("c1_ptr", _vp),
                ("n_nodes", _c_i64), ("n_edges", _c_i64), ("len_cluster1", _c_i64), ("n_graphs", _c_i64),
                ("max_nodes", _c_i32), ("max_edges", _c_i32),
                ("ws_i32", _vp), ("ws_f32", _vp), ("scratch_i32", _vp),
                # resident-set mode (include/drgnn.h); left NULL by the Python-level Topology
                ("set", _vp), ("ids", _vp), ("x_out", _vp), ("y_out", _vp),
                ("flags", _c_i32), ("reserved", _c_i32),
                ("x", _vp), ("tiles", _vp), ("n_feat", _c_i32), ("reserved2", _c_i32)]


TOPO_HIER = 1          # drgnn_topology_request.flags: also build the hierarchical node order (HORD / HMP0 / HSPLIT)
TOPO_LEAN = 2          # ... and ONLY what the aggregation-first training kernels read (no CSC0, no depth-0 member lists)
TOPO_TILES = 4         # ... and the level-0 neighbour aggregation of every node (drgnn_topology_request.x / tiles)


class GraphSet(ctypes.Structure):
    """drgnn_graph_set: the dataset resident in HBM, graph-major (include/drgnn.h)."""
    _fields_ = [("n_graphs", _c_i64), ("n_nodes", _c_i64), ("n_edges", _c_i64), ("len_cluster1", _c_i64),
                ("n_feat", _c_i32), ("y_bytes", _c_i32),
                ("node_ptr", _vp), ("edge_ptr", _vp), ("c1_ptr", _vp),
                ("x", _vp), ("edge_index", _vp), ("edge_attr", _vp),
                ("cluster0", _vp), ("cluster1", _vp), ("y", _vp)]


class EpochPlan(ctypes.Structure):
    """drgnn_epoch_plan (include/drgnn.h); pointer members are filled by FusedTrainer.train_epoch."""
    _fields_ = [("set", _vp), ("host_node_ptr", _vp), ("host_edge_ptr", _vp), ("host_c1_ptr", _vp),
                ("ids", _vp), ("host_ids", _vp), ("n_ids", _c_i64),
                ("batch_size", _c_i32), ("need_weights", _c_i32), ("inference", _c_i32), ("reserved", _c_i32),
                ("net", _vp), ("head", _vp), ("g_conv1", _vp), ("g_conv2", _vp),
                ("head_offset", _c_i64),
                ("flat_param", _vp), ("flat_grad", _vp), ("exp_avg", _vp), ("exp_avg_sq", _vp), ("n_param", _c_i64),
                ("step2", _vp),
                ("lr", ctypes.c_float), ("beta1", ctypes.c_float), ("beta2", ctypes.c_float), ("eps", ctypes.c_float),
                ("cache", _vp), ("exchange", _vp), ("exchange_user", _vp), ("step_overrides", _vp), ("last_loss", _vp)]


EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p)


class StepPlan(ctypes.Structure):
    """drgnn_step_plan (include/drgnn.h): the launch (in), the overrides (in), the layout (out)."""
    _fields_ = [("kind", _c_i32), ("n_feat", _c_i32), ("max_nodes", _c_i32), ("max_edges", _c_i32), ("max_c0", _c_i32),
                ("R", _c_i32), ("H", _c_i32), ("O", _c_i32),
                ("n_graphs", _c_i64), ("co_built_graphs", _c_i64),
                ("train", _c_i32), ("topo_flags", _c_i32),
                ("force_wgs", _c_i32), ("no_class", _c_i32), ("no_aggregate", _c_i32), ("no_split", _c_i32),
                ("no_paired", _c_i32),
                ("family", _c_i32), ("wgs_per_graph", _c_i32), ("slabs_per_graph", _c_i32), ("width", _c_i32),
                ("cls", _c_i32), ("lean_ok", _c_i32), ("builder_wgs_per_graph", _c_i32),
                ("lds_bytes", _c_i64), ("xchg_words", _c_i64),
                ("from_memory", _c_i32), ("reserved", _c_i32)]

    OVERRIDES = ("force_wgs", "no_class", "no_aggregate", "no_split", "no_paired")


STEP_FAMILY_NONE, STEP_FAMILY_PRODUCT, STEP_FAMILY_AGGREGATE = 0, 1, 2


class StepHints(ctypes.Structure):
    """drgnn_step_hints: host-side offset tables of a launch's graphs (pointers to HOST memory), what the workspace holds,
    the plan the caller sized its buffers from."""
    _fields_ = [("host_node_ptr", _vp), ("host_edge_ptr", _vp), ("set_node_ptr", _vp), ("set_edge_ptr", _vp),
                ("host_ids", _vp), ("topo_flags", _c_i32), ("reserved", _c_i32), ("tiles", _vp), ("plan", _vp),
                ("next_ids", _vp), ("n_next", _c_i64)]


class TopologyCacheDesc(ctypes.Structure):
    """drgnn_topology_cache: one topology workspace over a whole resident set + its node features / targets."""
    _fields_ = [("n_graphs", _c_i64), ("n_nodes", _c_i64), ("n_edges", _c_i64),
                ("ws_i32", _vp), ("ws_f32", _vp), ("x", _vp), ("y", _vp), ("y_bytes", _c_i32), ("flags", _c_i32),
                ("tiles", _vp)]


class HeadDesc(ctypes.Structure):
    _fields_ = [("R", _c_i32), ("H", _c_i32), ("O", _c_i32), ("task", _c_i32), ("train", _c_i32),
                ("p_drop", ctypes.c_float), ("seed", ctypes.c_uint32), ("transform_sigmoid", _c_i32),
                ("w1", _vp), ("b1", _vp), ("w2", _vp), ("b2", _vp), ("class_w", _vp), ("drop_mask", _vp)]


TASK_REG, TASK_CLASS, TASK_GRAD = 0, 1, 2      # (TASK_GRAD: `target` = d loss / d pred, the autograd boundary)
ZERO_RANGES = 8


def _ptr(t):
    return None if t is None else t.data_ptr()


class DrgnnError(RuntimeError):
    pass


def _check(rc, what):
    if rc != 0:
        kind = {-1: "bad argument", -2: "workspace/capacity", -3: "unsupported width"}.get(
            rc, "hipError_t %d" % rc if rc > 0 else "error %d" % rc)
        raise DrgnnError("%s failed: %s" % (what, kind))


class Api(object):
    """Typed wrapper around one loaded libdrgnn*.so."""

    def __init__(self, path):
        self.path = path
        lib = ctypes.CDLL(path)
        self.lib = lib
        lib.drgnn_abi_version.restype = ctypes.c_int
        lib.drgnn_topology_layout.argtypes = [_c_i64, _c_i64, _c_i64, ctypes.POINTER(_c_i64),
                                              ctypes.POINTER(_c_i64)]
        lib.drgnn_topology_scratch_elems.argtypes = [_c_i64] * 3
        lib.drgnn_topology_scratch_elems.restype = _c_i64
        lib.drgnn_topology_lds_bytes.argtypes = [_c_i32, _c_i32]
        lib.drgnn_topology_lds_bytes.restype = _c_i64
        lib.drgnn_topology_build.argtypes = [_vp] * 8 + [_c_i64] * 4 + [_c_i32] * 2 + [_vp] * 4
        lib.drgnn_topology_finalize.argtypes = [_vp, _c_i64, _c_i64, _c_i64, _vp]
        lib.drgnn_topology_status.argtypes = [_vp, _c_i64, _c_i64, _c_i64, ctypes.POINTER(_c_i32), _vp]
        lib.drgnn_net_lds_bytes.argtypes = [_c_i32] * 6
        lib.drgnn_net_lds_bytes.restype = _c_i64
        lib.drgnn_net_partial_elems.argtypes = [_c_i32, _c_i32]
        lib.drgnn_net_partial_elems.restype = _c_i64
        lib.drgnn_net_scratch_elems.argtypes = [_c_i32, _c_i32, _c_i64, _c_i64, _c_i64]
        lib.drgnn_net_scratch_elems.restype = _c_i64
        lib.drgnn_net_forward.argtypes = ([ctypes.POINTER(NetDesc)] + [_vp] * 3 + [_c_i64] * 3 +
                                          [_c_i32] * 3 + [_vp] * 7)
        lib.drgnn_net_backward_fused_head.argtypes = ([ctypes.POINTER(NetDesc), ctypes.POINTER(HeadDesc)] +
                                                      [_vp] * 6 + [_c_i64] * 3 + [_c_i32] * 3 + [_vp] * 8 +
                                                      [ctypes.POINTER(TopologyRequest), _vp])
        lib.drgnn_net_backward.argtypes = ([ctypes.POINTER(NetDesc)] + [_vp] * 4 + [_c_i64] * 3 +
                                           [_c_i32] * 3 + [_vp] * 3 + [_vp] * 5)
        lib.drgnn_train_update.argtypes = ([ctypes.POINTER(NetDesc), _vp, _c_i64] +
                                           [ctypes.POINTER(ConvGrads)] * 2 + [_vp, _c_i64] + [_c_i32] * 3 +
                                           [_c_i64] + [_vp] * 4 + [_c_i64] + [_vp] * 2 +
                                           [ctypes.c_float] * 4 + [_c_i32, _vp])
        lib.drgnn_net_step_lds_bytes.argtypes = [_c_i32] * 8
        lib.drgnn_net_step_lds_bytes.restype = _c_i64
        lib.drgnn_net_step_variant.argtypes = [_c_i32, _vp] + [_c_i32] * 6
        lib.drgnn_net_step_variant.restype = _c_i32
        lib.drgnn_net_step_plan.argtypes = [ctypes.POINTER(StepPlan)]
        lib.drgnn_net_step_plan.restype = _c_i32
        lib.drgnn_head_compact_elems.argtypes = [_c_i32] * 3
        lib.drgnn_head_compact_elems.restype = _c_i64
        lib.drgnn_net_train_step.argtypes = ([ctypes.POINTER(NetDesc), ctypes.POINTER(HeadDesc)] + [_vp] * 5 +
                                             [_c_i64] * 3 + [_c_i32] * 3 + [_vp] * 5 +
                                             [ctypes.POINTER(TopologyRequest), ctypes.POINTER(StepHints), _vp])
        lib.drgnn_net_train_step_cached.argtypes = ([ctypes.POINTER(NetDesc), ctypes.POINTER(HeadDesc),
                                                     ctypes.POINTER(TopologyCacheDesc), _vp, _c_i64] + [_c_i32] * 3 +
                                                    [_vp] * 6 + [ctypes.POINTER(StepHints), _vp])
        lib.drgnn_step_update.argtypes = ([ctypes.POINTER(NetDesc), _vp, _c_i64] +
                                          [ctypes.POINTER(ConvGrads)] * 2 + [_vp, _vp] + [_c_i32] * 3 +
                                          [_c_i64] + [_vp] * 4 + [_c_i64] + [_vp] * 2 +
                                          [ctypes.c_float] * 4 + [_c_i32, _c_i32, _vp])
        lib.drgnn_net_step_xchg_elems.argtypes = [_c_i32] * 4
        lib.drgnn_net_step_xchg_elems.restype = _c_i64
        lib.drgnn_net_reduce_grads.argtypes = ([ctypes.POINTER(NetDesc), _vp, _c_i64, _c_i64] +
                                               [ctypes.POINTER(ConvGrads)] * 2 + [_vp] * 2)
        lib.drgnn_conv_layer_slabs.argtypes = [_c_i64]
        lib.drgnn_conv_layer_slabs.restype = _c_i64
        lib.drgnn_conv_layer_partial_elems.argtypes = [_c_i32] * 3
        lib.drgnn_conv_layer_partial_elems.restype = _c_i64
        lib.drgnn_conv_layer_forward.argtypes = ([_c_i32, _vp, _c_i64, _c_i32, _c_i32, ctypes.POINTER(ConvParams)] +
                                                 [_vp] * 2 + [_c_i64] + [_vp] * 3)
        lib.drgnn_conv_layer_backward.argtypes = ([_c_i32, _vp, _c_i64, _c_i32, _c_i32, ctypes.POINTER(ConvParams)] +
                                                  [_vp] * 2 + [_c_i64] + [_vp] * 3 +
                                                  [ctypes.POINTER(ConvGrads)] + [_vp] * 2)
        lib.drgnn_segpool_forward.argtypes = [_vp, _c_i64, _c_i64, _c_i64, _vp, _c_i32, _c_i32, _vp, _vp, _vp]
        lib.drgnn_segmax_backward.argtypes = [_vp, _vp, _c_i64, _c_i32, _c_i64, _vp, _vp]
        lib.drgnn_pooled_edges_export.argtypes = [_vp, _vp, _c_i64, _c_i64, _c_i64, _c_i64, _vp, _vp, _vp]
        lib.drgnn_cluster_offset.argtypes = [_vp, _vp, _c_i64, _vp, _vp]
        lib.drgnn_graclus.argtypes = [_vp, _c_i64, _c_i64, _c_i64, _c_i32, _c_i32, _vp, _vp, _vp, _vp]
        lib.drgnn_mcl.argtypes = [_vp, _c_i64, _vp, _vp, _vp, _c_i64, _vp, _vp, _vp, _vp, _vp]
        lib.drgnn_p2p_bytes.argtypes = [_c_i64]
        lib.drgnn_p2p_bytes.restype = _c_i64
        lib.drgnn_p2p_alloc.argtypes = [_c_i64, ctypes.POINTER(_vp), _vp]
        lib.drgnn_p2p_open.argtypes = [_vp, ctypes.POINTER(_vp)]
        lib.drgnn_p2p_close.argtypes = [_vp]
        lib.drgnn_p2p_free.argtypes = [_vp]
        lib.drgnn_allreduce_oneshot.argtypes = [_vp, _c_i64, ctypes.POINTER(_vp), _c_i32, _c_i32, ctypes.c_float, _vp, _vp,
                                                _c_i32, _vp]
        lib.drgnn_train_epoch_scratch_bytes.argtypes = [ctypes.POINTER(EpochPlan)]
        lib.drgnn_train_epoch_scratch_bytes.restype = _c_i64
        lib.drgnn_train_epoch.argtypes = [ctypes.POINTER(EpochPlan), _vp, _c_i64, _vp, _vp, _vp]
        lib.drgnn_topology_build_request.argtypes = [ctypes.POINTER(TopologyRequest), _vp]
        lib.drgnn_topology_tiles.argtypes = [_vp, _vp, _c_i64, _c_i64, _c_i64, _vp, _c_i32, _c_i32, _vp, _vp]
        lib.drgnn_topology_tiles_elems.restype = _c_i64
        lib.drgnn_topology_tiles_elems.argtypes = [_c_i64, _c_i32]
        lib.drgnn_topology_tiles_ok.restype = _c_i32
        lib.drgnn_topology_tiles_ok.argtypes = [_c_i32, _c_i32, _c_i32]
        lib.drgnn_batch_offsets.argtypes = [ctypes.POINTER(GraphSet), _vp, _c_i64, _c_i32, _vp, _vp]
        lib.drgnn_collate.argtypes = [ctypes.POINTER(GraphSet), _vp] + [_c_i64] * 3 + [_vp] * 11
        lib.drgnn_head_partial_elems.argtypes = [_c_i32] * 3
        lib.drgnn_head_partial_elems.restype = _c_i64
        lib.drgnn_head_num_slabs.argtypes = [_c_i64]
        lib.drgnn_head_num_slabs.restype = _c_i64
        lib.drgnn_head_step.argtypes = [ctypes.POINTER(HeadDesc), _vp, _vp, _c_i64] + [_vp] * 5
        lib.drgnn_head_reduce.argtypes = [_vp, _c_i64, _c_i32, _c_i32, _c_i32] + [_vp] * 4
        lib.drgnn_adam_step.argtypes = [_vp] * 5 + [_c_i64] + [ctypes.c_float] * 5 + [_vp]
        lib.drgnn_step_gradients.argtypes = ([ctypes.POINTER(NetDesc), _vp, _c_i64] + [ctypes.POINTER(ConvGrads)] * 2 +
                                             [_vp, _vp] + [_c_i32] * 3 + [_vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(_c_i64), _c_i32,
                                                                       _vp, _c_i32, _vp])
        if lib.drgnn_abi_version() != 4:
            raise DrgnnError("ABI mismatch in %s" % path)

    # -- topology ---------------------------------------------------------------
    def topology_layout(self, n_nodes, n_edges, n_graphs):
        oi = (_c_i64 * (TI_COUNT + 1))()
        of = (_c_i64 * (TF_COUNT + 1))()
        _check(self.lib.drgnn_topology_layout(n_nodes, n_edges, n_graphs, oi, of), "drgnn_topology_layout")
        return list(oi), list(of)

    def topology_scratch_elems(self, n_nodes, n_edges, n_graphs):
        return int(self.lib.drgnn_topology_scratch_elems(n_nodes, n_edges, n_graphs))

    def topology_lds_bytes(self, max_nodes, max_edges):
        return int(self.lib.drgnn_topology_lds_bytes(max_nodes, max_edges))

    def topology_build(self, edge_index, edge_attr, batch, cluster0, cluster1, node_ptr, edge_ptr,
                       c1_ptr, n_nodes, n_edges, len_c1, n_graphs, max_nodes, max_edges, ws_i32,
                       ws_f32, scratch, stream):
        _check(self.lib.drgnn_topology_build(
            _ptr(edge_index), _ptr(edge_attr), _ptr(batch), _ptr(cluster0), _ptr(cluster1),
            _ptr(node_ptr), _ptr(edge_ptr), _ptr(c1_ptr), n_nodes, n_edges, len_c1, n_graphs,
            max_nodes, max_edges, _ptr(ws_i32), _ptr(ws_f32), _ptr(scratch), stream),
            "drgnn_topology_build")

    def topology_finalize(self, ws_i32, n_nodes, n_edges, n_graphs, stream):
        _check(self.lib.drgnn_topology_finalize(_ptr(ws_i32), n_nodes, n_edges, n_graphs, stream),
               "drgnn_topology_finalize")

    def topology_status(self, ws_i32, n_nodes, n_edges, n_graphs, stream):
        st = (_c_i32 * 4)()
        _check(self.lib.drgnn_topology_status(_ptr(ws_i32), n_nodes, n_edges, n_graphs, st, stream),
               "drgnn_topology_status")
        return list(st)

    # -- fused net --------------------------------------------------------------
    def net_lds_bytes(self, kind, n_feat, max_nodes, max_edges, max_c0, backward):
        return int(self.lib.drgnn_net_lds_bytes(kind, n_feat, max_nodes, max_edges, max_c0, int(backward)))

    def net_partial_elems(self, kind, n_feat):
        return int(self.lib.drgnn_net_partial_elems(kind, n_feat))

    def net_scratch_elems(self, kind, n_feat, n_nodes, n_edges, n_graphs):
        return int(self.lib.drgnn_net_scratch_elems(kind, n_feat, n_nodes, n_edges, n_graphs))

    def net_forward(self, desc, x, ws_i32, ws_f32, n_nodes, n_edges, n_graphs, max_nodes, max_edges,
                    max_c0, xp, arg0, arg1, readout, scratch, stream, step_inc=None):
        _check(self.lib.drgnn_net_forward(
            ctypes.byref(desc), _ptr(x), _ptr(ws_i32), _ptr(ws_f32), n_nodes, n_edges, n_graphs,
            max_nodes, max_edges, max_c0, _ptr(xp), _ptr(arg0), _ptr(arg1), _ptr(readout), _ptr(scratch),
            _ptr(step_inc), stream), "drgnn_net_forward")

    def net_backward_fused_head(self, desc, head, x, readout, target, step, ws_i32, ws_f32, n_nodes, n_edges,
                                n_graphs, max_nodes, max_edges, max_c0, xp, arg0, arg1, pred, head_partials,
                                grad_x, partials, scratch, stream, next_topology=None):
        _check(self.lib.drgnn_net_backward_fused_head(
            ctypes.byref(desc), ctypes.byref(head), _ptr(x), _ptr(readout), _ptr(target), _ptr(step),
            _ptr(ws_i32), _ptr(ws_f32), n_nodes, n_edges, n_graphs, max_nodes, max_edges, max_c0, _ptr(xp),
            _ptr(arg0), _ptr(arg1), _ptr(pred), _ptr(head_partials), _ptr(grad_x), _ptr(partials),
            _ptr(scratch), None if next_topology is None else ctypes.byref(next_topology), stream),
            "drgnn_net_backward_fused_head")

    def net_backward(self, desc, x, grad_readout, ws_i32, ws_f32, n_nodes, n_edges, n_graphs,
                     max_nodes, max_edges, max_c0, xp, arg0, arg1, grad_x, partials, scratch, stream,
                     step_inc=None):
        _check(self.lib.drgnn_net_backward(
            ctypes.byref(desc), _ptr(x), _ptr(grad_readout), _ptr(ws_i32), _ptr(ws_f32), n_nodes,
            n_edges, n_graphs, max_nodes, max_edges, max_c0, _ptr(xp), _ptr(arg0), _ptr(arg1),
            _ptr(grad_x), _ptr(partials), _ptr(scratch), _ptr(step_inc), stream), "drgnn_net_backward")

    def train_update(self, desc, conv_partials, n_graphs, g1, g2, head_partials, R, H, O, head_offset,
                     flat_p, flat_g, exp_avg, exp_avg_sq, step, loss, lr, beta1, beta2, eps, stream,
                     apply_adam=True):
        _check(self.lib.drgnn_train_update(
            ctypes.byref(desc), _ptr(conv_partials), n_graphs, g1, g2, _ptr(head_partials),
            head_partials.size(0), R, H, O,
            head_offset, _ptr(flat_p), _ptr(flat_g), _ptr(exp_avg), _ptr(exp_avg_sq), flat_p.numel(),
            _ptr(step), _ptr(loss), lr, beta1, beta2, eps, 1 if apply_adam else 0, stream),
            "drgnn_train_update")

    # -- fused training step ------------------------------------------------------
    def step_plan(self, kind, n_feat, max_nodes, max_edges, max_c0, R, H, O, n_graphs, co_built_graphs=0, train=True,
                  topo_flags=0, overrides=None):
        """The launch plan (StepPlan, ``out`` members filled) of a fused step of ``n_graphs`` graphs with these bounds on a
        workspace built with ``topo_flags``, the same launch building the topology of ``co_built_graphs`` graphs.
        ``overrides``: dict over StepPlan.OVERRIDES (tests, A/B runs).  ``family == 0``: outside the fused kernels."""
        p = StepPlan()
        p.kind, p.n_feat, p.max_nodes, p.max_edges, p.max_c0 = int(kind), int(n_feat), int(max_nodes), int(max_edges), int(max_c0)
        p.R, p.H, p.O = int(R), int(H), int(O)
        p.n_graphs, p.co_built_graphs = int(n_graphs), int(co_built_graphs)
        p.train, p.topo_flags = int(bool(train)), int(topo_flags)
        for k, v in (overrides or {}).items():
            if k not in StepPlan.OVERRIDES:
                raise KeyError("unknown step-plan override %r" % (k,))
            setattr(p, k, int(v))
        self.lib.drgnn_net_step_plan(ctypes.byref(p))
        return p

    def net_step_plan(self, kind, n_feat, max_nodes, max_edges, max_c0, R, H, O, n_graphs, co_built_graphs=0, train=True,
                      topo_flags=TOPO_HIER | TOPO_TILES, overrides=None):
        """(workgroups per graph, LDS bytes per workgroup) of ``step_plan`` -- by default for a training launch on a workspace
        with the hierarchical order and aggregation tiles (what the trainers build); (0, 0): outside the fused kernels."""
        p = self.step_plan(kind, n_feat, max_nodes, max_edges, max_c0, R, H, O, n_graphs, co_built_graphs, train, topo_flags,
                           overrides)
        return int(p.wgs_per_graph), int(p.lds_bytes)

    def net_step_lds_bytes(self, kind, n_feat, max_nodes, max_edges, max_c0, R, H, O):
        return int(self.lib.drgnn_net_step_lds_bytes(kind, n_feat, max_nodes, max_edges, max_c0, R, H, O))

    def step_is_specialised(self, kind, x, n_feat, max_nodes, max_edges, max_c0, H, O):
        """Padded feature width of the width-specialised fused step kernel these bounds launch (0: generic)."""
        return int(self.lib.drgnn_net_step_variant(kind, _ptr(x), n_feat, max_nodes, max_edges, max_c0, H, O))

    def head_compact_elems(self, R, H, O):
        return int(self.lib.drgnn_head_compact_elems(R, H, O))

    def net_train_step(self, desc, head, x, target, step2, ws_i32, ws_f32, n_nodes, n_edges, n_graphs,
                       max_nodes, max_edges, max_c0, pred, readout, head_partials, partials, xchg, stream,
                       next_topology=None, hints=None):
        _check(self.lib.drgnn_net_train_step(
            ctypes.byref(desc), ctypes.byref(head), _ptr(x), _ptr(target), _ptr(step2), _ptr(ws_i32),
            _ptr(ws_f32), n_nodes, n_edges, n_graphs, max_nodes, max_edges, max_c0, _ptr(pred), _ptr(readout),
            _ptr(head_partials), _ptr(partials), _ptr(xchg),
            None if next_topology is None else ctypes.byref(next_topology),
            None if hints is None else ctypes.byref(hints), stream), "drgnn_net_train_step")

    def net_train_step_cached(self, desc, head, cache, ids, n_graphs, max_nodes, max_edges, max_c0, step2, pred,
                              readout, head_partials, partials, xchg, stream, hints=None):
        _check(self.lib.drgnn_net_train_step_cached(
            ctypes.byref(desc), ctypes.byref(head), ctypes.byref(cache), _ptr(ids), n_graphs, max_nodes, max_edges,
            max_c0, _ptr(step2), _ptr(pred), _ptr(readout), _ptr(head_partials), _ptr(partials), _ptr(xchg),
            None if hints is None else ctypes.byref(hints), stream),
            "drgnn_net_train_step_cached")


    def step_update(self, desc, conv_partials, n_graphs, g1, g2, head_partials, readout, R, H, O, head_offset,
                    flat_p, flat_g, exp_avg, exp_avg_sq, step2, loss, lr, beta1, beta2, eps, stream,
                    apply_adam=True, slabs_per_graph=0):
        _check(self.lib.drgnn_step_update(
            ctypes.byref(desc), _ptr(conv_partials), n_graphs, g1, g2, _ptr(head_partials), _ptr(readout),
            R, H, O, head_offset, _ptr(flat_p), _ptr(flat_g), _ptr(exp_avg), _ptr(exp_avg_sq), flat_p.numel(),
            _ptr(step2), _ptr(loss), lr, beta1, beta2, eps, 1 if apply_adam else 0, int(slabs_per_graph), stream),
            "drgnn_step_update")

    def step_gradients(self, desc, conv_partials, n_graphs, g1, g2, head_partials, readout, R, H, O, head_grad,
                       graph_weight, zero_ptr, zero_len, n_zero, step2, slabs_per_graph, stream):
        """drgnn_step_gradients: the slabs of a fused step summed into the gradient tensors (each graph's slab times
        ``graph_weight[g]`` when given); ``zero_ptr`` / ``zero_len``: ctypes arrays of the ranges to clear."""
        _check(self.lib.drgnn_step_gradients(
            ctypes.byref(desc), _ptr(conv_partials), n_graphs, g1, g2, _ptr(head_partials), _ptr(readout), R, H, O,
            head_grad, _ptr(graph_weight), zero_ptr, zero_len, int(n_zero), _ptr(step2), int(slabs_per_graph), stream),
            "drgnn_step_gradients")

    def net_step_xchg_elems(self, kind, max_nodes, max_c0, H):
        return int(self.lib.drgnn_net_step_xchg_elems(kind, max_nodes, max_c0, H))

    def net_reduce_grads(self, desc, partials, n_nodes, n_graphs, g1, g2, grad_x, stream):
        _check(self.lib.drgnn_net_reduce_grads(ctypes.byref(desc), _ptr(partials), n_nodes, n_graphs,
                                               g1, g2, _ptr(grad_x), stream), "drgnn_net_reduce_grads")


    # -- stand-alone layers / pooling functions ----------------------------------
    def conv_layer_slabs(self, n_nodes):
        return int(self.lib.drgnn_conv_layer_slabs(n_nodes))

    def conv_layer_partial_elems(self, kind, F, H):
        return int(self.lib.drgnn_conv_layer_partial_elems(kind, F, H))

    def conv_layer_forward(self, kind, x, F, H, cp, ws_i32, ws_f32, n_edges, u, out, stream):
        _check(self.lib.drgnn_conv_layer_forward(kind, _ptr(x), x.size(0), F, H, ctypes.byref(cp), _ptr(ws_i32),
                                                 _ptr(ws_f32), n_edges, _ptr(u), _ptr(out), stream),
               "drgnn_conv_layer_forward")

    def conv_layer_backward(self, kind, x, F, H, cp, ws_i32, ws_f32, n_edges, grad_out, du, partials, cg,
                            grad_x, stream):
        _check(self.lib.drgnn_conv_layer_backward(kind, _ptr(x), x.size(0), F, H, ctypes.byref(cp), _ptr(ws_i32),
                                                  _ptr(ws_f32), n_edges, _ptr(grad_out), _ptr(du),
                                                  _ptr(partials), ctypes.byref(cg), _ptr(grad_x), stream),
               "drgnn_conv_layer_backward")

    def segpool_forward(self, ws_i32, n_nodes, n_edges, n_graphs, x, H, op, out, arg, stream):
        _check(self.lib.drgnn_segpool_forward(_ptr(ws_i32), n_nodes, n_edges, n_graphs, _ptr(x), H, op,
                                              _ptr(out), _ptr(arg), stream), "drgnn_segpool_forward")

    def segmax_backward(self, grad_out, arg, n_clusters, H, n_nodes, grad_x, stream):
        _check(self.lib.drgnn_segmax_backward(_ptr(grad_out), _ptr(arg), n_clusters, H, n_nodes, _ptr(grad_x),
                                              stream), "drgnn_segmax_backward")

    def pooled_edges_export(self, ws_i32, ws_f32, n_nodes, n_edges, n_graphs, e1_total, edge_index, edge_attr,
                            stream):
        _check(self.lib.drgnn_pooled_edges_export(_ptr(ws_i32), _ptr(ws_f32), n_nodes, n_edges, n_graphs,
                                                  e1_total, _ptr(edge_index), _ptr(edge_attr), stream),
               "drgnn_pooled_edges_export")

    def cluster_offset(self, cluster, node_ptr, n_graphs, scratch, stream):
        _check(self.lib.drgnn_cluster_offset(_ptr(cluster), _ptr(node_ptr), n_graphs, _ptr(scratch), stream),
               "drgnn_cluster_offset")

    def graclus(self, ws_i32, n_nodes, n_edges, n_graphs, max_nodes, max_edges, weight, perm, cluster, stream):
        _check(self.lib.drgnn_graclus(_ptr(ws_i32), n_nodes, n_edges, n_graphs, max_nodes, max_edges, _ptr(weight),
                                      _ptr(perm), _ptr(cluster), stream), "drgnn_graclus")

    def mcl(self, edge_index, n_edges, node_ptr, edge_ptr, mat_ptr, n_graphs, mat_scratch, int_scratch, labels,
            info, stream):
        _check(self.lib.drgnn_mcl(_ptr(edge_index), n_edges, _ptr(node_ptr), _ptr(edge_ptr), _ptr(mat_ptr), n_graphs,
                                  _ptr(mat_scratch), _ptr(int_scratch), _ptr(labels), _ptr(info), stream), "drgnn_mcl")

    def collate(self, gset, ids, n_graphs, n_nodes, n_edges, x, edge_index, edge_attr, batch, cluster0, cluster1,
                y, node_ptr, edge_ptr, c1_ptr, stream):
        _check(self.lib.drgnn_collate(ctypes.byref(gset), _ptr(ids), n_graphs, n_nodes, n_edges, _ptr(x),
                                      _ptr(edge_index), _ptr(edge_attr), _ptr(batch), _ptr(cluster0),
                                      _ptr(cluster1), _ptr(y), _ptr(node_ptr), _ptr(edge_ptr), _ptr(c1_ptr),
                                      stream), "drgnn_collate")

    def topology_tiles(self, ws_i32, ws_f32, n_nodes, n_edges, n_graphs, x, n_feat, use_weights, tiles, stream):
        """Aggregation tiles of an already built workspace (a second flavour for a shared cached topology)."""
        _check(self.lib.drgnn_topology_tiles(_ptr(ws_i32), _ptr(ws_f32), int(n_nodes), int(n_edges), int(n_graphs), _ptr(x),
                                             int(n_feat), int(bool(use_weights)), _ptr(tiles), stream), "drgnn_topology_tiles")

    def topology_tiles_elems(self, n_nodes, n_feat):
        return int(self.lib.drgnn_topology_tiles_elems(int(n_nodes), int(n_feat)))

    def topology_tiles_ok(self, max_nodes, max_edges, n_feat):
        """True when the builder can form the level-0 aggregation tiles (TOPO_TILES) for graphs of these bounds."""
        return bool(self.lib.drgnn_topology_tiles_ok(int(max_nodes), int(max_edges), int(n_feat)))

    def topology_build_request(self, request, stream):
        _check(self.lib.drgnn_topology_build_request(ctypes.byref(request), stream), "drgnn_topology_build_request")

    def batch_offsets(self, gset, ids, n_ids, batch_size, ptrs, stream):
        _check(self.lib.drgnn_batch_offsets(ctypes.byref(gset), _ptr(ids), n_ids, batch_size, _ptr(ptrs), stream),
               "drgnn_batch_offsets")

    # -- one-shot all-reduce over peer-mapped exchange buffers -------------------------
    def p2p_bytes(self, n_floats):
        return int(self.lib.drgnn_p2p_bytes(n_floats))

    def p2p_alloc(self, nbytes):
        """(device pointer, 64-byte IPC handle) of a zero-filled fine-grained exchange buffer."""
        ptr = _vp()
        handle = ctypes.create_string_buffer(64)
        _check(self.lib.drgnn_p2p_alloc(nbytes, ctypes.byref(ptr), handle), "drgnn_p2p_alloc")
        return ptr.value, handle.raw

    def p2p_open(self, handle):
        ptr = _vp()
        buf = ctypes.create_string_buffer(bytes(handle), 64)
        _check(self.lib.drgnn_p2p_open(buf, ctypes.byref(ptr)), "drgnn_p2p_open")
        return ptr.value

    def p2p_close(self, ptr):
        _check(self.lib.drgnn_p2p_close(ptr), "drgnn_p2p_close")

    def p2p_free(self, ptr):
        _check(self.lib.drgnn_p2p_free(ptr), "drgnn_p2p_free")

    def allreduce_oneshot(self, grad, n, peers, world, rank, weight, seq, status, stream, part=0):
        arr = (_vp * world)(*peers)
        _check(self.lib.drgnn_allreduce_oneshot(_ptr(grad), n, arr, world, rank, float(weight), _ptr(seq), _ptr(status),
                                                part, stream), "drgnn_allreduce_oneshot")

    def train_epoch_scratch_bytes(self, plan):
        """Bytes of device scratch for the plan; None when a graph does not fit the fused kernels."""
        n = int(self.lib.drgnn_train_epoch_scratch_bytes(ctypes.byref(plan)))
        if n == -2:
            return None
        if n < 0:
            _check(n, "drgnn_train_epoch_scratch_bytes")
        return n

    def train_epoch(self, plan, scratch, pred, losses, stream):
        _check(self.lib.drgnn_train_epoch(ctypes.byref(plan), _ptr(scratch), scratch.numel() * scratch.element_size(),
                                          _ptr(pred), _ptr(losses), stream), "drgnn_train_epoch")

    # -- head / loss / optimiser ------------------------------------------------
    def head_partial_elems(self, R, H, O):
        return int(self.lib.drgnn_head_partial_elems(R, H, O))

    def head_num_slabs(self, n_graphs):
        return int(self.lib.drgnn_head_num_slabs(n_graphs))

    def head_step(self, desc, readout, target, n_graphs, step, pred, grad_readout, partials, stream):
        _check(self.lib.drgnn_head_step(ctypes.byref(desc), _ptr(readout), _ptr(target), n_graphs,
                                        _ptr(step), _ptr(pred), _ptr(grad_readout), _ptr(partials),
                                        stream), "drgnn_head_step")

    def head_reduce(self, partials, n_graphs, R, H, O, grad_block_ptr, loss, step, stream):
        _check(self.lib.drgnn_head_reduce(_ptr(partials), n_graphs, R, H, O, grad_block_ptr, _ptr(loss),
                                          _ptr(step), stream), "drgnn_head_reduce")

    def adam_step(self, param, grad, exp_avg, exp_avg_sq, step, lr, beta1, beta2, eps, weight_decay,
                  stream):
        _check(self.lib.drgnn_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq),
                                        _ptr(step), param.numel(), lr, beta1, beta2, eps, weight_decay,
                                        stream), "drgnn_adam_step")


_API = None


def get():
    """The product library.  Raises if it has not been built: no CPU fallback."""
    global _API
    if _API is None:
        if not os.path.exists(LIB_PATH):
            raise DrgnnError(
                "HIP library %s not found. Build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` or `make -C deeprank-gnn_amd/csrc` (hipcc, gfx950)." % LIB_PATH)
        _API = Api(LIB_PATH)
    return _API


def require_device(*tensors):
    """The kernels run on the GPU only; refuse anything else loudly."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DrgnnError("deeprank_gnn_amd kernels need CUDA/HIP tensors (MI355X); got a %s "
                             "tensor. There is no CPU path." % t.device.type)


def current_stream(ref):
    """Raw hipStream_t of torch's current stream on the device of ``ref``."""
    if ref.is_cuda:
        return torch.cuda.current_stream(ref.device).cuda_stream
    return None


def step_hints(node_ptr=None, edge_ptr=None, set_node_ptr=None, set_edge_ptr=None, ids=None, topo_flags=0, tiles=None, plan=None,
               next_ids=None):
    """(StepHints, keep-alive tuple) from numpy arrays: int32 per-mini-batch tables, or int64 set tables + int32 ids;
    ``plan``: the StepPlan the caller sized its buffers from."""
    import numpy as np
    h = StepHints()
    keep = []

    def pin(a, dtype):
        if a is None:
            return None
        a = np.ascontiguousarray(np.asarray(a, dtype=dtype))
        keep.append(a)
        return a.ctypes.data
    h.host_node_ptr, h.host_edge_ptr = pin(node_ptr, np.int32), pin(edge_ptr, np.int32)
    h.set_node_ptr, h.set_edge_ptr, h.host_ids = pin(set_node_ptr, np.int64), pin(set_edge_ptr, np.int64), pin(ids, np.int32)
    h.topo_flags = int(topo_flags)
    h.tiles = _ptr(tiles)
    if tiles is not None:
        keep.append(tiles)
    if plan is not None:
        h.plan = ctypes.addressof(plan)
        keep.append(plan)
    if next_ids is not None and next_ids.numel() > 0:      # (device int32 tensor: the next mini-batch's graph numbers)
        h.next_ids, h.n_next = next_ids.data_ptr(), int(next_ids.numel())
        keep.append(next_ids)
    return h, tuple(keep)
