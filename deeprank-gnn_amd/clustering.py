"""Offline community detection on the device: the reference's ``PreCluster`` /
``community_detection(..., method='mcl')`` (DataSet.py:45-88, community_pooling.py:95-158) for
whole batches of graphs at once.

The reference runs Markov clustering (package ``markov-clustering``, default parameters) on the
unweighted INTERNAL-contact graph of every complex, pools that graph with the resulting clusters,
clusters the pooled graph again, and stores both label vectors with the dataset
(``clustering/mcl/depth_0`` and ``depth_1``); training only reads them back.  ``precluster`` does the
same for a ``Batch``: one workgroup per graph, dense fp64 (``drgnn_mcl``), pooling through the
topology builder.  Pinned on the fixture's stored labels (all graphs, both depths, exact).
Louvain (``python-louvain``, randomised) is not provided.
"""
import types

import torch

from . import _lib
from .topology import Topology

__all__ = ["mcl_labels", "precluster", "community_detection_mcl", "PreCluster"]


def _ptr_from_counts(counts, device):
    ptr = torch.zeros(counts.numel() + 1, dtype=torch.int32, device=device)
    ptr[1:] = counts.cumsum(0).to(torch.int32)
    return ptr


def mcl_labels(edge_index, node_ptr, edge_ptr, api=None):
    """Markov clustering of B graphs given as one block-diagonal edge list.
    edge_index int64 [2,E] (global node ids, grouped by graph), node_ptr / edge_ptr int32 [B+1].
    Returns (labels int64 [N], iterations int32 [B])."""
    api = api or _lib.get()
    if api is _lib._API:
        _lib.require_device(edge_index, node_ptr, edge_ptr)
    dev = node_ptr.device
    sizes = (node_ptr[1:] - node_ptr[:-1]).to(torch.int64)
    mat_ptr = torch.zeros(sizes.numel() + 1, dtype=torch.int64, device=dev)
    mat_ptr[1:] = (sizes * sizes).cumsum(0)
    total = int(mat_ptr[-1])                      # offline step: a host sync is fine
    n_nodes = int(node_ptr[-1])
    B = sizes.numel()
    mat = torch.empty(max(3 * total, 1), dtype=torch.float64, device=dev)
    iscr = torch.empty(max(4 * n_nodes, 1), dtype=torch.int32, device=dev)
    labels = torch.zeros(n_nodes, dtype=torch.int64, device=dev)
    info = torch.zeros(max(B, 1), dtype=torch.int32, device=dev)
    edge_index = edge_index.to(torch.int64).contiguous()
    api.mcl(edge_index, edge_index.size(1), node_ptr.contiguous(), edge_ptr.contiguous(), mat_ptr, B, mat, iscr,
            labels, info, _lib.current_stream(labels))
    return labels, info[:B]


def community_detection_mcl(edge_index, num_nodes, api=None):
    """One graph: ``community_detection(edge_index, num_nodes, method='mcl')``."""
    dev = edge_index.device
    node_ptr = torch.tensor([0, num_nodes], dtype=torch.int32, device=dev)
    edge_ptr = torch.tensor([0, edge_index.size(1)], dtype=torch.int32, device=dev)
    return mcl_labels(edge_index, node_ptr, edge_ptr, api=api)[0]


def precluster(batch, api=None):
    """(cluster0 [N], cluster1 [sum C0]) of a Batch that carries ``internal_edge_index``: what
    PreCluster would store as clustering/mcl/depth_0 and depth_1 for each of its graphs (per-graph
    local ids, concatenated in graph order)."""
    api = api or _lib.get()
    iei = batch.internal_edge_index.to(torch.int64).contiguous()
    bvec = batch.batch
    dev = bvec.device
    B = getattr(batch, "num_graphs", None) or (int(bvec.max()) + 1)
    node_ptr = _ptr_from_counts(torch.bincount(bvec, minlength=B), dev)
    edge_ptr = _ptr_from_counts(torch.bincount(bvec[iei[0]], minlength=B), dev)
    d0, _ = mcl_labels(iei, node_ptr, edge_ptr, api=api)
    # pool the internal-contact graph with depth_0 (community_pooling on internal edges, DataSet.py:81)
    shadow = types.SimpleNamespace(edge_index=iei, edge_attr=None, batch=bvec, cluster0=d0, cluster1=None)
    shadow.__dict__["_num_graphs"] = B
    topo = Topology.from_batch(shadow, api=api, with_level1=False, need_weights=False)
    topo.check()
    c0, e1, _ = topo.totals()
    pooled = torch.empty((2, e1), dtype=torch.int64, device=dev)
    api.pooled_edges_export(topo.ws_i32, None, topo.n_nodes, topo.n_edges, B, e1, pooled, None,
                            _lib.current_stream(pooled))
    node_ptr1 = topo.array("CPTR0")[:B + 1].clone()
    edge_ptr1 = topo.array("E1PTR")[:B + 1].clone()
    d1, _ = mcl_labels(pooled, node_ptr1, edge_ptr1, api=api)
    return d0, d1


def PreCluster(dataset, method='mcl', batch_size=64, device=None, api=None):
    """Pre-clusters the nodes of every graph of a ``GraphDataSet`` and attaches the labels to its
    store as ``clustering/<method>/depth_0`` and ``depth_1`` (reference DataSet.py:45-88, which
    writes them into the HDF5).  Call ``store.save_native(path)`` / ``save_npz(path)`` on ``dataset.stores`` to persist."""
    from .data import Batch
    if method.lower() != 'mcl':
        raise ValueError("only 'mcl' is available on the device (louvain is randomised)")
    api = api or _lib.get()
    if device is None:
        device = 'cuda' if api is _lib._API or torch.cuda.is_available() else 'cpu'
    where = [dataset.store_of(i) for i in range(len(dataset))]       # (store, mol) per entry: several files allowed
    for lo in range(0, len(where), batch_size):
        chunk = where[lo:lo + batch_size]
        graphs = [dataset.load_one_graph(m, st) for st, m in chunk]
        for g in graphs:
            g.cluster0 = None
            g.cluster1 = None
        batch = Batch.from_data_list(graphs).to(device)
        d0, d1 = precluster(batch, api=api)
        d0, d1 = d0.cpu().numpy(), d1.cpu().numpy()
        n_off = c_off = 0
        for (st, m), g in zip(chunk, graphs):
            n = g.num_nodes
            lab0 = d0[n_off:n_off + n]
            c = int(len(set(lab0.tolist())))
            st.set(m, "clustering/%s/depth_0" % method.lower(), lab0.copy())
            st.set(m, "clustering/%s/depth_1" % method.lower(), d1[c_off:c_off + c].copy())
            n_off += n
            c_off += c
    return dataset
