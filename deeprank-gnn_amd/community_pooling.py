"""Function-level pooling API with the reference's names (deeprank_gnn/community_pooling.py and
the torch_geometric / torch_scatter helpers its models import), on the device path.

    get_preloaded_cluster(cluster, batch)        community_pooling.py:25-30
    community_pooling(cluster, data)             community_pooling.py:161-251
    max_pool_x(cluster, x, batch)                [torch_geometric.nn] used at ginet.py:114
    graclus / normalized_cut / max_pool          [torch_geometric.nn] README.md:98-126 custom-net recipe
    scatter_mean / scatter_sum / scatter_max     [torch_scatter] used at ginet.py:133 etc.

These return dynamically-shaped tensors, so -- like the reference -- they synchronise with the
host; the shipped nets avoid them and run fused (see functional.net_body).  The OFFLINE half of
the reference module (community_detection with method 'mcl', run once per dataset and cached with
the graphs) is provided on the device through deeprank_gnn_amd.clustering; Louvain is not.
"""
import types

import torch

from . import _lib
from .data import Batch, Data
from .topology import Topology

__all__ = ["get_preloaded_cluster", "community_pooling", "max_pool", "max_pool_x", "graclus", "normalized_cut",
           "scatter_mean", "scatter_sum", "scatter_max", "community_detection", "community_detection_per_batch"]

_API = None      # tests point this at the host-emulation build


def _api():
    return _API or _lib.get()


def _num_graphs(batch):
    return int(batch.max()) + 1 if batch.numel() else 0


def _node_ptr(batch, n_graphs):
    counts = torch.bincount(batch, minlength=n_graphs)
    ptr = torch.zeros(n_graphs + 1, dtype=torch.int32, device=batch.device)
    ptr[1:] = counts.cumsum(0).to(torch.int32)
    return ptr


def get_preloaded_cluster(cluster, batch):
    """Make per-graph cluster ids globally unique with a running offset, IN PLACE (returns the
    same tensor), like the reference -- but as three small launches instead of a Python loop
    with two boolean masks and a host sync per graph."""
    api = _api()
    if api is _lib._API:
        _lib.require_device(cluster, batch)
    if cluster.dtype != torch.int64 or not cluster.is_contiguous():
        raise TypeError("cluster must be a contiguous int64 tensor (it is updated in place)")
    B = _num_graphs(batch)
    if B <= 1:
        return cluster
    scratch = torch.empty(B + 1, dtype=torch.int64, device=cluster.device)
    api.cluster_offset(cluster, _node_ptr(batch, B), B, scratch, _lib.current_stream(cluster))
    return cluster


class _SegMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, topo, n_clusters):
        api = topo.api
        x = x.contiguous()
        H = x.size(1)
        out = torch.empty((n_clusters, H), dtype=torch.float32, device=x.device)
        arg = torch.empty((n_clusters, H), dtype=torch.int64, device=x.device)
        api.segpool_forward(topo.ws_i32, topo.n_nodes, topo.n_edges, topo.n_graphs, x, H, 0, out, arg,
                            _lib.current_stream(x))
        ctx.api, ctx.n_nodes = api, x.size(0)
        ctx.save_for_backward(arg)
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _):
        (arg,) = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        gx = torch.zeros((ctx.n_nodes, grad_out.size(1)), dtype=torch.float32, device=grad_out.device)
        ctx.api.segmax_backward(grad_out, arg, arg.size(0), grad_out.size(1), ctx.n_nodes, gx,
                                _lib.current_stream(grad_out))
        return gx, None, None


def _segmean(x, topo, n_clusters):
    x = x.to(torch.float32).contiguous()
    out = torch.empty((n_clusters, x.size(1)), dtype=torch.float32, device=x.device)
    topo.api.segpool_forward(topo.ws_i32, topo.n_nodes, topo.n_edges, topo.n_graphs, x, x.size(1), 1, out, None,
                             _lib.current_stream(x))
    return out


def _shadow(data, cluster, edge_index, edge_attr):
    s = types.SimpleNamespace(edge_index=edge_index, edge_attr=edge_attr, cluster0=cluster, cluster1=None)
    batch = getattr(data, "batch", None)
    n = cluster.numel()
    if batch is None:
        batch = torch.zeros(n, dtype=torch.int64, device=cluster.device)
        s.__dict__["_num_graphs"] = 1 if n else 0
    s.batch = batch
    return s


def _pool_edges(topo, e1_total, with_attr):
    dev = topo.ws_i32.device
    ei = torch.empty((2, e1_total), dtype=torch.int64, device=dev)
    ea = torch.empty((e1_total, 1), dtype=torch.float32, device=dev) if with_attr else None
    topo.api.pooled_edges_export(topo.ws_i32, topo.ws_f32, topo.n_nodes, topo.n_edges, topo.n_graphs, e1_total,
                                 ei, ea, _lib.current_stream(topo.ws_i32))
    return ei, ea


def graclus(edge_index, weight=None, num_nodes=None, perm=None, batch=None):
    """``torch_geometric.nn.graclus(edge_index, weight, num_nodes)`` (README.md:98-126): greedy maximal matching,
    every node labelled ``min(u, v)`` of its pair (or itself).  torch_cluster visits the nodes in a random
    permutation; here the order is explicit: ``perm`` [N] (graph-major, LOCAL node ids of each graph) or identity,
    so results are reproducible.  ``batch`` [N] splits a block-diagonal batch into its graphs (one workgroup
    each); without it the input is one graph."""
    api = _api()
    edge_index = edge_index.to(torch.int64).contiguous()
    dev = edge_index.device
    if batch is not None:
        n = batch.numel()
    elif num_nodes is not None:
        n = int(num_nodes)
    else:
        n = int(edge_index.max()) + 1 if edge_index.numel() else 0
    s = types.SimpleNamespace(edge_index=edge_index, edge_attr=None, cluster0=None, cluster1=None)
    if batch is None:
        s.batch = torch.zeros(n, dtype=torch.int64, device=dev)
        s.__dict__["_num_graphs"] = 1 if n else 0
    else:
        s.batch = batch
    topo = Topology.from_batch(s, api=api, with_level1=False, graph_only=True, need_weights=False)
    topo.check()
    cluster = torch.empty((n,), dtype=torch.int64, device=dev)
    if weight is not None:
        weight = weight.to(torch.float32).reshape(-1).contiguous()
        if weight.numel() != edge_index.size(1):
            raise ValueError("graclus: one weight per edge expected")
    if perm is not None:
        perm = perm.to(torch.int64).contiguous()
        if perm.numel() != n:
            raise ValueError("graclus: perm must hold one (local) node id per node")
    if n:
        api.graclus(topo.ws_i32, n, topo.n_edges, topo.n_graphs, topo.max_nodes, topo.max_edges, weight, perm,
                    cluster, _lib.current_stream(cluster))
    return cluster


def normalized_cut(edge_index, edge_attr, num_nodes=None):
    """``torch_geometric.utils.normalized_cut``: edge_attr * (1/deg[row] + 1/deg[col]), deg = degree by target."""
    row, col = edge_index[0], edge_index[1]
    n = int(num_nodes) if num_nodes is not None else (int(edge_index.max()) + 1 if edge_index.numel() else 0)
    inv = 1.0 / torch.bincount(col, minlength=n).to(torch.float32)
    return edge_attr.reshape(-1).to(torch.float32) * (inv[row] + inv[col])


def max_pool(cluster, data):
    """``torch_geometric.nn.max_pool(cluster, data)`` (README.md:98-126): per-cluster feature maximum, pooled edges
    (relabelled, self loops dropped, duplicates merged with summed attributes), mean positions, pooled batch vector.
    ``cluster`` must not span graphs (graclus / get_preloaded_cluster labels do not)."""
    plain = types.SimpleNamespace(x=data.x, edge_index=data.edge_index, edge_attr=getattr(data, "edge_attr", None),
                                  pos=getattr(data, "pos", None), batch=getattr(data, "batch", None))
    return community_pooling(cluster, plain)


def community_pooling(cluster, data):
    """Pools features (max) and edges (relabel, drop self loops, merge duplicates with summed
    attributes) of all cluster members; positions are averaged.  ``cluster`` must not span
    graphs and must be graph-major (what get_preloaded_cluster returns)."""
    api = _api()
    cluster = cluster.to(torch.int64).contiguous()
    edge_attr = getattr(data, "edge_attr", None)
    topo = Topology.from_batch(_shadow(data, cluster, data.edge_index, edge_attr), api=api, with_level1=False)
    topo.check()
    c0, e1, _ = topo.totals()
    x, _ = _SegMax.apply(data.x, topo, c0)
    edge_index, pooled_attr = _pool_edges(topo, e1, edge_attr is not None)
    B = topo.n_graphs
    counts = topo.array("NC0")[:B].to(torch.int64)
    has_batch = getattr(data, "batch", None) is not None
    batch = torch.repeat_interleave(torch.arange(B, device=cluster.device), counts) if has_batch else None
    pos = _segmean(data.pos, topo, c0) if getattr(data, "pos", None) is not None else None
    if has_batch:
        out = Batch(batch=batch, x=x, edge_index=edge_index, edge_attr=pooled_attr, pos=pos)
        out.__dict__["_num_graphs"] = B
    else:
        out = Data(x=x, edge_index=edge_index, edge_attr=pooled_attr, pos=pos)
    iei = getattr(data, "internal_edge_index", None)
    if iei is not None:
        iea = getattr(data, "internal_edge_attr", None)
        t2 = Topology.from_batch(_shadow(data, cluster, iei, iea), api=api, with_level1=False)
        _, e1i, _ = t2.totals()
        out.internal_edge_index, out.internal_edge_attr = _pool_edges(t2, e1i, iea is not None)
    if getattr(data, "pos2D", None) is not None and not has_batch:
        out.pos2D = _segmean(data.pos2D, topo, c0)
    if getattr(data, "cluster0", None) is not None:
        out.cluster0 = data.cluster0
        out.cluster1 = getattr(data, "cluster1", None)
    return out


def max_pool_x(cluster, x, batch, size=None):
    """Per-cluster feature maximum + the graph id of every cluster.  ``size`` (torch_geometric): ``cluster`` already holds
    ids in ``[0, num_graphs * size)`` (``size`` slots per graph): the maxima go to exactly those rows (absent ids -> 0) and
    no batch vector is returned."""
    if size is not None:
        n_graphs = (int(batch.max()) + 1) if batch.numel() else 0
        out, _ = scatter_max(x, cluster, dim=0, dim_size=n_graphs * int(size))
        return out, None
    api = _api()
    cluster = cluster.to(torch.int64).contiguous()
    s = types.SimpleNamespace(edge_index=torch.zeros((2, 0), dtype=torch.int64, device=x.device), edge_attr=None,
                              cluster0=cluster, cluster1=None, batch=batch)
    topo = Topology.from_batch(s, api=api, with_level1=False)
    c0, _, _ = topo.totals()
    out, _ = _SegMax.apply(x, topo, c0)
    B = topo.n_graphs
    counts = topo.array("NC0")[:B].to(torch.int64)
    return out, torch.repeat_interleave(torch.arange(B, device=x.device), counts)


def _scatter_topology(index, n):
    return Topology.single_graph(None, None, n, api=_api(), cluster=index.to(torch.int64).contiguous())


def _dense_rows(pooled, index, dim_size, fill=0.0):
    """torch_scatter places segment k at row k (absent ids -> empty rows); the kernels pool the
    distinct ids in ascending order, so spread them back when ids have gaps."""
    ids = torch.unique(index)
    if dim_size is None:
        dim_size = int(ids[-1]) + 1 if ids.numel() else 0
    if ids.numel() == dim_size:
        return pooled, ids
    out = pooled.new_full((dim_size,) + tuple(pooled.shape[1:]), fill)
    out[ids] = pooled
    return out, ids


def _to_rows(src, index, dim):
    """(src as [n, h] with the scattered dimension first, restore) for a 1-D ``index`` over dimension ``dim`` of ``src``."""
    if index.dim() != 1:
        raise NotImplementedError("scatter_*: a 1-D index over the scattered dimension (what torch_scatter broadcasts) only")
    dim = dim % src.dim()
    moved = src.movedim(dim, 0)
    if moved.size(0) != index.numel():
        raise ValueError("index has %d entries, dimension %d of src has %d" % (index.numel(), dim, moved.size(0)))
    rest = tuple(moved.shape[1:])
    flat = moved.reshape(moved.size(0), -1)

    def restore(t):
        return t.reshape((t.size(0),) + rest).movedim(0, dim)
    return flat, restore


def _per_column(fn, src, index, dim, out, dim_size):
    """torch_scatter's general form: ``index`` has (or broadcasts to) the SHAPE of ``src`` -- every column scatters along ``dim``
    by its own index vector.  Neither the reference nor its layers use it (their indices are 1-D: ginet.py:71,133); it is served
    column by column through the 1-D path (h x the launches): completeness of the function API, not a hot path."""
    if out is not None:
        raise NotImplementedError("scatter_*: out= with an N-D index")
    d = dim % src.dim()
    idx = index.expand_as(src).movedim(d, 0)
    moved = src.movedim(d, 0)
    rest = tuple(moved.shape[1:])
    s2, i2 = moved.reshape(moved.size(0), -1), idx.reshape(moved.size(0), -1)
    if dim_size is None:
        dim_size = int(i2.max()) + 1 if i2.numel() else 0
    cols = [fn(s2[:, j:j + 1].contiguous(), i2[:, j].contiguous(), 0, None, dim_size) for j in range(s2.size(1))]

    def join(parts):
        t = torch.cat(parts, dim=1) if parts else s2.new_zeros((dim_size, 0))
        return t.reshape((t.size(0),) + rest).movedim(0, d)
    if cols and isinstance(cols[0], tuple):
        return join([c[0] for c in cols]), join([c[1] for c in cols])
    return join(cols)


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    """torch_scatter.scatter_max: (maxima, argmax along ``dim``); absent ids -> 0 / ``src.size(dim)``.  With ``out=``: the
    result is ``max(out, segment maxima)`` written into ``out`` (argmax = ``src.size(dim)`` where ``out`` keeps its value)."""
    if index.dim() > 1:
        return _per_column(scatter_max, src, index, dim, out, dim_size)
    if src.dim() != 2 or dim not in (0, -2) or out is not None:
        flat, restore = _to_rows(src, index, dim)
        n = flat.size(0)
        if out is not None:
            dim_size = out.size(dim % src.dim())
        if flat.size(1) == 0 or n == 0:
            size = dim_size if dim_size is not None else ((int(index.max()) + 1) if n else 0)
            return restore(flat.new_zeros((size, flat.size(1)))), restore(torch.full((size, flat.size(1)), n, dtype=torch.int64, device=src.device))
        dense, arg = scatter_max(flat, index, dim=0, dim_size=dim_size)
        if out is not None:
            present = torch.zeros(dense.size(0), dtype=torch.bool, device=src.device)
            present[torch.unique(index)] = True
            seg = torch.where(present.view(-1, 1), dense, torch.full_like(dense, float("-inf")))
            o2, _ = _to_rows(out, torch.arange(out.size(dim % src.dim()), device=src.device), dim)
            keep = ~(seg > o2)
            arg = torch.where(keep, torch.full_like(arg, n), arg)
            out.copy_(restore(torch.where(keep, o2, seg)))
            return out, restore(arg)
        return restore(dense), restore(arg)
    topo = _scatter_topology(index, src.size(0))
    c0, _, _ = topo.totals()
    pooled, arg = _SegMax.apply(src, topo, c0)
    dense, ids = _dense_rows(pooled, index, dim_size)
    if dense is not pooled:
        full = arg.new_full(dense.shape, src.size(0))
        full[ids] = arg
        arg = full
    return dense, arg


class _SegSumMean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, topo, n_clusters, mean):
        out = _segmean(x, topo, n_clusters)
        B = topo.n_graphs
        cl = topo.array("CL0")[:x.size(0)].to(torch.int64)
        counts = torch.bincount(cl, minlength=n_clusters).clamp(min=1).to(torch.float32)
        if not mean:
            out = out * counts.view(-1, 1)
        ctx.save_for_backward(cl, counts)
        ctx.mean = mean
        return out

    @staticmethod
    def backward(ctx, g):
        cl, counts = ctx.saved_tensors
        if ctx.mean:
            g = g / counts.view(-1, 1)
        return g[cl], None, None, None


def _scatter_reduce(src, index, dim, out, dim_size, mean):
    """torch_scatter.scatter_sum / scatter_mean along dim 0.  With ``out=``: the segment sums are
    ADDED into ``out`` (``out.scatter_add_``) and, for the mean, the WHOLE buffer -- old content included -- is divided
    by the clamped counts, in place; ``out`` itself is returned (what the reference's sGAT layer relies on, sGAT.py:82-87)."""
    if index.dim() > 1:
        return _per_column(lambda s, i, d, o, n: _scatter_reduce(s, i, d, o, n, mean), src, index, dim, out, dim_size)
    if src.dim() >= 1 and (dim % src.dim()) != 0:
        # any other dimension: bring it to the front, reduce, put it back (``out=``: the same view of the caller's buffer)
        d = dim % src.dim()
        res = _scatter_reduce(src.movedim(d, 0), index, 0, None if out is None else out.movedim(d, 0), dim_size, mean)
        return out if out is not None else res.movedim(0, d)
    if out is not None:
        dim_size = out.size(0)
    flat = src if src.dim() == 2 else src.reshape(src.size(0), -1)
    topo = _scatter_topology(index, flat.size(0))
    c0, _, _ = topo.totals()
    pooled = _SegSumMean.apply(flat, topo, c0, mean and out is None)
    dense, _ = _dense_rows(pooled, index, dim_size)
    dense = dense if src.dim() == 2 else dense.reshape((dense.size(0),) + tuple(src.shape[1:]))
    if out is None:
        return dense
    out.add_(dense)
    if mean:
        count = torch.bincount(index.reshape(-1).to(torch.int64), minlength=dim_size).clamp(min=1).to(out.dtype)
        out.div_(count.view(-1, *([1] * (out.dim() - 1))))
    return out


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    return _scatter_reduce(src, index, dim, out, dim_size, True)


def scatter_sum(src, index, dim=0, out=None, dim_size=None):
    return _scatter_reduce(src, index, dim, out, dim_size, False)


def community_detection(edge_index, num_nodes, edge_attr=None, method='mcl'):
    """Cluster labels of ONE graph (reference community_pooling.py:95-158).  'mcl' runs on the device
    (unweighted, like the reference's PreCluster call); 'louvain' (python-louvain, randomised, so
    without a reproducible answer to match) is not provided."""
    if method != 'mcl':
        raise ValueError('Clustering method %s not supported' % method)
    if edge_attr is not None:
        raise NotImplementedError("weighted MCL is never used by the reference's PreCluster")
    from .clustering import community_detection_mcl
    return community_detection_mcl(edge_index, num_nodes, api=_api())


def community_detection_per_batch(edge_index, batch, num_nodes, edge_attr=None, method='mcl'):
    """Per-graph MCL with the reference's running label offset (community_pooling.py:33-92:
    ``ncluster = max(cluster)`` after every graph, i.e. consecutive graphs SHARE one id)."""
    if method != 'mcl':
        raise ValueError('Clustering method %s not supported' % method)
    from .clustering import mcl_labels, _ptr_from_counts
    batch = torch.as_tensor(batch, device=edge_index.device)
    B = int(batch.max()) + 1
    node_ptr = _ptr_from_counts(torch.bincount(batch, minlength=B), edge_index.device)
    edge_ptr = _ptr_from_counts(torch.bincount(batch[edge_index[0]], minlength=B), edge_index.device)
    labels, _ = mcl_labels(edge_index, node_ptr, edge_ptr, api=_api())
    tops = torch.stack([labels[batch == g].max() for g in range(B)])
    offset = torch.cat([tops.new_zeros(1), tops.cumsum(0)[:-1]])
    return labels + offset[batch]
