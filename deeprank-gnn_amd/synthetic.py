"""Seeded synthetic residue-level interface graphs (the SYN workload of SURVEY.md §8(d)).

Shapes follow what HDF5DataSet.load_one_graph (reference DataSet.py:231-366) hands to
the models; ratios (E/N, N/C0, C0/C1) follow the reference's 1ATN fixture.

Per graph ``g``: ``rng = numpy.random.default_rng(1234 + g)``;
  * N nodes, first half "chain A", second half "chain B";
  * ``n_pairs`` distinct undirected A-B contacts, plus one extra contact for every
    node left uncovered (min degree >= 1, so FoutLayer's mean is finite);
  * ``edge_index = [pairs ; flipped pairs]^T`` (int64, unsorted, as DataSet.py:266-269);
  * ``edge_attr = tanh(-d/2 + 2) + 1`` with ``d ~ U(1.5, 8.5)`` (DataSet.py:96,281),
    duplicated for both directions;
  * ``x ~ N(0,1)`` [N,F], ``pos ~ N(0,1)`` [N,3], ``y ~ U(0,20)``;
  * ``cluster0``: inside each chain a random assignment to clusters of exactly 4 nodes
    (ids 0..C0-1, never spanning chains, like MCL on internal edges);
  * ``cluster1``: random assignment of the C0 clusters to ``n_c1`` groups;
  * ``internal_edge_index``: same-chain pairs, carried but unused by the convolutions.
"""
import numpy as np
import torch

from .data import Batch, Data

__all__ = ["make_graph", "make_batch"]


def _edge_feature_transform(d):
    return np.tanh(-d / 2.0 + 2.0) + 1.0


def make_graph(graph_id, n_nodes=200, n_pairs=500, n_feat=32, n_c1=16, n_internal=350,
               isolate_node=None):
    """Build one synthetic graph.  ``isolate_node`` (int) removes every contact of that
    node afterwards (used by the parity tests for the degree-0 edge case)."""
    rng = np.random.default_rng(1234 + graph_id)
    half = n_nodes // 2
    n_b = n_nodes - half

    # distinct cross-chain pairs, sampled without replacement from the half x n_b grid
    flat = rng.choice(half * n_b, size=min(n_pairs, half * n_b), replace=False)
    a = flat // n_b
    b = half + flat % n_b
    covered = np.zeros(n_nodes, dtype=bool)
    covered[a] = True
    covered[b] = True
    extra_a, extra_b = [], []
    have = set(zip(a.tolist(), b.tolist()))
    for node in np.nonzero(~covered)[0].tolist():
        while True:
            if node < half:
                cand = (node, half + int(rng.integers(n_b)))
            else:
                cand = (int(rng.integers(half)), node)
            if cand not in have:
                have.add(cand)
                extra_a.append(cand[0])
                extra_b.append(cand[1])
                break
    a = np.concatenate([a, np.asarray(extra_a, dtype=a.dtype)])
    b = np.concatenate([b, np.asarray(extra_b, dtype=b.dtype)])
    dist = rng.uniform(1.5, 8.5, size=a.shape[0])

    x = rng.standard_normal((n_nodes, n_feat)).astype(np.float32)
    pos = rng.standard_normal((n_nodes, 3)).astype(np.float32)
    y = np.float32(rng.uniform(0.0, 20.0))

    # cluster0: groups of 4 inside each chain
    def chain_clusters(lo, count, first_id):
        order = rng.permutation(count)
        ids = np.empty(count, dtype=np.int64)
        ids[order] = first_id + np.arange(count) // 4
        return ids, first_id + (count + 3) // 4
    c_a, nxt = chain_clusters(0, half, 0)
    c_b, n_c0 = chain_clusters(half, n_b, nxt)
    cluster0 = np.concatenate([c_a, c_b])
    order = rng.permutation(n_c0)
    cluster1 = np.empty(n_c0, dtype=np.int64)
    cluster1[order] = np.arange(n_c0) % max(1, min(n_c1, n_c0))

    # internal (same-chain) contacts
    ia, ib = [], []
    seen = set()
    while len(ia) < n_internal:
        chain = int(rng.integers(2))
        lo, cnt = (0, half) if chain == 0 else (half, n_b)
        if cnt < 2:
            break
        u, v = (lo + rng.choice(cnt, size=2, replace=False)).tolist()
        u, v = min(u, v), max(u, v)
        if (u, v) not in seen:
            seen.add((u, v))
            ia.append(u)
            ib.append(v)
    idist = rng.uniform(1.5, 8.5, size=len(ia))

    if isolate_node is not None:
        keep = (a != isolate_node) & (b != isolate_node)
        a, b, dist = a[keep], b[keep], dist[keep]

    def sym(u, v, d):
        ei = np.stack([np.concatenate([u, v]), np.concatenate([v, u])]).astype(np.int64)
        ea = _edge_feature_transform(np.concatenate([d, d])).astype(np.float32).reshape(-1, 1)
        return torch.from_numpy(ei), torch.from_numpy(ea)

    edge_index, edge_attr = sym(a, b, dist)
    iei, iea = sym(np.asarray(ia, dtype=np.int64), np.asarray(ib, dtype=np.int64), idist)
    g = Data(x=torch.from_numpy(x), edge_index=edge_index, edge_attr=edge_attr,
             y=torch.tensor([y]), pos=torch.from_numpy(pos))
    g.internal_edge_index = iei
    g.internal_edge_attr = iea
    g.mol = "syn_%d" % graph_id
    g.cluster0 = torch.from_numpy(cluster0)
    g.cluster1 = torch.from_numpy(cluster1)
    return g


def make_batch(first_id=0, n_graphs=64, **kw):
    return Batch.from_data_list([make_graph(first_id + i, **kw) for i in range(n_graphs)])


def save_store(path, n_graphs, first_id=0, **kw):
    """``n_graphs`` synthetic graphs written as a graph file in the GraphStore ``.npz`` layout (the reference's HDF5 tree,
    Graph.py:61-139: per molecule ``node_data/*``, ``edge_index`` (undirected pairs), ``edge_data/dist``,
    ``internal_edge_*``, ``score/irmsd``, ``clustering/mcl/depth_{0,1}``) -- what ``NeuralNet(database, Net, node_feature=['feat'],
    edge_feature=['dist'], target='irmsd')`` reads back into exactly these graphs (up to the float32 rounding of the distance)."""
    arrays, mols = {}, []
    for i in range(n_graphs):
        g = make_graph(first_id + i, **kw)
        m = "syn_%06d" % (first_id + i)
        mols.append(m)
        half = g.edge_index.shape[1] // 2
        ih = g.internal_edge_index.shape[1] // 2
        attr = g.edge_attr.reshape(-1)[:half].double().numpy()
        iattr = g.internal_edge_attr.reshape(-1)[:ih].double().numpy()
        inv = lambda a: 2.0 * (2.0 - np.arctanh(np.clip(a - 1.0, -0.999999, 0.999999)))       # noqa: E731  (tanh(-d/2 + 2) + 1)^-1
        arrays[m + "/node_data/feat"] = g.x.numpy()
        arrays[m + "/node_data/pos"] = g.pos.numpy()
        arrays[m + "/edge_index"] = g.edge_index[:, :half].t().contiguous().numpy()
        arrays[m + "/edge_data/dist"] = inv(attr)
        arrays[m + "/internal_edge_index"] = g.internal_edge_index[:, :ih].t().contiguous().numpy()
        arrays[m + "/internal_edge_data/dist"] = inv(iattr)
        arrays[m + "/score/irmsd"] = np.float64(g.y.item())
        arrays[m + "/clustering/mcl/depth_0"] = g.cluster0.numpy()
        arrays[m + "/clustering/mcl/depth_1"] = g.cluster1.numpy()
    arrays["__mols__"] = np.array(mols)
    np.savez(path, **arrays)
    return path
