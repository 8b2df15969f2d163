"""CPU oracle for graclus / normalized_cut / max_pool.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED.  No shipped net of the reference calls these; they only appear in the README's custom-net
recipe (reference README.md:98-126) and in example/model.py:12 as imports from torch_geometric.nn.  The
arithmetic lives in the un-vendored packages torch-geometric / torch-cluster (no version in reference
setup.py:42-47), and torch_cluster's graclus visits the nodes in a RANDOM permutation (and, without weights,
the neighbours in a random order), so there is no reference answer to match and the reference's tests hold no
vector for it.  Restated here is the published greedy matching with the permutation made an explicit input:

    graclus(rowptr, col, weight, perm):      labels <- -1
        for u in perm:  if labels[u] >= 0: continue
            v = the unmatched neighbour of u (v != u) with the largest weight -- first one in adjacency
                order when there are no weights or on ties -- or none
            labels[u] = labels[v] = min(u, v)   (labels[u] = u when there is none)
    adjacency order = input edge order (stable by source node)
    normalized_cut(edge_index, edge_attr, N) = edge_attr * (1/deg[row] + 1/deg[col]),  deg = in-degree by col
    max_pool(cluster, data) = consecutive_cluster -> per-cluster max of x, pool_edge (relabel, drop self loops,
        coalesce with summed attributes), batch[perm], mean pos      (same pieces as oracle/cpu_ref.py's
        community_pooling, which IS pinned on the fixture)
"""
import numpy as np


def graclus(edge_index, weight=None, num_nodes=None, perm=None):
    row, col = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    n = int(num_nodes if num_nodes is not None else (max(row.max(), col.max()) + 1 if row.size else 0))
    adj = [[] for _ in range(n)]
    for e in range(row.size):                       # input edge order, stable per source node
        adj[int(row[e])].append((int(col[e]), e))
    labels = np.full(n, -1, dtype=np.int64)
    order = range(n) if perm is None else [int(p) for p in perm]
    for u in order:
        if labels[u] >= 0:
            continue
        best, bw = -1, 0.0
        for v, e in adj[u]:
            if v == u or labels[v] >= 0:
                continue
            if weight is None:
                best = v
                break
            w = float(weight[e])
            if best < 0 or w > bw:
                best, bw = v, w
        m = min(u, best) if best >= 0 else u
        labels[u] = m
        if best >= 0:
            labels[best] = m
    return labels


def normalized_cut(edge_index, edge_attr, num_nodes):
    row, col = np.asarray(edge_index[0]), np.asarray(edge_index[1])
    deg = np.bincount(col, minlength=num_nodes).astype(np.float32)
    with np.errstate(divide="ignore"):
        inv = np.float32(1.0) / deg              # isolated targets never index an edge
    return np.asarray(edge_attr, dtype=np.float32).reshape(-1) * (inv[row] + inv[col])
