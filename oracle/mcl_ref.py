"""CPU oracle for the OFFLINE clustering step.  TEST INFRASTRUCTURE ONLY.

Restates ``community_detection(edge_index, num_nodes, method='mcl')`` (reference
community_pooling.py:95-158) and ``PreCluster`` (DataSet.py:45-88).  The arithmetic lives in
the un-vendored package ``markov-clustering`` (listed without version in reference setup.py:45;
``run_mcl`` defaults: expansion 2, inflation 2, loop_value 1, iterations 100, pruning_threshold
1e-3, pruning_frequency 1, convergence_check_frequency 1) whose published algorithm is restated
here on dense float64 matrices:

    M <- A with unit self loops, columns normalised (l1)
    repeat <= 100 times:
        last = M ; M <- M @ M (expansion) ; M <- colnorm(M ** 2) (inflation)
        prune: entries < 1e-3 -> 0 but every column keeps its maximum entry
        stop when max(|M - last| - 1e-5 |last|) <= 1e-8
    clusters = { nonzero columns of row i : M[i,i] != 0 }, sorted as tuples;
    label[members of cluster k] = k in that order (later clusters overwrite on overlap)

PINNED: reproduces the reference's own stored results ``clustering/mcl/depth_0`` AND ``depth_1`` of
all 10 graphs of tests/hdf5/1ATN_residue.hdf5 exactly (tests/test_mcl.py) -- those datasets were
written by the reference's real PreCluster (real markov_clustering + real torch_geometric).
"""
import numpy as np


def _colnorm(m):
    s = np.abs(m).sum(axis=0)
    s[s == 0.0] = 1.0
    return m / s


def run_mcl(adj, expansion=2, inflation=2, loop_value=1.0, iterations=100, pruning_threshold=1e-3):
    m = np.array(adj, dtype=np.float64)
    n = m.shape[0]
    if loop_value > 0:
        m[np.arange(n), np.arange(n)] = loop_value
    m = _colnorm(m)
    n_iter = 0
    for n_iter in range(1, iterations + 1):
        last = m.copy()
        m = np.linalg.matrix_power(m, expansion)
        m = _colnorm(np.power(m, inflation))
        if pruning_threshold > 0:
            pruned = np.where(m >= pruning_threshold, m, 0.0)
            top = m.argmax(axis=0)
            cols = np.arange(n)
            pruned[top, cols] = m[top, cols]
            m = pruned
        if (np.abs(m - last) - 1e-5 * np.abs(last)).max() <= 1e-8:
            break
    return m, n_iter


def get_clusters(m):
    attractors = np.nonzero(np.diagonal(m))[0]
    clusters = set()
    for a in attractors:
        clusters.add(tuple(np.nonzero(m[a])[0].tolist()))
    return sorted(clusters)


def community_detection_mcl(edge_index, num_nodes):
    """edge_index: int array [2, E] (undirected graph, any duplication).  Returns labels [num_nodes]."""
    adj = np.zeros((num_nodes, num_nodes), dtype=np.float64)
    ei = np.asarray(edge_index)
    adj[ei[0], ei[1]] = 1.0
    adj[ei[1], ei[0]] = 1.0            # networkx Graph: undirected
    m, _ = run_mcl(adj)
    labels = np.zeros(num_nodes, dtype=np.int64)
    for k, members in enumerate(get_clusters(m)):
        labels[list(members)] = k
    return labels


def pool_edge_index(cluster, edge_index):
    """index part of pool_edge after consecutive_cluster (community_pooling.py:195-210)."""
    uniq, cons = np.unique(cluster, return_inverse=True)
    row, col = cons[edge_index[0]], cons[edge_index[1]]
    keep = row != col
    key = np.unique(row[keep] * cluster.size + col[keep])
    return np.stack([key // cluster.size, key % cluster.size]), uniq.size


def precluster(internal_edge_index, num_nodes):
    """depth_0, depth_1 exactly as PreCluster computes them (DataSet.py:77-86)."""
    d0 = community_detection_mcl(internal_edge_index, num_nodes)
    pooled, n_pooled = pool_edge_index(d0, np.asarray(internal_edge_index))
    d1 = community_detection_mcl(pooled, n_pooled)
    return d0, d1
