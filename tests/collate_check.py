"""drgnn_collate (resident graph set -> mini-batch on the device) against Batch.from_data_list, which is the
restated PyG collate pinned by tests/golden/collate.npz.  Shared by the emulated (CPU) and the MI355X test."""
import numpy as np
import torch

from deeprank_gnn_amd.data import Batch
from deeprank_gnn_amd.resident import ResidentGraphSet

KEYS = ("x", "edge_index", "edge_attr", "batch", "cluster0", "cluster1", "y")
HINTS = ("_node_ptr", "_edge_ptr", "_c1_ptr")


def ragged_graphs(seed, n_feat, count=9):
    from test_emu_topology import random_graph
    rng = np.random.default_rng(seed)
    graphs = []
    for k in range(count):
        n = 1 if k == 2 else int(rng.integers(2, 50))
        e = 0 if k == 2 else int(rng.integers(0, 4 * n))
        g = random_graph(rng, n, e, int(rng.integers(1, n + 1)), int(rng.integers(1, 5)), sym=bool(k % 2))
        g.x = torch.from_numpy(rng.standard_normal((n, n_feat)).astype(np.float32))
        g.edge_attr = torch.from_numpy(rng.uniform(0.1, 2.0, (g.edge_index.size(1), 1)).astype(np.float32))
        g.y = torch.tensor([float(rng.uniform(0, 20))])
        g.mol = "g%d" % k
        graphs.append(g)
    return graphs


def assert_same_batch(got, want):
    assert got.num_graphs == want.num_graphs
    for k in KEYS:
        a, b = got[k], want[k]
        assert (a is None) == (b is None), k
        if a is not None:
            assert a.dtype == b.dtype and tuple(a.shape) == tuple(b.shape), (k, a.dtype, b.dtype, a.shape, b.shape)
            assert torch.equal(a.cpu(), b.cpu()), k                     # copies and integer shifts: exact
    assert list(got["mol"]) == list(want["mol"])
    for k in HINTS:
        if want.__dict__.get(k) is not None:
            assert torch.equal(got.__dict__[k].cpu(), want.__dict__[k].cpu()), k
    for k in ("_max_nodes", "_max_edges", "_max_c0"):
        if k in want.__dict__:
            assert got.__dict__[k] == want.__dict__[k], k


def check_collate(graphs, device, api=None, selections=None):
    rs = ResidentGraphSet(graphs, device, api=api)
    G = len(graphs)
    if selections is None:
        rng = np.random.default_rng(5)
        selections = [list(range(G)), list(range(G - 1, -1, -1)), [G - 1], [0, 0, G // 2],
                      rng.permutation(G)[: max(1, G // 2)].tolist()]
    for ids in selections:
        want = Batch.from_data_list([graphs[i] for i in ids])
        assert_same_batch(rs.batch(ids), want)
    # ids already on the device (one upload per epoch, sliced per batch)
    flat = [i for ids in selections for i in ids]
    dev_ids = rs.upload_ids(flat)
    lo = 0
    for ids in selections:
        assert_same_batch(rs.batch(ids, dev_ids[lo:lo + len(ids)]), Batch.from_data_list([graphs[i] for i in ids]))
        lo += len(ids)
    return rs
