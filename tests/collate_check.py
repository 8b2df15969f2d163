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


def check_set_topology(graphs, device, api=None, need_weights=True, batch_size=4):
    """Resident-set mode of the topology builder (no collate) against the builder run on the collated mini-batch:
    identical workspaces, node features and targets; slot offset tables against numpy prefix sums."""
    from deeprank_gnn_amd.topology import Topology
    rs = ResidentGraphSet(graphs, device, api=api)
    order = np.random.default_rng(9).permutation(len(graphs)).tolist()
    ids_dev = rs.upload_ids(order)
    ptrs = rs.batch_offsets(ids_dev, batch_size)
    for k, lo in enumerate(range(0, len(order), batch_size)):
        ids = order[lo:lo + batch_size]
        B = len(ids)
        for t, counts in enumerate((rs.n_nodes, rs.n_edges, rs.n_c1)):
            want = np.concatenate([[0], np.cumsum(counts[ids])])
            assert ptrs[k, t, :B + 1].cpu().tolist() == want.tolist()
        ref_batch = rs.batch(ids)
        ref = Topology.from_batch(ref_batch, api=rs.api, need_weights=need_weights, build=False)
        mine = Topology(rs.api, ref.n_nodes, ref.n_edges, B, rs.device, need_weights)
        for t in (ref, mine):                       # unused tails of the workspace arrays compare equal
            t.ws_i32.zero_()
            if t.ws_f32 is not None:
                t.ws_f32.zero_()
        ref.rebuild()
        topo, x, y = rs.build_topology(ids, ids_dev[lo:lo + B], ptrs[k, :, :B + 1].contiguous(),
                                       need_weights=need_weights, topo=mine)
        assert (topo.max_nodes, topo.max_edges, topo.max_c0) == (ref.max_nodes, ref.max_edges, ref.max_c0)
        assert torch.equal(x.cpu(), ref_batch.x.cpu()) and torch.equal(y.cpu(), ref_batch.y.cpu())
        assert topo.status()[0] == 0 and ref.status()[0] == 0
        for name in ("NPTR", "EPTR", "ROWPTR0", "COL0", "COLPTR0", "ROWIDX0", "CL0", "MPTR0", "MEM0", "NC0", "NE1",
                     "ROWPTR1", "COL1", "COLPTR1", "ROWIDX1", "CL1", "MPTR1", "MEM1", "NC1"):
            assert torch.equal(topo.array(name).cpu(), ref.array(name).cpu()), name
        if need_weights:
            for name in ("W0", "W1"):
                assert torch.equal(topo.weights(name).cpu(), ref.weights(name).cpu()), name
