"""Topology kernels (host-emulation build of the HIP source) vs the oracle.  CPU only."""
import numpy as np
import pytest
import torch

from helpers import fixture_batch, fixture_graphs, syn4_batch
from emu_api import emu
from topo_check import check_against_oracle, check_tiles
from deeprank_gnn_amd.data import Batch, Data
from deeprank_gnn_amd.topology import Topology
import deeprank_gnn_amd.synthetic as synth


def strip_layout(batch):
    """Forget the collate-time offsets -> exercises the device-side derivation."""
    for k in ("_node_ptr", "_edge_ptr", "_c1_ptr", "_max_nodes", "_max_edges", "_max_c0"):
        batch.__dict__.pop(k, None)
    return batch


def random_graph(rng, n, e, n_c0, n_c1, sym=True, self_loops=False, dup=False):
    row = rng.integers(0, n, size=e)
    col = rng.integers(0, n, size=e)
    if not self_loops:
        col = np.where(col == row, (col + 1) % max(n, 1), col)
    if dup and e > 2:
        row[1], col[1] = row[0], col[0]
    if sym:
        row, col = np.concatenate([row, col]), np.concatenate([col, row])
    ei = torch.from_numpy(np.stack([row, col]).astype(np.int64))
    ids0 = rng.permutation(np.arange(n) % max(n_c0, 1)) * 3 + 5          # gaps + offset: not consecutive
    c0 = int(np.unique(ids0).size)
    ids1 = rng.permutation(np.arange(c0) % max(n_c1, 1)) * 2
    g = Data(x=torch.from_numpy(rng.standard_normal((n, 5)).astype(np.float32)), edge_index=ei,
             edge_attr=torch.from_numpy(rng.uniform(0.1, 2.0, size=(ei.size(1), 1)).astype(np.float32)),
             y=torch.tensor([1.0]), pos=torch.zeros(n, 3))
    g.cluster0 = torch.from_numpy(ids0.astype(np.int64))
    g.cluster1 = torch.from_numpy(ids1.astype(np.int64))
    return g


@pytest.mark.parametrize("make", [lambda: fixture_batch(8), lambda: fixture_batch(10), syn4_batch,
                                  lambda: synth.make_batch(0, 3)])
@pytest.mark.parametrize("derive", [False, "host", "device"])
@pytest.mark.parametrize("weights", [True, False])      # False: pooled graph through the bitmap path
def test_topology_matches_oracle(make, derive, weights):
    # derive: the batch has forgotten its collate-time offsets (a foreign Batch object): "host" = from_batch derives the tables
    # with one host round trip (the default), "device" = the builder's own derivation (k_ptrs; host_tables=False)
    batch = make()
    if derive:
        strip_layout(batch)
    topo = Topology.from_batch(batch, api=emu(), need_weights=weights, host_tables=(derive != "device"))
    if derive == "host":
        assert topo.host_node_ptr is not None and topo.max_c0 > 0
    assert topo.status()[0] == 0
    check_against_oracle(topo, batch, weights=weights)


@pytest.mark.parametrize("make", [lambda: fixture_batch(8), lambda: fixture_batch(10), syn4_batch,
                                  lambda: synth.make_batch(0, 3)])
@pytest.mark.parametrize("weights", [True, False])
def test_lean_topology_matches_oracle_and_the_full_build(make, weights):
    """TOPO_LEAN (what the aggregation-first training kernels read, built by the short chains: concatenated scan, orders by
    counting, transposed bitmap): every array it promises equals the oracle's and the full build's."""
    from deeprank_gnn_amd import _lib
    batch = make()
    full = Topology.from_batch(batch, api=emu(), need_weights=weights)
    lean = Topology.from_batch(batch, api=emu(), need_weights=weights, flags=_lib.TOPO_HIER | _lib.TOPO_LEAN)
    assert lean.status()[0] == 0 and (lean.flags & _lib.TOPO_LEAN)
    check_against_oracle(lean, batch, weights=weights)
    names = ["ROWPTR0", "COL0", "EID0", "CL0", "NC0", "ROWPTR1", "COL1", "NE1", "COLPTR1", "ROWIDX1", "CL1", "NC1", "MPTR1",
             "MEM1", "HORD", "HMP0", "HSPLIT"] + (["TSLOT1"] if weights else [])
    nptr, eptr = full.array("NPTR").numpy(), full.array("EPTR").numpy()
    nc0, ne1 = full.array("NC0").numpy(), full.array("NE1").numpy()
    for name in names:
        a, b = full.array(name).numpy(), lean.array(name).numpy()
        for g in range(full.n_graphs):
            n0, N, e0, C, E1 = nptr[g], nptr[g + 1] - nptr[g], eptr[g], nc0[g], ne1[g]
            seg = {"ROWPTR0": (n0 + g, N + 1), "COL0": (e0, eptr[g + 1] - e0), "EID0": (e0, eptr[g + 1] - e0), "CL0": (n0, N),
                   "NC0": (g, 1), "ROWPTR1": (n0 + g, C + 1), "COL1": (e0, E1), "NE1": (g, 1), "COLPTR1": (n0 + g, C + 1),
                   "ROWIDX1": (e0, E1), "TSLOT1": (e0, E1), "CL1": (n0, C), "NC1": (g, 1), "MEM1": (n0, C),
                   "MPTR1": (n0 + g, lean.array("NC1").numpy()[g] + 1), "HORD": (n0, N), "HMP0": (n0 + g, C + 1),
                   "HSPLIT": (4 * g, 4)}[name]
            np.testing.assert_array_equal(a[seg[0]:seg[0] + seg[1]], b[seg[0]:seg[0] + seg[1]], err_msg="%s graph %d" % (name, g))
    if weights:
        np.testing.assert_array_equal(full.weights("W0").numpy()[:full.n_edges], lean.weights("W0").numpy()[:full.n_edges])
        for g in range(full.n_graphs):
            # (the lean chain sums the pooled weights in exact fixed point, the general chain in float32 in sorted order)
            np.testing.assert_allclose(full.weights("W1").numpy()[eptr[g]:eptr[g] + ne1[g]],
                                       lean.weights("W1").numpy()[eptr[g]:eptr[g] + ne1[g]], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("weights", [True, False])
@pytest.mark.parametrize("lean", [False, True])
def test_aggregation_tiles(weights, lean):
    """TOPO_TILES: the builder forms the level-0 neighbour sums (+ 1 / deg, mean edge weight) of every node and the inverse of
    the hierarchical order -- by the general chains and by the lean ones, with and without edge weights."""
    from deeprank_gnn_amd import _lib
    flags = _lib.TOPO_HIER | _lib.TOPO_TILES | (_lib.TOPO_LEAN if lean else 0)
    # (feature counts that are not a multiple of 4 -- 7, 26 -- : the tile rows are padded to 8 / 28 floats)
    for batch in (synth.make_batch(0, 3), synth.make_batch(5, 4, n_nodes=30, n_pairs=50, n_feat=8, n_c1=3, n_internal=8),
                  synth.make_batch(9, 3, n_nodes=30, n_pairs=50, n_feat=7, n_c1=3, n_internal=8),
                  synth.make_batch(2, 2, n_nodes=40, n_pairs=70, n_feat=26, n_c1=4, n_internal=20)):
        topo = Topology.from_batch(batch, api=emu(), need_weights=weights, flags=flags)
        assert topo.status()[0] == 0 and topo.tiles is not None
        check_against_oracle(topo, batch, weights=weights)
        check_tiles(topo, batch, weights)
    rng = np.random.default_rng(7)
    many = Batch.from_data_list([random_graph(rng, int(rng.integers(1, 12)), int(rng.integers(0, 20)), 3, 2) for _ in range(165)])
    many.x = torch.cat([many.x, many.x[:, :3]], dim=1).contiguous()      # F = 8
    topo = Topology.from_batch(many, api=emu(), need_weights=weights, flags=flags)
    check_against_oracle(topo, many, weights=weights)
    check_tiles(topo, many, weights)


@pytest.mark.parametrize("seed", range(6))
def test_lean_topology_random_ragged(seed):
    """Ragged random graphs (single nodes, no edges, gapped ids), both pooling paths, through the lean chains."""
    from deeprank_gnn_amd import _lib
    rng = np.random.default_rng(100 + seed)
    graphs = []
    for k in range(7):
        n = int(rng.integers(1, 40))
        e = int(rng.integers(0, 80))
        graphs.append(random_graph(rng, n, e, int(rng.integers(1, n + 1)), int(rng.integers(1, 6)), dup=(k % 2 == 0)))
    batch = Batch.from_data_list(graphs)
    for weights in (False, True):
        lean = Topology.from_batch(batch, api=emu(), need_weights=weights, flags=_lib.TOPO_HIER | _lib.TOPO_LEAN)
        assert lean.status()[0] == 0
        check_against_oracle(lean, batch, weights=weights)
    if seed == 0:
        # beyond 160 graphs the builder runs ONE workgroup per graph: the pool chain, then the structure chain
        many = Batch.from_data_list([random_graph(rng, int(rng.integers(1, 12)), int(rng.integers(0, 20)), 3, 2) for _ in range(165)])
        for weights in (False, True):
            lean = Topology.from_batch(many, api=emu(), need_weights=weights, flags=_lib.TOPO_HIER | _lib.TOPO_LEAN)
            assert lean.status()[0] == 0
            check_against_oracle(lean, many, weights=weights)


@pytest.mark.parametrize("seed", range(6))
def test_topology_random_ragged(seed):
    rng = np.random.default_rng(seed)
    graphs = []
    for k in range(7):
        n = int(rng.integers(1, 40))
        e = int(rng.integers(0, 4 * n))
        graphs.append(random_graph(rng, n, e, int(rng.integers(1, n + 1)), int(rng.integers(1, 5)),
                                   sym=bool(k % 2), self_loops=(k == 3), dup=(k == 4)))
    graphs.insert(2, random_graph(rng, 1, 0, 1, 1))              # single node, no edge
    batch = Batch.from_data_list(graphs)
    if seed % 2:
        strip_layout(batch)
    for weights in (True, False):
        topo = Topology.from_batch(batch, api=emu(), need_weights=weights)
        assert topo.status()[0] == 0
        check_against_oracle(topo, batch, weights=weights)


def test_topology_global_scratch_path():
    """max_nodes unknown / too large for LDS -> the same kernels run out of global scratch."""
    batch = syn4_batch()
    batch.__dict__["_max_nodes"] = 100000          # forces the LDS estimate over 160 KiB
    batch.__dict__["_max_edges"] = 100000
    topo = Topology.from_batch(batch, api=emu())
    assert topo.status()[0] == 0
    check_against_oracle(topo, batch)


def test_topology_flags_bad_input():
    batch = syn4_batch()
    batch.edge_index[1, 3] = batch.x.size(0) - 1            # endpoint in another graph
    topo = Topology.from_batch(batch, api=emu())
    assert topo.status()[0] & 1
    batch = syn4_batch()
    batch.cluster1 = batch.cluster1[:-1]                    # wrong length
    strip_layout(batch)
    topo = Topology.from_batch(batch, api=emu())
    assert topo.status()[0] & 8


def test_arbitrary_cluster_ids_take_the_general_path():
    """Ids far outside any small range (e.g. already offset globally, or hashed labels) are
    ranked by the O(n^2) rank-sort branch; result identical to torch.unique semantics."""
    batch = syn4_batch()
    batch.cluster0 = batch.cluster0 * 1000003 + 7 * 10 ** 11
    batch.cluster1 = batch.cluster1 * 99991 + 10 ** 12
    topo = Topology.from_batch(batch, api=emu())
    assert topo.status()[0] == 0
    check_against_oracle(topo, batch)


def test_finalize_scans():
    batch = fixture_batch(8)
    topo = Topology.from_batch(batch, api=emu()).finalize()
    nc0 = topo.array("NC0").numpy()[:8]
    np.testing.assert_array_equal(topo.array("CPTR0").numpy()[:9], np.concatenate([[0], np.cumsum(nc0)]))
    ne1 = topo.array("NE1").numpy()[:8]
    np.testing.assert_array_equal(topo.array("E1PTR").numpy()[:9], np.concatenate([[0], np.cumsum(ne1)]))
    assert topo.array("CPTR0").numpy()[8] == 244 and topo.array("E1PTR").numpy()[8] == 994
    assert topo.array("CPTR1").numpy()[8] == 83


def test_weighted_pooling_of_a_graph_with_more_than_65536_edges():
    """The weighted pooled-edge path ranks (target cluster, edge id) keys packed into one word while edge ids fit 16 bits; a
    graph beyond that (only the global-scratch builder takes one) keeps two words per candidate -- same pooled graph and
    summed weights as the oracle either way."""
    rng = np.random.default_rng(5)
    big = random_graph(rng, 400, 35000, 12, 3, sym=True)          # 70 000 directed edges, duplicates included
    small = random_graph(rng, 30, 60, 6, 2, sym=True)
    batch = Batch.from_data_list([small, big])
    assert int(batch.edge_index.size(1)) > 70000
    topo = Topology.from_batch(batch, api=emu(), need_weights=True)
    assert topo.status()[0] == 0
    check_against_oracle(topo, batch, weights=True)
