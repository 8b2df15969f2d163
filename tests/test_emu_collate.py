"""Resident graph set + device collate, kernels emulated on the CPU."""
import numpy as np
import pytest
import torch

from collate_check import check_collate, check_set_topology, ragged_graphs
from emu_api import emu
from helpers import fixture_graphs, syn4_graphs
from deeprank_gnn_amd.resident import ResidentGraphSet
from deeprank_gnn_amd import _lib


@pytest.mark.parametrize("n_feat", [1, 7, 12, 32])
def test_collate_ragged(n_feat):
    check_collate(ragged_graphs(3 + n_feat, n_feat), "cpu", api=emu())


def test_collate_fixture_and_synthetic():
    check_collate(fixture_graphs(), "cpu", api=emu())
    check_collate(syn4_graphs(), "cpu", api=emu())


def test_collate_without_optional_fields():
    graphs = ragged_graphs(1, 5)
    for g in graphs:
        g.edge_attr = None
        g.y = None
        g.__dict__.pop("cluster1", None)
    check_collate(graphs, "cpu", api=emu())


def test_collate_class_targets_and_errors():
    graphs = ragged_graphs(2, 4)
    rs = ResidentGraphSet(graphs, "cpu", api=emu())
    rs.set_targets(torch.arange(len(graphs)) % 2)
    b = rs.batch([3, 1, 4])
    assert b.y.dtype == torch.int64 and b.y.tolist() == [1, 1, 0]
    with pytest.raises(IndexError):
        rs.batch([0, len(graphs)])
    with pytest.raises(ValueError):
        rs.batch([])
    with pytest.raises(ValueError):
        rs.batch([0, 1], torch.zeros(3, dtype=torch.int32))


def test_resident_set_refuses_cpu_without_emulation():
    if not __import__("os").path.exists(_lib.LIB_PATH):
        pytest.skip("product library not built")
    with pytest.raises(_lib.DrgnnError):
        ResidentGraphSet(ragged_graphs(0, 3), "cpu")


@pytest.mark.parametrize("need_weights", [False, True])
def test_topology_from_the_resident_set(need_weights):
    check_set_topology(ragged_graphs(4, 6), "cpu", api=emu(), need_weights=need_weights, batch_size=4)
    check_set_topology(fixture_graphs(), "cpu", api=emu(), need_weights=need_weights, batch_size=3)


def test_resident_set_image_round_trip(tmp_path):
    from collate_check import assert_same_batch
    graphs = ragged_graphs(8, 5)
    rs = ResidentGraphSet(graphs, "cpu", api=emu())
    path = str(tmp_path / "set.npz")
    rs.save(path)
    back = ResidentGraphSet.load(path, "cpu", api=emu())
    assert len(back) == len(rs) and back.mols == rs.mols
    for ids in ([0, 3, 5], list(range(len(graphs)))[::-1]):
        assert_same_batch(back.batch(ids), rs.batch(ids))


def test_topology_from_the_resident_set_global_scratch_path():
    """A graph too large for the builder's LDS budget: the resident-set mode also runs out of global scratch."""
    import numpy as np
    from test_emu_topology import random_graph
    rng = np.random.default_rng(12)
    graphs = ragged_graphs(6, 4, count=3)
    big = random_graph(rng, 2500, 30000, 300, 40, sym=True)
    big.x = torch.from_numpy(rng.standard_normal((2500, 4)).astype(np.float32))
    big.edge_attr = torch.from_numpy(rng.uniform(0.1, 2.0, (big.edge_index.size(1), 1)).astype(np.float32))
    big.y = torch.tensor([1.0])
    big.mol = "big"
    graphs.insert(1, big)
    assert emu().topology_lds_bytes(2500, int(big.edge_index.size(1))) > 160 * 1024
    check_set_topology(graphs, "cpu", api=emu(), need_weights=True, batch_size=4)
