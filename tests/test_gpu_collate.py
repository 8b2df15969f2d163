"""Resident graph set + drgnn_collate on the MI355X against the restated PyG collate (Batch.from_data_list)."""
import numpy as np
import pytest
import torch

from collate_check import check_collate, check_set_topology, ragged_graphs
from helpers import fixture_graphs, syn4_graphs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_feat", [1, 7, 32])
def test_collate_ragged(n_feat):
    check_collate(ragged_graphs(3 + n_feat, n_feat), "cuda")


def test_collate_fixture_and_synthetic():
    check_collate(fixture_graphs(), "cuda")
    check_collate(syn4_graphs(), "cuda")


def test_collate_full_size_batches_train_identically():
    """SYN graphs at BASELINE size: batches of 64 out of a resident set of 160 graphs are bit-identical to the
    host collate, and one epoch of fused training steps over them gives the same losses."""
    import copy
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.trainer import FusedTrainer
    from collate_check import assert_same_batch
    graphs = [synth.make_graph(i) for i in range(160)]
    rs = ResidentGraphSet(graphs, "cuda")
    order = np.random.default_rng(0).permutation(160).tolist()
    ids_dev = rs.upload_ids(order)
    torch.manual_seed(0)
    net = GINet(32, 1, 1).cuda()
    net.dropout = 0.0
    tr_a = FusedTrainer(net, lr=1e-3, task="reg")
    tr_b = FusedTrainer(copy.deepcopy(net), lr=1e-3, task="reg")
    for lo in range(0, 160, 64):
        ids = order[lo:lo + 64]
        dev_batch = rs.batch(ids, ids_dev[lo:lo + 64])
        host_batch = Batch.from_data_list([graphs[i] for i in ids]).to("cuda")
        assert_same_batch(dev_batch, host_batch)
        la = float(tr_a.train_step(dev_batch))
        lb = float(tr_b.train_step(host_batch))
        assert la == lb


@pytest.mark.parametrize("need_weights", [False, True])
def test_topology_from_the_resident_set(need_weights):
    import deeprank_gnn_amd.synthetic as synth
    check_set_topology(ragged_graphs(4, 6), "cuda", need_weights=need_weights, batch_size=4)
    check_set_topology(fixture_graphs(), "cuda", need_weights=need_weights, batch_size=3)
    check_set_topology([synth.make_graph(i) for i in range(96)], "cuda", need_weights=need_weights, batch_size=64)
