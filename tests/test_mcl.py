"""Offline clustering: the MCL oracle reproduces the reference's stored
clustering/mcl/depth_0 and depth_1 of every fixture graph exactly.  CPU only."""
import numpy as np
import pytest

from helpers import GOLDEN
from oracle import mcl_ref


def fixture():
    z = np.load(GOLDEN + "/fixture_1ATN.npz")
    return z, [str(m) for m in z["__mols__"]]


def internal_edges(z, mol):
    pairs = z[mol + "/internal_edge_index"]
    return np.vstack((pairs, pairs[:, ::-1])).T


@pytest.mark.parametrize("k", range(10))
def test_oracle_reproduces_stored_clustering(k):
    z, mols = fixture()
    mol = mols[k]
    n = z[mol + "/node_data/pos"].shape[0]
    d0, d1 = mcl_ref.precluster(internal_edges(z, mol), n)
    np.testing.assert_array_equal(d0, z[mol + "/clustering/mcl/depth_0"])
    np.testing.assert_array_equal(d1, z[mol + "/clustering/mcl/depth_1"])


def _fixture_batch_without_clusters():
    from helpers import fixture_graphs
    from deeprank_gnn_amd.data import Batch
    graphs = fixture_graphs(count=None)
    expect0 = np.concatenate([g.cluster0.numpy() for g in graphs])
    expect1 = np.concatenate([g.cluster1.numpy() for g in graphs])
    for g in graphs:
        g.cluster0 = None
        g.cluster1 = None
    return Batch.from_data_list(graphs), expect0, expect1


def test_device_precluster_emulated_reproduces_the_fixture():
    """drgnn_mcl + pooling through the topology builder (host-emulation build) == the labels the
    reference's real PreCluster stored for all 10 graphs, both depths."""
    from emu_api import emu
    from deeprank_gnn_amd.clustering import precluster
    batch, expect0, expect1 = _fixture_batch_without_clusters()
    d0, d1 = precluster(batch, api=emu())
    np.testing.assert_array_equal(d0.numpy(), expect0)
    np.testing.assert_array_equal(d1.numpy(), expect1)


def test_community_detection_function_on_the_reference_toy_graph(monkeypatch):
    """6-node path-pair graph of reference tests/test_community_pooling.py:12-19."""
    import torch
    from emu_api import emu
    from deeprank_gnn_amd import community_pooling as cp
    monkeypatch.setattr(cp, "_API", emu())
    ei = torch.tensor([[0, 1, 1, 2, 3, 4, 4, 5], [1, 0, 2, 1, 4, 3, 5, 4]])
    got = cp.community_detection(ei, 6, method='mcl').numpy()
    np.testing.assert_array_equal(got, mcl_ref.community_detection_mcl(ei.numpy(), 6))
    assert got[0] == got[1] == got[2] and got[3] == got[4] == got[5] and got[0] != got[3]
    with pytest.raises(ValueError):
        cp.community_detection(ei, 6, method='xxx')          # reference: expectedFailure test
    per_batch = cp.community_detection_per_batch(torch.cat([ei, ei + 6], 1), torch.tensor([0] * 6 + [1] * 6), 12)
    assert per_batch.tolist() == [0, 0, 0, 1, 1, 1, 1, 1, 1, 2, 2, 2]      # the reference's shared-id offset


def test_dataset_precluster_and_training_without_stored_clusters(tmp_path):
    """A graph file WITHOUT clustering/ groups: NeuralNet pre-clusters it on construction (as the
    reference does) and the labels equal the ones the reference stored."""
    import os
    from emu_api import emu
    from deeprank_gnn_amd.dataset import GraphStore, GraphDataSet
    from deeprank_gnn_amd.NeuralNet import NeuralNet
    from deeprank_gnn_amd.ginet import GINet
    from helpers import NODE_FEATURES
    full = GraphStore(GOLDEN + "/fixture_1ATN.npz")
    bare = GraphStore(GOLDEN + "/fixture_1ATN.npz")
    for mol in bare.mols():
        for k in [k for k in bare._mols[mol] if k.startswith("clustering/")]:
            del bare._mols[mol][k]
    path = os.path.join(str(tmp_path), "bare.npz")
    bare.save_npz(path)
    nn = NeuralNet(path, GINet, node_feature=NODE_FEATURES, edge_feature=['dist'], target='irmsd', batch_size=64,
                   percent=[0.8, 0.2], outdir=str(tmp_path), _api=emu(), device='cpu')
    for mol in full.mols():
        for depth in ("depth_0", "depth_1"):
            np.testing.assert_array_equal(nn.dataset.store.get(mol, "clustering/mcl/" + depth),
                                          full.get(mol, "clustering/mcl/" + depth))
    nn.train(nepoch=1, validate=False, save_model=None, hdf5=None)
    assert np.isfinite(nn.train_loss[0])
