"""Trainer counterpart (deeprank_gnn_amd.NeuralNet) on the fixture, kernels emulated on CPU --
mirrors what the reference's tests/test_nn.py exercises (train 5 epochs, save, reload)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, NODE_FEATURES
from emu_api import emu
from deeprank_gnn_amd.NeuralNet import NeuralNet
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet

DB = os.path.join(GOLDEN, "fixture_1ATN.npz")
REF_KEYS = {'model', 'optimizer', 'node', 'edge', 'target', 'task', 'classes', 'class_weight', 'batch_size',
            'percent', 'lr', 'index', 'shuffle', 'threshold', 'cluster_nodes', 'transform_sigmoid'}


@pytest.mark.parametrize("Net,task,target", [(GINet, None, 'irmsd'), (FoutNet, None, 'irmsd'), (sGAT, None, 'irmsd'),
                                            (GINet, 'class', 'binclass')])
def test_train_save_reload(Net, task, target, tmp_path):
    torch.manual_seed(0)
    np.random.seed(0)
    nn = NeuralNet(DB, Net, node_feature=NODE_FEATURES, edge_feature=['dist'], target=target, task=task,
                   batch_size=64, percent=[0.8, 0.2], outdir=str(tmp_path), _api=emu(), device='cpu')
    nn.train(nepoch=3, validate=True)
    assert len(nn.train_loss) == 3 and len(nn.valid_loss) == 3 and all(np.isfinite(nn.train_loss))
    if task is None:
        assert nn.train_loss[-1] < nn.train_loss[0]              # it learns
    ck = os.path.join(str(tmp_path), 'test.pth.tar')
    nn.save_model(ck)
    state = torch.load(ck, weights_only=False)
    assert set(state) == REF_KEYS                                  # reference checkpoint schema (NeuralNet.py:775-790)
    assert set(state['optimizer']) == {'state', 'param_groups'}
    # a torch Adam over the same parameter list accepts the optimiser state
    ref_net = Net(28, 1 if task is None else 2, 1)
    ref_net.load_state_dict(state['model'], strict=True)
    torch.optim.Adam(ref_net.parameters(), lr=0.01).load_state_dict(state['optimizer'])
    # reload and test: identical predictions
    cpy = NeuralNet(DB, Net, pretrained_model=ck, outdir=str(tmp_path), _api=emu(), device='cpu')
    assert int(cpy.trainer.step) == int(nn.trainer.step)
    a = cpy.test(hdf5=None)
    b = nn.test(hdf5=None)
    np.testing.assert_allclose(a['raw_outputs'], b['raw_outputs'], rtol=1e-6)
    assert a['mol'] == b['mol'] and len(a['mol']) == 10
    # the epoch export: the reference's group / dataset names and group attributes (tests/data/train_ref/train_data.hdf5,
    # layout recorded in tests/golden/train_ref_layout.json), in the native container
    import json
    from deeprank_gnn_amd.container import read_container
    meta, exp = read_container(os.path.join(str(tmp_path), 'train_data.drgs'))
    layout = json.load(open(os.path.join(GOLDEN, 'train_ref_layout.json')))
    (ref_epoch,) = [g for g in layout['groups'] if '/' not in g]          # 'epoch_0005' in the reference's file
    for name, d in layout['datasets'].items():
        mine = 'tree/' + name.replace(ref_epoch, 'epoch_0003')
        assert mine in exp, mine
        assert exp[mine].ndim == d['ndim'] and (exp[mine].dtype.kind == 'S') == (d['kind'] == 'S'), mine
    assert sorted(meta['attrs']['epoch_0003']) == layout['groups'][ref_epoch] == ['batch_size', 'target', 'task']
    assert sorted(exp['tree/epoch_0003/train/mol'].astype(str)) == sorted(nn.dataset.mols[i] for i in nn.train_index)
    assert 'tree/epoch_0003/train/raw_outputs' in exp            # the current reference code also writes these
    assert nn.update_name('train_data.drgs', str(tmp_path)).endswith('train_data_001.drgs')


def test_needs_gpu_without_emulation():
    from deeprank_gnn_amd import _lib
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DrgnnError):
        NeuralNet(DB, GINet, node_feature=NODE_FEATURES, target='irmsd')


@pytest.mark.parametrize("native_epoch", [False, True])
def test_epoch_with_several_batches_uses_the_lookahead(tmp_path, native_epoch):
    """batch_size 3 -> three mini-batches per epoch: the topology of batch k+1 is built inside
    the backward launch of batch k.  Same losses as stepping every batch with its own build, from the
    Python loop and from the native epoch loop (drgnn_train_epoch)."""
    from deeprank_gnn_amd.topology import Topology
    built = {"n": 0}
    orig = Topology.rebuild

    def counting(self, *args):
        built["n"] += 1
        return orig(self, *args)
    torch.manual_seed(0)
    np.random.seed(0)
    nn = NeuralNet(DB, sGAT, node_feature=NODE_FEATURES, edge_feature=['dist'], target='irmsd',
                   batch_size=3, percent=[0.8, 0.2], shuffle=False, outdir=str(tmp_path), _api=emu(), device='cpu')
    nn.native_epoch = native_epoch
    Topology.rebuild = counting
    try:
        nn.train(nepoch=2, validate=False, save_model=None, hdf5=None)
    finally:
        Topology.rebuild = orig
    # Python loop: one explicit build per epoch (first batch), the rest rode along; native loop: none from Python
    assert built["n"] == (0 if native_epoch else 2)
    assert np.isfinite(nn.train_loss).all() and nn.train_loss[1] < nn.train_loss[0]
    # reference run: every batch builds its own topology
    torch.manual_seed(0)
    np.random.seed(0)
    ref = NeuralNet(DB, sGAT, node_feature=NODE_FEATURES, edge_feature=['dist'], target='irmsd',
                    batch_size=3, percent=[0.8, 0.2], shuffle=False, outdir=str(tmp_path), _api=emu(), device='cpu')
    total = []
    for _ in range(2):
        run = 0.0
        for batch in ref._batches(ref.dataset, ref.train_index, False):
            run += float(ref.trainer.train_step(batch))
        total.append(run)
    np.testing.assert_allclose(nn.train_loss, total, rtol=1e-6)


def test_resident_batches_train_like_host_collated_batches(tmp_path):
    """The epoch loop takes its mini-batches from the resident set (device collate); stepping a twin trainer
    over Batch.from_data_list of the same graphs in the same order gives the same losses and predictions."""
    from deeprank_gnn_amd.data import Batch
    torch.manual_seed(0)
    np.random.seed(0)
    kw = dict(node_feature=NODE_FEATURES, edge_feature=['dist'], target='irmsd', batch_size=4, percent=[1.0, 0.0],
              shuffle=False, outdir=str(tmp_path), _api=emu(), device='cpu')
    nn = NeuralNet(DB, FoutNet, **kw)
    torch.manual_seed(0)
    np.random.seed(0)
    twin = NeuralNet(DB, FoutNet, **kw)
    nn.train(nepoch=2, validate=False, save_model=None, hdf5=None)
    losses, outs = [], []
    for _ in range(2):
        run, outs = 0.0, []
        order = list(twin.train_index)
        for lo in range(0, len(order), 4):
            host = Batch.from_data_list([twin.dataset[i] for i in order[lo:lo + 4]])
            run += float(twin.trainer.train_step(host))
            outs += twin.trainer.last_pred.reshape(-1).tolist()
        losses.append(run)
    np.testing.assert_allclose(nn.train_loss, losses, rtol=1e-6)
    np.testing.assert_allclose(nn.data['train']['outputs'], outs, rtol=1e-6)
    assert nn.data['train']['mol'] == [twin.dataset.mols[i] for i in twin.train_index]


def test_cached_topology_is_the_default_and_changes_nothing(tmp_path):
    """``cached_topology = "auto"``: the per-graph topology is built once per resident set when it fits the budget (the
    reference precomputes its clustering once per dataset too, DataSet.py:45-88), else rebuilt per mini-batch -- and either
    way the training run is the same, bit for bit."""
    runs = {}
    for mode in ("auto", False, "auto-no-budget"):
        torch.manual_seed(2)
        np.random.seed(2)
        nn = NeuralNet(DB, GINet, node_feature=NODE_FEATURES, edge_feature=['dist'], target='irmsd', batch_size=4,
                       percent=[1.0, 0.0], shuffle=False, outdir=str(tmp_path), _api=emu(), device='cpu')
        nn.model.dropout = 0.0
        assert nn.cached_topology == "auto"
        if mode is False:
            nn.cached_topology = False
        elif mode == "auto-no-budget":
            nn.topology_cache_budget = 1024
        rs = nn._resident(nn.dataset)
        assert nn._use_cache(rs) == (mode == "auto")
        nn.train(nepoch=2, validate=False)
        runs[mode] = (list(nn.train_loss), {k: v.clone() for k, v in nn.model.state_dict().items()})
    for mode in (False, "auto-no-budget"):
        assert runs[mode][0] == runs["auto"][0]
        for k, v in runs["auto"][1].items():
            assert torch.equal(v, runs[mode][1][k]), k
