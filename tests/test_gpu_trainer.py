"""Native training step on the MI355X vs the CPU oracle trained with torch.optim.Adam."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("net_name", ["GINet", "sGAT", "FoutNet"])
def test_five_native_steps_match_oracle_training(net_name):
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.trainer import FusedTrainer
    from test_gpu_parity import build
    dev = torch.device("cuda:0")
    batch_cpu = synth.make_batch(0, 16, n_nodes=120, n_pairs=260)
    params = cpu_ref.init_params(net_name, 32, 1, 1, seed=9)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=0.01)
    net = build(net_name, params, 1)              # dropout forced to 0 for parity
    tr = FusedTrainer(net, lr=0.01, task="reg")
    batch = batch_cpu.clone().to(dev)
    kw = {"looped": False} if net_name == "FoutNet" else {}
    for it in range(5):
        opt.zero_grad()
        pred = cpu_ref.FORWARD[net_name](leaves, batch_cpu, **kw)
        loss = F.mse_loss(pred.reshape(-1), batch_cpu.y)
        loss.backward()
        opt.step()
        got = tr.train_step(batch)
        np.testing.assert_allclose(float(got), float(loss.detach()), rtol=1e-4)
    sd = net.state_dict()
    for k, v in leaves.items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v.detach().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)
    # inference path
    np.testing.assert_allclose(tr.predict(batch).cpu().numpy(),
                               cpu_ref.FORWARD[net_name]({k: v.detach() for k, v in leaves.items()}, batch_cpu, **kw).numpy(),
                               rtol=1e-4, atol=1e-4)


def test_native_step_is_graph_capturable_and_deterministic():
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.trainer import FusedTrainer
    dev = torch.device("cuda:0")
    batch = synth.make_batch(0, 8, n_nodes=100, n_pairs=200).to(dev)
    results = []
    for rep in range(2):
        torch.manual_seed(0)
        net = GINet(32, 1, 1).to(dev)             # dropout 0.4 active: counter-based, reproducible
        tr = FusedTrainer(net, lr=1e-3, seed=7)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            tr.train_step(batch)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            tr.train_step(batch)
        for _ in range(4):
            g.replay()
        torch.cuda.synchronize()
        assert int(tr.step) == 5          # 1 eager warm-up + 4 replays (capture itself does not execute)
        results.append(tr.flat_p.clone())
    assert torch.equal(results[0], results[1])


@pytest.mark.parametrize("net_name,task,target", [("GINet", None, "irmsd"), ("GINet", "class", "binclass"),
                                                  ("FoutNet", None, "irmsd"), ("sGAT", None, "irmsd")])
def test_neuralnet_counterpart_on_gpu(tmp_path, net_name, task, target):
    """The reference's four tests/test_nn.py flows (test_ginet, test_ginet_class, test_fout, test_sgat: train 5 epochs
    with validation, save, reload as pretrained model; tests/test_nn.py:9-32) on the device."""
    import os
    from helpers import GOLDEN, NODE_FEATURES
    from deeprank_gnn_amd.NeuralNet import NeuralNet
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.foutnet import FoutNet
    from deeprank_gnn_amd.sGAT import sGAT
    Net = {"GINet": GINet, "FoutNet": FoutNet, "sGAT": sGAT}[net_name]
    db = os.path.join(GOLDEN, "fixture_1ATN.npz")
    torch.manual_seed(0)
    np.random.seed(0)
    nn = NeuralNet(db, Net, node_feature=NODE_FEATURES, edge_feature=['dist'], target=target, task=task,
                   batch_size=64, percent=[0.8, 0.2], outdir=str(tmp_path))
    nn.train(nepoch=5, validate=True)
    assert len(nn.train_loss) == 5 and len(nn.valid_loss) == 5 and all(np.isfinite(nn.train_loss))
    if task is None:
        assert nn.train_loss[-1] < nn.train_loss[0]
    ck = os.path.join(str(tmp_path), 'test.pth.tar')
    nn.save_model(ck)
    cpy = NeuralNet(db, Net, pretrained_model=ck, outdir=str(tmp_path))
    a, b = cpy.test(hdf5=None), nn.test(hdf5=None)
    np.testing.assert_allclose(a['raw_outputs'], b['raw_outputs'], rtol=1e-6)
    assert a['mol'] == b['mol'] and len(a['mol']) == 10


def test_pipelined_topology_co_launch_matches_plain_training():
    """The next batch's topology built inside this step's backward launch (co-launched workgroups)."""
    from helpers import fixture_graphs
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd.trainer import FusedTrainer
    import copy
    dev = torch.device("cuda:0")
    graphs = fixture_graphs(count=10)
    batches = [Batch.from_data_list(graphs[0:4]).to(dev), Batch.from_data_list(graphs[4:7]).to(dev),
               Batch.from_data_list(graphs[7:10]).to(dev)]
    torch.manual_seed(2)
    a = GINet(28, 1, 1).to(dev)
    a.dropout = 0.0
    b = copy.deepcopy(a)
    ta, tb = FusedTrainer(a, lr=0.01), FusedTrainer(b, lr=0.01)
    topo = Topology.from_batch(batches[0], need_weights=False)
    for i, batch in enumerate(batches):
        nxt = Topology.from_batch(batches[i + 1], need_weights=False, build=False) if i + 1 < len(batches) else None
        la = float(ta.train_step(batch, topo=topo, next_topo=nxt))
        lb = float(tb.train_step(batch))
        assert la == lb
        if nxt is not None:
            assert nxt.status()[0] == 0
        topo = nxt
    assert torch.equal(ta.flat_p, tb.flat_p)


@pytest.mark.parametrize("net_name", ["GINet", "sGAT", "FoutNet"])
@pytest.mark.parametrize("n_feat,task", [(5, "reg"), (16, "class"), (40, "reg"), (32, "reg")])
def test_fused_step_matches_launch_pair_on_ragged_batches(net_name, n_feat, task):
    """Fused training-step launch (incl. the cross-workgroup exchange of the two GINet branches, the generic and
    the width-specialised instantiations, odd feature widths, single-node graphs) vs forward + backward launches."""
    from step_check import check_fused_matches_pair
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.sGAT import sGAT
    from deeprank_gnn_amd.foutnet import FoutNet
    Net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[net_name]
    assert check_fused_matches_pair(Net, n_feat, task, torch.device("cuda:0"), seed=n_feat)


@pytest.mark.parametrize("net_name,n_feat,task", [("GINet", 32, "reg"), ("GINet", 5, "class"), ("sGAT", 16, "reg"),
                                                  ("FoutNet", 40, "reg")])
def test_fused_inference_launch(net_name, n_feat, task):
    from step_check import check_fused_predict
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.sGAT import sGAT
    from deeprank_gnn_amd.foutnet import FoutNet
    Net = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}[net_name]
    check_fused_predict(Net, n_feat, task, torch.device("cuda:0"), seed=7 + n_feat)


def test_fused_step_full_size_properties():
    """BASELINE size (64 graphs x 200 nodes): (1) the fused training step is bit-reproducible, (2) a graph's
    prediction does not depend on the rest of the batch (fused inference on the batch == on the graph alone)."""
    import copy
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.trainer import FusedTrainer
    dev = torch.device("cuda:0")
    batch = synth.make_batch(0, 64).to(dev)
    torch.manual_seed(3)
    net = GINet(32, 1, 1).to(dev)
    runs = []
    for _ in range(2):
        tr = FusedTrainer(copy.deepcopy(net), lr=1e-3, seed=9)
        for _ in range(3):
            tr.train_step(batch)
        runs.append((tr.flat_p.clone(), tr.flat_g.clone(), tr.loss.clone(), tr.last_pred.clone()))
    for a, b in zip(runs[0], runs[1]):
        assert torch.equal(a, b)
    tr = FusedTrainer(copy.deepcopy(net), lr=1e-3)
    full = tr.predict(batch).cpu().numpy()
    for g in (0, 17, 63):
        single = synth.make_batch(g, 1).to(dev)
        np.testing.assert_array_equal(tr.predict(single).cpu().numpy()[0], full[g])


@pytest.mark.parametrize("net_name", ["GINet", "sGAT"])
def test_transform_sigmoid_on_the_device(net_name):
    """transform_sigmoid (reference NeuralNet.py:616-631) in the fused step's head: 3 steps vs the oracle trained on
    sigmoid(pred) with torch Adam."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.trainer import FusedTrainer
    from test_gpu_parity import build
    dev = torch.device("cuda:0")
    batch_cpu = synth.make_batch(0, 12, n_nodes=90, n_pairs=200)
    batch_cpu.y = torch.rand(12, generator=torch.Generator().manual_seed(3))
    params = cpu_ref.init_params(net_name, 32, 1, 1, seed=4)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=0.01)
    net = build(net_name, params, 1)
    tr = FusedTrainer(net, lr=0.01, task="reg", transform_sigmoid=True)
    batch = batch_cpu.clone().to(dev)
    for it in range(3):
        opt.zero_grad()
        pred = torch.sigmoid(cpu_ref.FORWARD[net_name](leaves, batch_cpu).reshape(-1))
        loss = F.mse_loss(pred, batch_cpu.y)
        loss.backward()
        opt.step()
        got = tr.train_step(batch)
        np.testing.assert_allclose(float(got), float(loss.detach()), rtol=1e-4)
        np.testing.assert_allclose(tr.last_pred.reshape(-1).cpu().numpy(), pred.detach().numpy(), rtol=1e-4, atol=1e-5)
    sd = net.state_dict()
    for k, v in leaves.items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v.detach().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


def test_resident_set_native_image_on_the_device(tmp_path):
    """save_native (set + cached topology) -> load_native on the MI355X: training out of the stored topology gives the
    bits of rebuilding every mini-batch's topology."""
    import copy
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.trainer import FusedTrainer
    graphs = [synth.make_graph(i, n_nodes=150, n_pairs=330) for i in range(40)]
    rs = ResidentGraphSet(graphs, "cuda")
    path = str(tmp_path / "set.drgs")
    rs.save_native(path)
    back = ResidentGraphSet.load_native(path, "cuda")
    assert True in back._topo_cache
    torch.manual_seed(1)
    net = GINet(32, 1, 1).to("cuda")
    ta, tb = FusedTrainer(net, lr=0.01, seed=3), FusedTrainer(copy.deepcopy(net), lr=0.01, seed=3)
    order = np.random.default_rng(0).permutation(40).tolist()
    la, pa = ta.train_epoch(rs, order, 16)
    lb, pb = tb.train_epoch(back, order, 16, cached=True)
    assert torch.equal(la, lb) and torch.equal(pa, pb) and torch.equal(ta.flat_p, tb.flat_p)
