"""Native epoch loop (drgnn_train_epoch) on the MI355X against stepping host-collated mini-batches one by one."""
import pytest

from collate_check import ragged_graphs
from epoch_check import check_epoch
from helpers import fixture_graphs
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("Net,task,bs", [(GINet, "reg", 4), (sGAT, "reg", 3), (FoutNet, "reg", 9), (GINet, "class", 2)])
def test_epoch_ragged(Net, task, bs):
    check_epoch(Net, ragged_graphs(11, 12), 12, task, "cuda", bs)


def test_epoch_fixture():
    check_epoch(GINet, fixture_graphs(), 28, "reg", "cuda", 4)


@pytest.mark.parametrize("Net", [GINet, sGAT, FoutNet])
def test_epoch_full_size(Net):
    """SYN graphs at BASELINE size, 200 graphs in mini-batches of 64 (ragged last batch), two epochs."""
    import deeprank_gnn_amd.synthetic as synth
    check_epoch(Net, [synth.make_graph(i) for i in range(200)], 32, "reg", "cuda", 64)


@pytest.mark.parametrize("Net,task,bs", [(GINet, "reg", 4), (sGAT, "reg", 3), (FoutNet, "class", 5)])
def test_epoch_cached_topology_ragged(Net, task, bs):
    check_epoch(Net, ragged_graphs(11, 12), 12, task, "cuda", bs, cached=True)


@pytest.mark.parametrize("Net", [GINet, sGAT, FoutNet])
def test_epoch_cached_topology_full_size(Net):
    """Cached-topology mode at BASELINE size: same bits as rebuilding the topology of every mini-batch."""
    import deeprank_gnn_amd.synthetic as synth
    check_epoch(Net, [synth.make_graph(i) for i in range(200)], 32, "reg", "cuda", 64, cached=True)


@pytest.mark.parametrize("net_name", ["GINet", "sGAT", "FoutNet"])
def test_native_epochs_match_oracle_training(net_name):
    """Two shuffled epochs of the native loop over a resident set of 40 SYN graphs (mini-batches of 16, ragged last one)
    against the CPU oracle trained with torch.optim.Adam on host-collated mini-batches in the same order: per-batch
    losses, predictions and final parameters within 1e-4."""
    import numpy as np
    import torch
    import torch.nn.functional as F
    import deeprank_gnn_amd.synthetic as synth
    from oracle import cpu_ref
    from test_gpu_parity import build
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.trainer import FusedTrainer
    graphs = [synth.make_graph(i, n_nodes=120, n_pairs=260) for i in range(40)]
    params = cpu_ref.init_params(net_name, 32, 1, 1, seed=5)
    leaves = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    opt = torch.optim.Adam(list(leaves.values()), lr=0.01)
    net = build(net_name, params, 1)              # dropout forced to 0 for parity
    tr = FusedTrainer(net, lr=0.01, task="reg")
    rs = ResidentGraphSet(graphs, "cuda")
    kw = {"looped": False} if net_name == "FoutNet" else {}
    rng = np.random.default_rng(1)
    for _ in range(2):
        order = rng.permutation(40).tolist()
        losses, pred = tr.train_epoch(rs, order, 16)
        want_l, want_p = [], []
        for lo in range(0, 40, 16):
            b = Batch.from_data_list([graphs[i] for i in order[lo:lo + 16]])
            opt.zero_grad()
            out = cpu_ref.FORWARD[net_name](leaves, b, **kw)
            loss = F.mse_loss(out.reshape(-1), b.y)
            loss.backward()
            opt.step()
            want_l.append(float(loss.detach()))
            want_p.append(out.detach().reshape(-1))
        np.testing.assert_allclose(losses.cpu().numpy(), want_l, rtol=1e-4)
        np.testing.assert_allclose(pred.cpu().reshape(-1).numpy(), torch.cat(want_p).numpy(), rtol=1e-4, atol=1e-4)
    sd = net.state_dict()
    for k, v in leaves.items():
        np.testing.assert_allclose(sd[k].cpu().numpy(), v.detach().numpy(), rtol=1e-4, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("cached", [False, True])
def test_pipelined_epochs_equal_synchronised_epochs(cached):
    """Thirty short epochs enqueued back to back WITHOUT a host synchronisation (the id upload of epoch e+1 runs on its own
    stream under epoch e, its `losses` are not pre-filled, trainer.loss is filled on first read) against the same epochs with
    a synchronisation after every one: losses, predictions and final parameters bit for bit."""
    import copy
    import numpy as np
    import torch
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.trainer import FusedTrainer
    graphs = [synth.make_graph(i, n_nodes=80, n_pairs=160) for i in range(96)]
    torch.manual_seed(2)
    net = GINet(32, 1, 1).to("cuda")
    rs = ResidentGraphSet(graphs, "cuda")
    ta = FusedTrainer(net, lr=1e-2, task="reg", seed=3)
    tb = FusedTrainer(copy.deepcopy(net), lr=1e-2, task="reg", seed=3)
    rng = np.random.default_rng(4)
    orders = [rng.permutation(96).tolist() for _ in range(30)]
    pending = [ta.train_epoch(rs, o, 16, cached=cached) for o in orders]          # six mini-batches each, nobody waits
    want = []
    for o in orders:
        lo, pr = tb.train_epoch(rs, o, 16, cached=cached)
        torch.cuda.synchronize()
        want.append((lo.cpu().clone(), pr.cpu().clone(), float(tb.loss)))
    torch.cuda.synchronize()
    assert ta.faults() == 0 and tb.faults() == 0
    for (la, pa), (lb, pb, last) in zip(pending, want):
        assert torch.equal(la.cpu(), lb) and torch.equal(pa.cpu(), pb)
        assert float(lb[-1]) == last
    assert float(ta.loss) == float(want[-1][0][-1])
    assert torch.equal(ta.flat_p.cpu(), tb.flat_p.cpu()) and torch.equal(ta.exp_avg_sq.cpu(), tb.exp_avg_sq.cpu())
