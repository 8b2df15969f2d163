"""Edge cases on the MI355X: ragged batches, graphs without edges, single-node graphs, graphs too
large for LDS (global-scratch variant of every kernel), batches larger than the CU count."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cpu_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def run_vs_oracle(net_name, batch_cpu, n_feat, seed=0, tol=1e-4):
    from test_gpu_parity import build
    from deeprank_gnn_amd.topology import Topology
    params = cpu_ref.init_params(net_name, n_feat, 1, 1, seed=seed)
    kw = {"looped": False} if net_name == "FoutNet" else {}
    ref_pred, ref_loss, ref_grads = cpu_ref.loss_and_grads(net_name, params, batch_cpu, batch_cpu.y, **kw)
    net = build(net_name, params, 1)
    batch = batch_cpu.clone().to(DEV)
    topo = Topology.from_batch(batch, check=True, need_weights=(net_name == "sGAT"))
    out = net(batch, topo=topo)
    loss = F.mse_loss(out.reshape(-1), batch.y)
    loss.backward()
    from elementwise import Lazy64, check_step
    check_step(net_name, Lazy64(net_name, params, batch_cpu, **kw), loss.item(), out.detach().cpu().numpy(),
               {k: p.grad.cpu().numpy() for k, p in net.named_parameters()}, ref_loss, ref_pred.numpy(),
               {k: v.numpy() for k, v in ref_grads.items()})


@pytest.mark.parametrize("net_name", ["GINet", "sGAT", "FoutNet"])
def test_ragged_batch_with_degenerate_graphs(net_name):
    from test_emu_topology import random_graph
    from deeprank_gnn_amd.data import Batch
    rng = np.random.default_rng(11)
    graphs = []
    for k in range(10):
        n = int(rng.integers(2, 70))
        e = int(rng.integers(1, 5 * n))
        graphs.append(random_graph(rng, n, e, int(rng.integers(1, n + 1)), int(rng.integers(1, 5)), sym=True,
                                   self_loops=(k == 3), dup=(k == 4)))
    graphs.insert(2, random_graph(rng, 1, 0, 1, 1))                # single node, no edges
    graphs.insert(5, random_graph(rng, 9, 0, 3, 2))                # nodes but no edges at all
    for g in graphs:
        g.y = torch.tensor([float(rng.uniform(0, 5))])
    run_vs_oracle(net_name, Batch.from_data_list(graphs), 5, seed=1)


@pytest.mark.parametrize("net_name", ["GINet", "sGAT"])
def test_graph_too_large_for_lds_runs_from_global_scratch(net_name):
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.data import Batch
    big = synth.make_graph(3, n_nodes=1400, n_pairs=4000, n_feat=32, n_c1=40, n_internal=50)
    small = synth.make_graph(4, n_nodes=60, n_pairs=100, n_feat=32, n_c1=4, n_internal=10)
    batch = Batch.from_data_list([small, big, small.clone()])
    api = _lib.get()
    assert api.net_lds_bytes(_lib.GINET, 32, 1400, big.edge_index.size(1), 350, False) > 160 * 1024
    run_vs_oracle(net_name, batch, 32, seed=2)


def test_more_graphs_than_compute_units():
    import deeprank_gnn_amd.synthetic as synth
    batch = synth.make_batch(0, 300, n_nodes=30, n_pairs=50, n_feat=16, n_c1=3, n_internal=10)
    run_vs_oracle("GINet", batch, 16, seed=3)


def test_native_trainer_on_a_ragged_batch_with_class_weights():
    """CrossEntropy + class weights + 3 classes through the fused per-graph head."""
    import copy
    from test_emu_topology import random_graph
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.trainer import FusedTrainer
    from deeprank_gnn_amd.topology import Topology
    rng = np.random.default_rng(5)
    graphs = [random_graph(rng, int(rng.integers(3, 40)), int(rng.integers(2, 90)), 5, 2) for _ in range(21)]
    for g in graphs:
        g.y = torch.tensor([int(rng.integers(0, 3))])
    batch = Batch.from_data_list(graphs).to(DEV)
    torch.manual_seed(0)
    ref = GINet(5, 3, 1).to(DEV)
    ref.dropout = 0.0
    net = copy.deepcopy(ref)
    cw = torch.tensor([0.2, 0.5, 0.3], device=DEV)
    opt = torch.optim.Adam(ref.parameters(), lr=0.01)
    tr = FusedTrainer(net, lr=0.01, task="class", class_weights=cw)
    for _ in range(4):
        opt.zero_grad()
        out = ref(batch, topo=Topology.from_batch(batch, need_weights=False))
        loss = F.cross_entropy(out, batch.y.view(-1), weight=cw)
        loss.backward()
        opt.step()
        got = tr.train_step(batch)
        np.testing.assert_allclose(float(got), float(loss.detach()), rtol=1e-4)
    for (k, p), (_, q) in zip(net.named_parameters(), ref.named_parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.gpu
def test_scatter_functions_with_an_index_of_the_shape_of_src():
    """torch_scatter's general form (index broadcast to src's shape: every column its own index vector) -- not used by the
    reference (1-D indices: ginet.py:71,133), served column by column: against a plain loop."""
    from deeprank_gnn_amd.community_pooling import scatter_sum, scatter_mean, scatter_max
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(5)
    src = torch.randn((30, 4), generator=gen)
    index = torch.randint(0, 7, (30, 4), generator=gen)
    want_sum = torch.zeros((7, 4)).scatter_add_(0, index, src)
    cnt = torch.zeros((7, 4)).scatter_add_(0, index, torch.ones_like(src)).clamp(min=1)
    want_max = torch.zeros((7, 4))
    want_arg = torch.full((7, 4), 30, dtype=torch.int64)
    for j in range(4):
        for k in range(7):
            rows = (index[:, j] == k).nonzero().reshape(-1)
            if rows.numel():
                v, a = src[rows, j].max(dim=0)
                want_max[k, j], want_arg[k, j] = v, rows[a]
    s, i = src.to(dev), index.to(dev)
    np.testing.assert_allclose(scatter_sum(s, i, dim=0).cpu().numpy(), want_sum.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(scatter_mean(s, i, dim=0).cpu().numpy(), (want_sum / cnt).numpy(), rtol=1e-5, atol=1e-6)
    mx, arg = scatter_max(s, i, dim=0)
    np.testing.assert_allclose(mx.cpu().numpy(), want_max.numpy(), rtol=1e-6)
    np.testing.assert_array_equal(arg.cpu().numpy(), want_arg.numpy())
    # along the last dimension, with dim_size
    got = scatter_sum(s.t().contiguous(), i.t().contiguous(), dim=1, dim_size=9)
    np.testing.assert_allclose(got.cpu().numpy(), torch.cat([want_sum, torch.zeros((2, 4))]).t().numpy(), rtol=1e-5, atol=1e-6)
