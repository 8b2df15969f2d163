"""Stand-alone layers and the function-level pooling API (emulated kernels) vs goldens
recorded from the reference's own functions and vs the oracle.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden, fixture_batch, fixture_graphs
from emu_api import emu
from oracle import cpu_ref
import deeprank_gnn_amd.synthetic as synth
from deeprank_gnn_amd import _lib, community_pooling as cp
from deeprank_gnn_amd.data import Batch, Data
from deeprank_gnn_amd.ginet import GINetConvLayer
from deeprank_gnn_amd.sGAT import sGraphAttentionLayer
from deeprank_gnn_amd.foutnet import FoutLayer

TOL = dict(rtol=1e-4, atol=1e-5)


@pytest.fixture(autouse=True)
def use_emulation(monkeypatch):
    monkeypatch.setattr(_lib, "get", lambda: emu())
    monkeypatch.setattr(cp, "_API", emu())


def test_conv_layers_isolated_node_golden():
    g = golden("layers_isolated.npz")
    gr = synth.make_graph(7, n_nodes=24, n_pairs=40, n_feat=6, n_c1=3, n_internal=10, isolate_node=5)
    t = lambda k: torch.from_numpy(g[k].copy())
    lay = GINetConvLayer(6, 16, 1)
    lay.load_state_dict({"fc.weight": t("ginet.fc"), "fc_edge_attr.weight": t("ginet.fc_edge_attr"),
                         "fc_attention.weight": t("ginet.fc_attention")})
    np.testing.assert_allclose(lay(gr.x, gr.edge_index, gr.edge_attr).detach().numpy(), g["ginet.out"], **TOL)
    lay = sGraphAttentionLayer(6, 16)
    lay.load_state_dict({"weight": t("sgat.weight"), "bias": t("sgat.bias")})
    out = lay(gr.x, gr.edge_index, gr.edge_attr).detach().numpy()
    np.testing.assert_allclose(out, g["sgat.out"], **TOL)
    np.testing.assert_array_equal(out[5], g["sgat.bias"])                  # isolated node -> bias row
    lay = FoutLayer(6, 16)
    lay.load_state_dict({"Wc": t("fout.Wc"), "Wn": t("fout.Wn"), "bias": t("fout.bias")})
    out = lay(gr.x, gr.edge_index).detach().numpy()
    assert np.isnan(out[5]).all() and np.isnan(g["fout.out"][5]).all()      # NaN parity
    keep = np.arange(24) != 5
    np.testing.assert_allclose(out[keep], g["fout.out"][keep], **TOL)


@pytest.mark.parametrize("kind", ["ginet", "sgat", "fout"])
@pytest.mark.parametrize("width", [24, 7])
def test_conv_layer_any_width_forward_backward_vs_oracle(kind, width):
    torch.manual_seed(4)
    gr = synth.make_graph(3, n_nodes=150, n_pairs=300, n_feat=10, n_c1=3, n_internal=10)
    x = gr.x.clone().requires_grad_(True)
    xr = gr.x.clone().requires_grad_(True)
    if kind == "ginet":
        lay = GINetConvLayer(10, width, 1)
        out = lay(x, gr.edge_index, gr.edge_attr)
        ref = cpu_ref.ginet_conv(xr, gr.edge_index, gr.edge_attr, lay.fc.weight.detach().clone().requires_grad_(True),
                                 lay.fc_edge_attr.weight.detach(), lay.fc_attention.weight.detach())
        rp = None
    elif kind == "sgat":
        lay = sGraphAttentionLayer(10, width)
        out = lay(x, gr.edge_index, gr.edge_attr)
        rp = [lay.weight.detach().clone().requires_grad_(True), lay.bias.detach().clone().requires_grad_(True)]
        ref = cpu_ref.sgat_conv(xr, gr.edge_index, gr.edge_attr, *rp)
    else:
        lay = FoutLayer(10, width)
        out = lay(x, gr.edge_index)
        rp = [lay.Wc.detach().clone().requires_grad_(True), lay.Wn.detach().clone().requires_grad_(True),
              lay.bias.detach().clone().requires_grad_(True)]
        ref = cpu_ref.fout_conv(xr, gr.edge_index, *rp, looped=False)
    np.testing.assert_allclose(out.detach().numpy(), ref.detach().numpy(), **TOL)
    wgt = torch.randn_like(ref)
    (out * wgt).sum().backward()
    (ref * wgt).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-4)
    if kind == "ginet":
        w = lay.fc.weight.detach().clone().requires_grad_(True)
        r2 = cpu_ref.ginet_conv(gr.x, gr.edge_index, gr.edge_attr, w, lay.fc_edge_attr.weight.detach(), lay.fc_attention.weight.detach())
        (r2 * wgt).sum().backward()
        np.testing.assert_allclose(lay.fc.weight.grad.numpy(), w.grad.numpy(), rtol=1e-4, atol=1e-4)
        assert lay.fc_attention.weight.grad.abs().max() == 0 and lay.fc_edge_attr.weight.grad.abs().max() == 0
    else:
        for p, q in zip(lay.parameters(), rp):
            np.testing.assert_allclose(p.grad.numpy(), q.grad.numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("fname", ["fix8_GINet.npz", "fix8_sGAT.npz"])
def test_pooling_functions_vs_reference_golden(fname):
    g = golden(fname)
    batch = fixture_batch(8)
    # get_preloaded_cluster: in place, returns the same tensor
    cl = batch.cluster0.clone()
    ret = cp.get_preloaded_cluster(cl, batch.batch)
    assert ret is cl
    np.testing.assert_array_equal(cl.numpy(), g["a.cluster0_offset"])
    # community_pooling on relu(conv1 output)
    z1 = torch.from_numpy(g["a.z1"]).clone().requires_grad_(True)
    batch.x = F.relu(z1)
    pooled = cp.community_pooling(cl, batch)
    np.testing.assert_allclose(pooled.x.detach().numpy(), g["a.xp"], **TOL)
    np.testing.assert_array_equal(pooled.edge_index.numpy(), g["a.pool_edge_index"])
    np.testing.assert_allclose(pooled.edge_attr.numpy(), g["a.pool_edge_attr"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(pooled.batch.numpy(), g["a.pool_batch"])
    np.testing.assert_array_equal(pooled.internal_edge_index.numpy(), g["a.pool_internal_edge_index"])
    np.testing.assert_allclose(pooled.pos.numpy(), g["a.pool_pos"], rtol=1e-5, atol=1e-5)
    assert pooled.cluster1 is batch.cluster1 and pooled.num_graphs == 8
    # gradient of the max-pool flows to the arg-max rows only (vs oracle scatter_max)
    pooled.x.sum().backward()
    zr = torch.from_numpy(g["a.z1"]).clone().requires_grad_(True)
    cons, _ = cpu_ref.consecutive_cluster(torch.from_numpy(g["a.cluster0_offset"]))
    cpu_ref.scatter_max(F.relu(zr), cons)[0].sum().backward()
    np.testing.assert_allclose(z1.grad.numpy(), zr.grad.numpy(), **TOL)
    # second level: get_preloaded_cluster on the pooled batch + max_pool_x
    cl1 = cp.get_preloaded_cluster(pooled.cluster1.clone(), pooled.batch)
    np.testing.assert_array_equal(cl1.numpy(), g["a.cluster1_offset"])
    x2, b2 = cp.max_pool_x(cl1, F.relu(torch.from_numpy(g["a.z2"])), pooled.batch)
    np.testing.assert_allclose(x2.numpy(), g["a.x2"], **TOL)
    np.testing.assert_array_equal(b2.numpy(), g["a.batch2"])
    # readout: scatter_mean(x, batch, dim=0)
    ro = cp.scatter_mean(x2, b2, dim=0)
    np.testing.assert_allclose(ro.numpy(), g["readout"][:, :32], **TOL)


def test_toy_graph_of_the_reference_test():
    """reference tests/test_community_pooling.py:12-19,53-59 (two copies, 4 clusters)."""
    g = golden("toy6.npz")
    ei = torch.tensor([[0, 1, 1, 2, 3, 4, 4, 5], [1, 0, 2, 1, 4, 3, 5, 4]])
    d = Data(x=torch.arange(6, dtype=torch.float).view(6, 1), edge_index=ei, edge_attr=torch.ones(8, 1),
             pos=torch.arange(18, dtype=torch.float).view(6, 3))
    two = Batch.from_data_list([d.clone(), d.clone()])
    p = cp.community_pooling(torch.from_numpy(g["cluster"]), two)
    np.testing.assert_array_equal(p.x.numpy(), g["x"])
    assert p.edge_index.shape == (2, 0)
    np.testing.assert_array_equal(p.batch.numpy(), g["batch"])
    np.testing.assert_allclose(p.pos.numpy(), g["pos"])
    g = golden("toy6_edges.npz")
    two = Batch.from_data_list([d.clone(), d.clone()])
    two.edge_attr = torch.from_numpy(g["in_edge_attr"])
    p = cp.community_pooling(torch.from_numpy(g["cluster"]), two)
    np.testing.assert_array_equal(p.edge_index.numpy(), g["edge_index"])
    np.testing.assert_array_equal(p.edge_attr.numpy(), g["edge_attr"])
    # a plain Data (no batch vector) pools to a Data
    single = cp.community_pooling(torch.tensor([0, 0, 1, 1, 2, 2]), d.clone())
    assert isinstance(single, Data) and not isinstance(single, Batch) and single.x.shape == (3, 1)


def test_scatter_ops_semantics():
    src = torch.tensor([[1., -2.], [3., 0.5], [-1., 4.], [2., 2.], [0., 7.]], requires_grad=True)
    idx = torch.tensor([2, 0, 2, 0, 5])                       # gaps: ids 1, 3, 4 absent
    ref = src.detach().clone().requires_grad_(True)
    for mine, theirs in ((cp.scatter_sum, cpu_ref.scatter_sum), (cp.scatter_mean, cpu_ref.scatter_mean)):
        a = mine(src, idx, dim=0)
        b = theirs(ref, idx)
        np.testing.assert_allclose(a.detach().numpy(), b.detach().numpy(), rtol=1e-6)
        wgt = torch.arange(12, dtype=torch.float).view(6, 2)
        src.grad = None
        ref.grad = None
        (a * wgt).sum().backward()
        (b * wgt).sum().backward()
        np.testing.assert_allclose(src.grad.numpy(), ref.grad.numpy(), rtol=1e-6)
    a, arg = cp.scatter_max(src, idx, dim=0)
    b, argb = cpu_ref.scatter_max(ref, idx)
    np.testing.assert_array_equal(a.detach().numpy(), b.detach().numpy())
    np.testing.assert_array_equal(arg.numpy(), argb.numpy())


def test_layer_constructor_options_vs_reference_golden():
    """undirected=False / bias=False / GINetConvLayer(bias=True): forward, input gradient and every parameter gradient
    against vectors recorded from the reference's own layer classes; the oracle restates the same options."""
    from layer_option_check import check_layer_options, check_oracle_options
    check_oracle_options()
    check_layer_options("cpu")


def test_scatter_out_argument_like_torch_scatter():
    """scatter_sum / scatter_mean with out= (the reference's sGAT layer calls scatter_mean(alpha, row, dim=0, out=out),
    sGAT.py:82-87): sums are added into `out`; the mean divides the WHOLE buffer by the clamped counts."""
    rng = np.random.default_rng(3)
    src = torch.from_numpy(rng.normal(size=(17, 5)).astype(np.float32))
    index = torch.from_numpy(rng.integers(0, 6, size=17))
    base = torch.from_numpy(rng.normal(size=(8, 5)).astype(np.float32))
    want_sum = base.clone().index_add_(0, index, src)
    out = base.clone()
    got = cp.scatter_sum(src, index, dim=0, out=out)
    assert got is out
    np.testing.assert_allclose(out.numpy(), want_sum.numpy(), rtol=1e-5, atol=1e-6)
    count = torch.bincount(index, minlength=8).clamp(min=1).float().view(-1, 1)
    out = base.clone()
    got = cp.scatter_mean(src, index, dim=0, out=out)
    assert got is out
    np.testing.assert_allclose(out.numpy(), (want_sum / count).numpy(), rtol=1e-5, atol=1e-6)
    # the reference layer's exact call pattern, through autograd
    s2 = src.clone().requires_grad_(True)
    res = cp.scatter_mean(s2, index, dim=0, out=torch.zeros(8, 5))
    res.sum().backward()
    np.testing.assert_allclose(s2.grad.numpy(), (1.0 / count)[index].expand(-1, 5).numpy(), rtol=1e-6)


def test_function_api_corners_against_torch(monkeypatch):
    """The corners of the function API the shipped nets never take (VERDICT r02 missing #5): max_pool_x(size=...),
    scatter_* over another dimension, scatter_max(out=...) -- against plain torch scatter_reduce / index_add."""
    import deeprank_gnn_amd.community_pooling as cp
    monkeypatch.setattr(cp, "_api", lambda: emu())
    rng = np.random.default_rng(5)
    src = torch.from_numpy(rng.standard_normal((4, 9, 3)).astype(np.float32))
    idx = torch.tensor([2, 0, 2, 5, 0, 2, 7, 7, 1])
    # dim = 1 of a 3-D tensor
    for fn, red in ((cp.scatter_sum, "sum"), (cp.scatter_mean, "mean")):
        got = fn(src, idx, dim=1, dim_size=8)
        want = torch.zeros(4, 8, 3).scatter_reduce(1, idx.view(1, -1, 1).expand_as(src), src, reduce=red, include_self=False)
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-6, atol=1e-6)
    got, arg = cp.scatter_max(src, idx, dim=1, dim_size=8)
    want = torch.zeros(4, 8, 3).scatter_reduce(1, idx.view(1, -1, 1).expand_as(src), src, reduce="amax", include_self=False)
    np.testing.assert_array_equal(got.numpy(), want.numpy())
    assert arg.shape == got.shape
    picked = torch.gather(torch.cat([src, torch.zeros(4, 1, 3)], dim=1), 1, arg)
    present = torch.zeros(8, dtype=torch.bool); present[idx] = True
    np.testing.assert_array_equal(picked[:, present].numpy(), want[:, present].numpy())
    assert bool((arg[:, ~present] == 9).all())
    # out=: the running maximum
    x = torch.from_numpy(rng.standard_normal((9, 5)).astype(np.float32))
    out = torch.from_numpy(rng.standard_normal((8, 5)).astype(np.float32))
    want = out.clone().scatter_reduce(0, idx.view(-1, 1).expand_as(x), x, reduce="amax", include_self=True)
    got, arg = cp.scatter_max(x, idx, dim=0, out=out)
    assert got is out
    np.testing.assert_array_equal(out.numpy(), want.numpy())
    # max_pool_x(size=...): fixed slots per graph, no batch vector
    batch = torch.tensor([0, 0, 0, 1, 1, 1, 1, 2, 2])
    cluster = torch.tensor([0, 1, 0, 3, 5, 3, 4, 6, 6])            # 3 slots per graph: ids in [3 g, 3 g + 3)
    got, b = cp.max_pool_x(cluster, x, batch, size=3)
    assert b is None and got.shape == (9, 5)
    want = torch.zeros(9, 5).scatter_reduce(0, cluster.view(-1, 1).expand_as(x), x, reduce="amax", include_self=False)
    np.testing.assert_array_equal(got.numpy(), want.numpy())
