"""Stand-alone layers and the function-level pooling API on the MI355X (HIP library)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import golden, fixture_batch
from oracle import cpu_ref

pytestmark = pytest.mark.gpu
TOL = dict(rtol=1e-4, atol=1e-5)
DEV = "cuda:0"


@pytest.mark.parametrize("kind", ["ginet", "sgat", "fout"])
@pytest.mark.parametrize("width", [24, 7])
def test_conv_layer_any_width_vs_oracle(kind, width):
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.ginet import GINetConvLayer
    from deeprank_gnn_amd.sGAT import sGraphAttentionLayer
    from deeprank_gnn_amd.foutnet import FoutLayer
    torch.manual_seed(4)
    batch = synth.make_batch(0, 6, n_nodes=150, n_pairs=300, n_feat=10, n_c1=3, n_internal=10)   # 900 nodes, one block-diagonal graph
    ei, ea = batch.edge_index, batch.edge_attr
    x = batch.x.to(DEV).requires_grad_(True)
    xr = batch.x.clone().requires_grad_(True)
    if kind == "ginet":
        lay = GINetConvLayer(10, width, 1)
        rp = [lay.fc.weight.detach().clone().requires_grad_(True)]
        ref = cpu_ref.ginet_conv(xr, ei, ea, rp[0], lay.fc_edge_attr.weight.detach(), lay.fc_attention.weight.detach())
        live = [lay.fc.weight]
    elif kind == "sgat":
        lay = sGraphAttentionLayer(10, width)
        rp = [lay.weight.detach().clone().requires_grad_(True), lay.bias.detach().clone().requires_grad_(True)]
        ref = cpu_ref.sgat_conv(xr, ei, ea, *rp)
        live = [lay.weight, lay.bias]
    else:
        lay = FoutLayer(10, width)
        rp = [lay.Wc.detach().clone().requires_grad_(True), lay.Wn.detach().clone().requires_grad_(True),
              lay.bias.detach().clone().requires_grad_(True)]
        ref = cpu_ref.fout_conv(xr, ei, *rp, looped=False)
        live = [lay.Wc, lay.Wn, lay.bias]
    lay = lay.to(DEV)
    live = list(lay.parameters()) if kind != "ginet" else [lay.fc.weight]
    out = lay(x, ei.to(DEV)) if kind == "fout" else lay(x, ei.to(DEV), ea.to(DEV))
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), **TOL)
    wgt = torch.randn_like(ref)
    (out * wgt.to(DEV)).sum().backward()
    (ref * wgt).sum().backward()
    np.testing.assert_allclose(x.grad.cpu().numpy(), xr.grad.numpy(), rtol=1e-4, atol=1e-4)
    for p, q in zip(live, rp):
        np.testing.assert_allclose(p.grad.cpu().numpy(), q.grad.numpy(), rtol=1e-4, atol=2e-4)


@pytest.mark.parametrize("fname", ["fix8_GINet.npz", "fix8_sGAT.npz"])
def test_pooling_functions_vs_reference_golden(fname):
    from deeprank_gnn_amd import community_pooling as cp
    g = golden(fname)
    batch = fixture_batch(8).to(DEV)
    cl = batch.cluster0.clone()
    assert cp.get_preloaded_cluster(cl, batch.batch) is cl
    np.testing.assert_array_equal(cl.cpu().numpy(), g["a.cluster0_offset"])
    batch.x = F.relu(torch.from_numpy(g["a.z1"]).to(DEV))
    pooled = cp.community_pooling(cl, batch)
    np.testing.assert_allclose(pooled.x.cpu().numpy(), g["a.xp"], **TOL)
    np.testing.assert_array_equal(pooled.edge_index.cpu().numpy(), g["a.pool_edge_index"])
    np.testing.assert_allclose(pooled.edge_attr.cpu().numpy(), g["a.pool_edge_attr"], rtol=1e-6, atol=1e-6)
    np.testing.assert_array_equal(pooled.batch.cpu().numpy(), g["a.pool_batch"])
    np.testing.assert_array_equal(pooled.internal_edge_index.cpu().numpy(), g["a.pool_internal_edge_index"])
    np.testing.assert_allclose(pooled.pos.cpu().numpy(), g["a.pool_pos"], rtol=1e-5, atol=1e-5)
    cl1 = cp.get_preloaded_cluster(pooled.cluster1.clone(), pooled.batch)
    np.testing.assert_array_equal(cl1.cpu().numpy(), g["a.cluster1_offset"])
    x2, b2 = cp.max_pool_x(cl1, F.relu(torch.from_numpy(g["a.z2"]).to(DEV)), pooled.batch)
    np.testing.assert_allclose(x2.cpu().numpy(), g["a.x2"], **TOL)
    np.testing.assert_array_equal(b2.cpu().numpy(), g["a.batch2"])
    np.testing.assert_allclose(cp.scatter_mean(x2, b2, dim=0).cpu().numpy(), g["readout"][:, :32], **TOL)


def test_custom_net_from_layers_trains():
    """The reference README's custom-GNN pattern: layers + pooling functions + autograd."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd import community_pooling as cp
    from deeprank_gnn_amd.ginet import GINetConvLayer
    batch = synth.make_batch(0, 4, n_nodes=60, n_pairs=120, n_feat=8, n_c1=4, n_internal=10).to(DEV)
    conv1, conv2 = GINetConvLayer(8, 12).to(DEV), GINetConvLayer(12, 20).to(DEV)
    fc = torch.nn.Linear(20, 1).to(DEV)
    opt = torch.optim.Adam(list(conv1.parameters()) + list(conv2.parameters()) + list(fc.parameters()), lr=0.01)
    losses = []
    for _ in range(8):
        opt.zero_grad()
        d = batch.clone()
        d.x = F.relu(conv1(d.x, d.edge_index, d.edge_attr))
        d = cp.community_pooling(cp.get_preloaded_cluster(d.cluster0, d.batch), d)
        d.x = F.relu(conv2(d.x, d.edge_index, d.edge_attr))
        x, b = cp.max_pool_x(cp.get_preloaded_cluster(d.cluster1, d.batch), d.x, d.batch)
        pred = fc(cp.scatter_mean(x, b, dim=0)).reshape(-1)
        loss = F.mse_loss(pred, batch.y)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert losses[-1] < losses[0]


def test_precluster_on_device_reproduces_the_fixture_labels():
    """Offline MCL (fp64, one workgroup per graph) + pooling on the MI355X == the reference's stored
    clustering/mcl/depth_0 and depth_1 for all 10 fixture graphs."""
    from test_mcl import _fixture_batch_without_clusters
    from deeprank_gnn_amd.clustering import precluster
    batch, expect0, expect1 = _fixture_batch_without_clusters()
    d0, d1 = precluster(batch.to(DEV))
    np.testing.assert_array_equal(d0.cpu().numpy(), expect0)
    np.testing.assert_array_equal(d1.cpu().numpy(), expect1)


def test_layer_constructor_options_vs_reference_golden():
    """Layer options the shipped nets never use (undirected=False, bias=False, GINetConvLayer(bias=True)) on the
    MI355X against the reference-recorded vectors (reference sGAT.py:86-87, :50-53, foutnet.py:43-46, ginet.py:26-37)."""
    from layer_option_check import check_layer_options
    check_layer_options(torch.device("cuda:0"))
