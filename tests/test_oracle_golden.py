"""Pins the CPU oracle (oracle/cpu_ref.py) against golden vectors produced by the
reference's own layer code (tests/golden/gen/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from helpers import CASES, golden, params_of, fixture_graphs, fixture_batch, treg_graphs
from oracle import cpu_ref

TOL = dict(rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("fname", sorted(CASES))
def test_forward_loss_and_every_gradient(fname):
    net, make_batch, task = CASES[fname]
    g = golden(fname)
    params = params_of(g)
    batch = make_batch()
    target = torch.from_numpy(g["target"])
    trace = {}
    pred, loss, grads = cpu_ref.loss_and_grads(net, params, batch, target, task=task, trace=trace)
    np.testing.assert_allclose(pred.numpy(), g["out"], **TOL)
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=2e-5)
    for name in params:
        ref = g["grad/" + name]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(grads[name].numpy(), ref, rtol=1e-4, atol=2e-5 * scale, err_msg=name)
    # per-stage intermediates: integer topology exactly, floats to tolerance
    for tag in ("a.", "b.") if net == "GINet" else ("a.",):
        np.testing.assert_array_equal(trace[tag + "pool_edge_index"].numpy(), g[tag + "pool_edge_index"])
        np.testing.assert_array_equal(trace[tag + "pool_batch"].numpy(), g[tag + "pool_batch"])
        np.testing.assert_array_equal(trace[tag + "batch2"].numpy(), g[tag + "batch2"])
        for key in ("z1", "xp", "pool_edge_attr", "z2", "x2"):
            np.testing.assert_allclose(trace[tag + key].detach().numpy(), g[tag + key], err_msg=tag + key, **TOL)
    np.testing.assert_allclose(trace["readout"].detach().numpy(), g["readout"], **TOL)


def test_ginet_attention_is_dead():
    """softmax over a size-1 axis == 1: attention/edge parameters get exactly-zero
    gradients -- both in the golden and in the oracle."""
    g = golden("fix8_GINet.npz")
    dead = [k for k in g if k.startswith("grad/") and ("fc_attention" in k or "fc_edge_attr" in k)]
    assert len(dead) == 8
    for k in dead:
        assert np.abs(g[k]).max() == 0.0


def test_collate_layout_matches_reference():
    g = golden("collate.npz")
    from deeprank_gnn_amd.data import Batch
    b = Batch.from_data_list(fixture_graphs(count=3))
    for key in ("x", "edge_index", "edge_attr", "internal_edge_index", "internal_edge_attr",
                "y", "pos", "cluster0", "cluster1", "batch"):
        got = b[key].numpy()
        assert got.dtype == g[key].dtype, key
        np.testing.assert_array_equal(got, g[key], err_msg=key)
    assert list(b.mol) == [str(m) for m in g["mol"]]
    assert b.num_graphs == 3


def test_conv_layers_with_isolated_node():
    g = golden("layers_isolated.npz")
    import deeprank_gnn_amd.synthetic as synth
    gr = synth.make_graph(7, n_nodes=24, n_pairs=40, n_feat=6, n_c1=3, n_internal=10, isolate_node=5)
    t = lambda k: torch.from_numpy(g[k])
    out = cpu_ref.ginet_conv(gr.x, gr.edge_index, gr.edge_attr, t("ginet.fc"), t("ginet.fc_edge_attr"), t("ginet.fc_attention"))
    np.testing.assert_allclose(out.numpy(), g["ginet.out"], **TOL)
    assert np.all(g["ginet.out"][5] == 0.0)
    out = cpu_ref.sgat_conv(gr.x, gr.edge_index, gr.edge_attr, t("sgat.weight"), t("sgat.bias"))
    np.testing.assert_allclose(out.numpy(), g["sgat.out"], **TOL)
    np.testing.assert_array_equal(g["sgat.out"][5], g["sgat.bias"])      # isolated row == bias
    for looped in (True, False):
        out = cpu_ref.fout_conv(gr.x, gr.edge_index, t("fout.Wc"), t("fout.Wn"), t("fout.bias"), looped=looped)
        assert np.isnan(g["fout.out"][5]).all() and np.isnan(out.numpy()[5]).all()   # NaN parity
        keep = np.arange(24) != 5
        np.testing.assert_allclose(out.numpy()[keep], g["fout.out"][keep], **TOL)


def test_pretrained_classifier_known_answer():
    """Real shipped weights (reference paper_pretrained_models/biological_vs_crystal_interfaces)
    on the 10 fixture graphs: logits recorded from the reference."""
    g = golden("pretrained_class.npz")
    params = params_of(g)
    graphs = fixture_graphs(node_feature=[str(s) for s in g["node_feature"]], target=None)
    from deeprank_gnn_amd.data import Batch
    out = cpu_ref.ginet_forward(params, Batch.from_data_list(graphs))
    np.testing.assert_allclose(out.numpy(), g["logits_batched"], rtol=1e-4, atol=1e-4)
    one = torch.cat([cpu_ref.ginet_forward(params, Batch.from_data_list([gr])) for gr in graphs])
    np.testing.assert_allclose(one.numpy(), g["logits_single"], rtol=1e-4, atol=1e-4)


def test_pretrained_regression_model_known_answer():
    """The reference's shipped regression model (paper_pretrained_models/scoring_of_docking_models/treg_yfnat_b128_*, GINet
    with 48 node features) on six synthetic 48-feature graphs: eval-mode predictions and one training step's loss and
    gradients recorded from the reference's own GINet (tests/golden/gen/make_width_golden.py)."""
    g = golden("pretrained_treg.npz")
    params = params_of(g)
    assert params["conv1.fc.weight"].shape == (16, 48)
    graphs = treg_graphs()
    from deeprank_gnn_amd.data import Batch
    out = cpu_ref.ginet_forward(params, Batch.from_data_list(graphs))
    np.testing.assert_allclose(out.numpy(), g["pred_batched"], rtol=1e-4, atol=1e-4)
    one = torch.cat([cpu_ref.ginet_forward(params, Batch.from_data_list([gr])) for gr in graphs])
    np.testing.assert_allclose(one.numpy(), g["pred_single"], rtol=1e-4, atol=1e-4)
    pred, loss, grads = cpu_ref.loss_and_grads("GINet", params, Batch.from_data_list(graphs), torch.from_numpy(g["target"]))
    np.testing.assert_allclose(pred.numpy(), g["out"], **TOL)
    np.testing.assert_allclose(loss.numpy(), g["loss"], rtol=2e-5)
    for name in params:
        ref = g["grad/" + name]
        np.testing.assert_allclose(grads[name].numpy(), ref, rtol=1e-4, atol=2e-5 * max(1.0, float(np.abs(ref).max())),
                                   err_msg=name)


def test_toy_pooling():
    """6-node graph of reference tests/test_community_pooling.py:12-19, two copies."""
    import types
    g = golden("toy6.npz")
    ei = torch.tensor([[0, 1, 1, 2, 3, 4, 4, 5], [1, 0, 2, 1, 4, 3, 5, 4]])
    x = torch.arange(6, dtype=torch.float).view(6, 1)
    two = types.SimpleNamespace(
        x=torch.cat([x, x]), edge_index=torch.cat([ei, ei + 6], dim=1), edge_attr=torch.ones(16, 1),
        batch=torch.tensor([0] * 6 + [1] * 6), pos=torch.arange(18, dtype=torch.float).view(6, 3).repeat(2, 1))
    p = cpu_ref.community_pooling(torch.from_numpy(g["cluster"]), two)
    np.testing.assert_array_equal(p.x.numpy(), g["x"])
    assert p.edge_index.numel() == 0 and g["edge_index"].size == 0
    np.testing.assert_array_equal(p.batch.numpy(), g["batch"])
    np.testing.assert_allclose(p.pos.numpy(), g["pos"])
    g = golden("toy6_edges.npz")
    two.edge_attr = torch.from_numpy(g["in_edge_attr"])
    p = cpu_ref.community_pooling(torch.from_numpy(g["cluster"]), two)
    np.testing.assert_array_equal(p.edge_index.numpy(), g["edge_index"])
    np.testing.assert_array_equal(p.edge_attr.numpy(), g["edge_attr"])
    np.testing.assert_array_equal(p.x.numpy(), g["x"])


def test_fixture_clustering_pins_pool_edge_indices():
    """The fixture's stored clustering/mcl/depth_1 has exactly as many entries as the number
    of depth_0 clusters, for every graph (output of the reference's real PreCluster)."""
    for gr in fixture_graphs(count=None):
        assert gr.cluster1.numel() == int(torch.unique(gr.cluster0).numel())
