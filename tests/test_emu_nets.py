"""Fused network body (host-emulation build of the HIP source) + the Python model classes
vs golden vectors recorded from the reference.  CPU only.  Tolerance: 1e-4 (fp32), the bar
BASELINE.json's north_star states."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import CASES, golden, params_of, fixture_graphs
from emu_api import emu
from elementwise import Lazy64, check, check_step, new_stats, assert_arbiter_rate
from deeprank_gnn_amd.topology import Topology
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet
from deeprank_gnn_amd.data import Batch

NETS = {"GINet": GINet, "sGAT": sGAT, "FoutNet": FoutNet}
TOL = 1e-4


def build(net_name, g, n_out):
    params = params_of(g)
    n_feat = {"GINet": lambda: params["conv1.fc.weight"].shape[1],
              "sGAT": lambda: params["conv1.weight"].shape[0] // 2,
              "FoutNet": lambda: params["conv1.Wc"].shape[0]}[net_name]()
    net = NETS[net_name](n_feat, n_out, 1)
    missing = net.load_state_dict(params, strict=True)
    return net


def close(got, ref, name=""):
    scale = max(1.0, float(np.abs(ref).max()))
    np.testing.assert_allclose(got, ref, rtol=TOL, atol=TOL * scale, err_msg=name)


@pytest.mark.parametrize("fname", sorted(CASES))
@pytest.mark.parametrize("global_scratch", [False, True])
def test_net_forward_backward_vs_reference_golden(fname, global_scratch):
    net_name, make_batch, task = CASES[fname]
    g = golden(fname)
    batch = make_batch()
    net = build(net_name, g, g["out"].shape[1])
    if hasattr(net, "dropout"):
        net.dropout = 0.0
    net.train()
    if global_scratch:
        batch.__dict__["_max_nodes"] = 0       # unknown bounds -> kernels run from global scratch
        batch.__dict__["_max_edges"] = 0
    topo = Topology.from_batch(batch, api=emu())
    if global_scratch:
        topo.max_nodes = 0
    assert topo.status()[0] == 0
    target = torch.from_numpy(g["target"])
    lazy = Lazy64(net_name, params_of(g), make_batch(), target=target, task=task, want_trace=True)
    stats = new_stats()
    readout = net.body(batch, topo)
    check(fname + " readout", readout.detach().numpy(), g["readout"], lambda: lazy.traced("readout"), stats)
    out = net(batch, topo=topo)
    loss = F.mse_loss(out.reshape(-1), target) if task == "reg" else F.cross_entropy(out, target)
    loss.backward()
    grads = {}
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        grads[name] = p.grad.numpy()
    check_step(fname, lazy, loss.item(), out.detach().numpy(), grads, g["loss"], g["out"],
               {name: g["grad/" + name] for name in grads}, stats)
    assert_arbiter_rate(stats, fname)


def test_grad_x_matches_oracle():
    """d(loss)/dx is not needed by the reference trainer (x is data) but autograd users may
    ask for it: check against the oracle's autograd."""
    from oracle import cpu_ref
    from helpers import syn4_batch
    for net_name in NETS:
        g = golden("syn4_%s.npz" % net_name)
        batch = syn4_batch()
        net = build(net_name, g, 1)
        if hasattr(net, "dropout"):
            net.dropout = 0.0
        topo = Topology.from_batch(batch, api=emu())
        x = batch.x.clone().requires_grad_(True)
        batch.x = x
        out = net(batch, topo=topo)
        out.sum().backward()
        ref_batch = syn4_batch()
        xr = ref_batch.x.clone().requires_grad_(True)
        ref_batch.x = xr
        kw = {"looped": False} if net_name == "FoutNet" else {}
        cpu_ref.FORWARD[net_name](params_of(g), ref_batch, **kw).sum().backward()
        close(x.grad.numpy(), xr.grad.numpy(), net_name + " grad_x")


def test_pretrained_classifier_known_answer():
    g = golden("pretrained_class.npz")
    graphs = fixture_graphs(node_feature=[str(s) for s in g["node_feature"]], target=None)
    net = GINet(20, 2, 1)
    net.load_state_dict(params_of(g), strict=True)      # shipped checkpoint: names/shapes must match
    net.eval()
    batch = Batch.from_data_list(graphs)
    out = net(batch, topo=Topology.from_batch(batch, api=emu()))
    np.testing.assert_allclose(out.detach().numpy(), g["logits_batched"], rtol=1e-4, atol=1e-4)
    for i in (0, 7):
        b1 = Batch.from_data_list([graphs[i]])
        o1 = net(b1, topo=Topology.from_batch(b1, api=emu()))
        np.testing.assert_allclose(o1.detach().numpy(), g["logits_single"][i:i + 1], rtol=1e-4, atol=1e-4)


def test_state_dict_contract():
    """Parameter names / shapes of the reference classes."""
    sd = GINet(32, 1, 1).state_dict()
    assert len(sd) == 16 and sum(v.numel() for v in sd.values()) == 10697
    assert sd["conv1_ext.fc_attention.weight"].shape == (1, 33)
    assert sum(v.numel() for v in sGAT(32).state_dict().values()) == 4273
    assert sum(v.numel() for v in FoutNet(32).state_dict().values()) == 4273
    assert GINet(4).dropout == 0.4 and GINet(4).clustering == 'mcl'


def test_product_path_refuses_cpu_tensors():
    from helpers import syn4_batch
    from deeprank_gnn_amd import _lib
    net = GINet(12, 1, 1)
    with pytest.raises(_lib.DrgnnError):
        net(syn4_batch())
