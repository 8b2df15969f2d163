"""Constructor options of the conv layers (reference sGAT.py:35-60,86-87, foutnet.py:30-48, ginet.py:22-40) against the
golden vectors recorded from the reference's own layer classes (tests/golden/gen/make_layer_option_golden.py) and the
oracle.  Shared by the emulated and the MI355X test."""
import numpy as np
import torch

from helpers import golden
from oracle import cpu_ref


def _load(layer, g, tag):
    sd = {k[len(tag) + 7:]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(tag + "/param/")}
    layer.load_state_dict(sd, strict=True)
    return layer


def check_layer_options(device, patch_api=None):
    from deeprank_gnn_amd.ginet import GINetConvLayer
    from deeprank_gnn_amd.sGAT import sGraphAttentionLayer
    from deeprank_gnn_amd.foutnet import FoutLayer
    g = golden("layer_options.npz")
    x0 = torch.from_numpy(g["x"])
    ei, ei_f = torch.from_numpy(g["edge_index"]), torch.from_numpy(g["edge_index_fout"])
    ea, G = torch.from_numpy(g["edge_attr"]), torch.from_numpy(g["G"])
    N, F = x0.shape
    H = G.shape[1]
    cases = [
        ("sgat_directed", sGraphAttentionLayer(F, H, undirected=False), lambda l, x: l(x, ei.to(device), ea.to(device))),
        ("sgat_directed_nobias", sGraphAttentionLayer(F, H, bias=False, undirected=False), lambda l, x: l(x, ei.to(device), ea.to(device))),
        ("sgat_nobias", sGraphAttentionLayer(F, H, bias=False), lambda l, x: l(x, ei.to(device), ea.to(device))),
        ("fout_nobias", FoutLayer(F, H, bias=False), lambda l, x: l(x, ei_f.to(device))),
        ("ginet_bias", GINetConvLayer(F, H, 1, bias=True), lambda l, x: l(x, ei.to(device), ea.to(device))),
    ]
    for tag, layer, call in cases:
        layer = _load(layer, g, tag).to(device)
        assert sorted(n for n, _ in layer.named_parameters()) == sorted(k[len(tag) + 7:] for k in g if k.startswith(tag + "/param/"))
        x = x0.clone().to(device).requires_grad_(True)
        out = call(layer, x)
        (out * G.to(device)).sum().backward()
        np.testing.assert_allclose(out.detach().cpu().numpy(), g[tag + "/out"], rtol=1e-4, atol=1e-5, err_msg=tag)
        np.testing.assert_allclose(x.grad.cpu().numpy(), g[tag + "/grad_x"], rtol=1e-4, atol=1e-5, err_msg=tag + " grad_x")
        for name, p in layer.named_parameters():
            ref = g[tag + "/grad/" + name]
            got = p.grad.cpu().numpy() if p.grad is not None else np.zeros_like(ref)
            np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-5, err_msg=tag + " grad " + name)


def check_oracle_options():
    """the oracle restatement of the same options against the same goldens"""
    g = golden("layer_options.npz")
    x = torch.from_numpy(g["x"])
    ei, ea = torch.from_numpy(g["edge_index"]), torch.from_numpy(g["edge_attr"])
    p = lambda tag, n: torch.from_numpy(g["%s/param/%s" % (tag, n)])
    out = cpu_ref.sgat_conv(x, ei, ea, p("sgat_directed", "weight"), p("sgat_directed", "bias"), undirected=False)
    np.testing.assert_allclose(out.numpy(), g["sgat_directed/out"], rtol=1e-5, atol=1e-6)
    out = cpu_ref.sgat_conv(x, ei, ea, p("sgat_directed_nobias", "weight"), None, undirected=False)
    np.testing.assert_allclose(out.numpy(), g["sgat_directed_nobias/out"], rtol=1e-5, atol=1e-6)
    out = cpu_ref.sgat_conv(x, ei, ea, p("sgat_nobias", "weight"), None)
    np.testing.assert_allclose(out.numpy(), g["sgat_nobias/out"], rtol=1e-5, atol=1e-6)
    out = cpu_ref.fout_conv(x, torch.from_numpy(g["edge_index_fout"]), p("fout_nobias", "Wc"), p("fout_nobias", "Wn"), None)
    np.testing.assert_allclose(out.numpy(), g["fout_nobias/out"], rtol=1e-5, atol=1e-6)
    t = "ginet_bias"
    out = cpu_ref.ginet_conv(x, ei, ea, p(t, "fc.weight"), p(t, "fc_edge_attr.weight"), p(t, "fc_attention.weight"),
                             p(t, "fc.bias"), p(t, "fc_edge_attr.bias"), p(t, "fc_attention.bias"))
    np.testing.assert_allclose(out.numpy(), g[t + "/out"], rtol=1e-5, atol=1e-6)
