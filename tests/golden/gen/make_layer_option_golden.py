"""Golden vectors for the constructor options of the reference's conv layers that the shipped nets never use but the
README's custom-net recipe exposes, by RUNNING THE REFERENCE's unmodified layer classes against
the third-party stand-ins of pyg_shims.py.  Build-container only:  python tests/golden/gen/make_layer_option_golden.py

  sGraphAttentionLayer(undirected=False)   second scatter_mean(alpha, col, out=out)  (reference sGAT.py:86-87)
  sGraphAttentionLayer(bias=False), FoutLayer(bias=False)                             (sGAT.py:50-53, foutnet.py:43-46)
  GINetConvLayer(bias=True)                                                            (ginet.py:26-37)
Graph: directed, asymmetric (both scatters of the undirected=False case differ), one node without outgoing edges.
Stored per case: parameters, output, and -- for the loss  sum(out * G)  with a fixed G -- every parameter gradient and
the input gradient."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", "..", ".."))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")

import numpy as np
import torch

import pyg_shims

pyg_shims.install()
from deeprank_gnn.ginet import GINetConvLayer               # noqa: E402  (reference)
from deeprank_gnn.sGAT import sGraphAttentionLayer          # noqa: E402
from deeprank_gnn.foutnet import FoutLayer                  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "layer_options.npz")
rng = np.random.default_rng(42)
N, F, H = 23, 6, 10
pairs = set()
while len(pairs) < 60:
    i, j = int(rng.integers(0, N - 1)), int(rng.integers(0, N))      # node N-1 never appears as a row
    if i != j:
        pairs.add((i, j))
pairs = sorted(pairs, key=lambda p: rng.random())
edge_index = torch.tensor(np.array(pairs, dtype=np.int64).T.copy())
# FoutLayer returns NaN for a node without outgoing edges: give its variant a graph where every node has one
pairs_f = list(pairs) + [(N - 1, 3)]
edge_index_f = torch.tensor(np.array(pairs_f, dtype=np.int64).T.copy())
edge_attr = torch.tensor(rng.uniform(0.2, 1.8, size=(len(pairs), 1)).astype(np.float32))
x0 = torch.tensor(rng.normal(size=(N, F)).astype(np.float32))
G = torch.tensor(rng.normal(size=(N, H)).astype(np.float32))
store = {"x": x0.numpy(), "edge_index": edge_index.numpy(), "edge_index_fout": edge_index_f.numpy(),
         "edge_attr": edge_attr.numpy(), "G": G.numpy()}


def run(tag, layer, call):
    x = x0.clone().requires_grad_(True)
    out = call(layer, x)
    (out * G).sum().backward()
    store[tag + "/out"] = out.detach().numpy()
    store[tag + "/grad_x"] = x.grad.numpy()
    for name, p in layer.named_parameters():
        store[tag + "/param/" + name] = p.detach().numpy()
        store[tag + "/grad/" + name] = (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
    print(tag, "out[0,:3] =", store[tag + "/out"][0, :3], " params:", [n for n, _ in layer.named_parameters()])


torch.manual_seed(7)
run("sgat_directed", sGraphAttentionLayer(F, H, undirected=False), lambda l, x: l(x, edge_index, edge_attr))
run("sgat_directed_nobias", sGraphAttentionLayer(F, H, bias=False, undirected=False), lambda l, x: l(x, edge_index, edge_attr))
run("sgat_nobias", sGraphAttentionLayer(F, H, bias=False), lambda l, x: l(x, edge_index, edge_attr))
run("fout_nobias", FoutLayer(F, H, bias=False), lambda l, x: l(x, edge_index_f))
run("ginet_bias", GINetConvLayer(F, H, 1, bias=True), lambda l, x: l(x, edge_index, edge_attr))
np.savez_compressed(OUT, **store)
print("wrote", OUT)
