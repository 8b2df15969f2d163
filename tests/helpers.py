"""Shared test helpers: golden loading, fixture batches, parameter dicts."""
import os

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")
NODE_FEATURES = ['type', 'polarity', 'bsa', 'depth', 'hse', 'ic', 'pssm']  # reference tests/test_nn.py:13-14


def golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


def params_of(g, prefix="param/"):
    return {k[len(prefix):]: torch.from_numpy(v.copy()) for k, v in g.items() if k.startswith(prefix)}


def fixture_graphs(node_feature=None, target="irmsd", count=None):
    from deeprank_gnn_amd.dataset import GraphDataSet
    ds = GraphDataSet(os.path.join(GOLDEN, "fixture_1ATN.npz"),
                      node_feature=list(node_feature or NODE_FEATURES), edge_feature=["dist"], target=target)
    n = len(ds) if count is None else count
    return [ds[i] for i in range(n)]


def fixture_batch(count=8, **kw):
    from deeprank_gnn_amd.data import Batch
    return Batch.from_data_list(fixture_graphs(count=count, **kw))


def syn4_graphs():
    import deeprank_gnn_amd.synthetic as synth
    return [synth.make_graph(i, n_nodes=40, n_pairs=70, n_feat=12, n_c1=4, n_internal=40) for i in range(4)]


def syn4_batch():
    from deeprank_gnn_amd.data import Batch
    return Batch.from_data_list(syn4_graphs())


def iso3_batch():
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.data import Batch
    return Batch.from_data_list([synth.make_graph(i, n_nodes=40, n_pairs=70, n_feat=12, n_c1=4, n_internal=40,
                                                  isolate_node=(7 if i == 1 else None)) for i in range(3)])


CASES = {
    # golden file            net       batch factory                task
    "fix8_GINet.npz": ("GINet", lambda: fixture_batch(8), "reg"),
    "fix8_sGAT.npz": ("sGAT", lambda: fixture_batch(8), "reg"),
    "fix8_FoutNet.npz": ("FoutNet", lambda: fixture_batch(8), "reg"),
    "fix8_GINet_class.npz": ("GINet", lambda: fixture_batch(8), "class"),
    "syn4_GINet.npz": ("GINet", syn4_batch, "reg"),
    "syn4_sGAT.npz": ("sGAT", syn4_batch, "reg"),
    "syn4_FoutNet.npz": ("FoutNet", syn4_batch, "reg"),
    # one graph has an isolated node: GINet row 0, sGAT row = bias, FoutLayer row NaN (dropped by the pooling)
    "iso3_GINet.npz": ("GINet", iso3_batch, "reg"),
    "iso3_sGAT.npz": ("sGAT", iso3_batch, "reg"),
    "iso3_FoutNet.npz": ("FoutNet", iso3_batch, "reg"),
}
