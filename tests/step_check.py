"""Fused training-step launch vs the forward/backward pair of launches on the same inputs."""
import copy

import numpy as np
import torch

from deeprank_gnn_amd.data import Batch
from deeprank_gnn_amd.trainer import FusedTrainer


def ragged_batch(seed, n_feat):
    from test_emu_topology import random_graph
    rng = np.random.default_rng(seed)
    graphs = []
    for k in range(6):
        n = int(rng.integers(2, 50))
        e = int(rng.integers(0, 4 * n))
        g = random_graph(rng, n, e, int(rng.integers(1, n + 1)), int(rng.integers(1, 5)), sym=bool(k % 2),
                         self_loops=(k == 3), dup=(k == 4))
        g.x = torch.from_numpy(rng.standard_normal((n, n_feat)).astype(np.float32))
        graphs.append(g)
    graphs.insert(2, random_graph(rng, 1, 0, 1, 1))              # single node, no edge
    graphs[2].x = torch.from_numpy(rng.standard_normal((1, n_feat)).astype(np.float32))
    return Batch.from_data_list(graphs)


def check_fused_matches_pair(Net, n_feat, task, device, api=None, seed=0, steps=2):
    torch.manual_seed(seed)
    batch = ragged_batch(seed, n_feat)
    n_out = 1 if task == "reg" else 3
    cw = None
    if task == "class":
        batch.y = torch.tensor([k % 3 for k in range(batch.num_graphs)])
        cw = torch.tensor([0.2, 0.5, 0.3])
    else:
        batch.y = torch.arange(batch.num_graphs, dtype=torch.float32) * 0.3 - 1.0
    a = Net(n_feat, n_out, 1)
    if hasattr(a, "dropout"):
        a.dropout = 0.0
    b = copy.deepcopy(a)
    batch = batch.to(device)
    kw = {} if api is None else {"api": api}
    cwd = None if cw is None else cw.to(device)
    ta = FusedTrainer(a.to(device), lr=0.01, task=task, class_weights=cwd, **kw)
    tb = FusedTrainer(b.to(device), lr=0.01, task=task, class_weights=cwd, **kw)
    tb.fused_step = False
    fused_used = False
    for it in range(steps):
        la = ta.train_step(batch)
        lb = tb.train_step(batch)
        from deeprank_gnn_amd.topology import Topology
        from deeprank_gnn_amd import _lib
        # (the workspace train_step builds for itself: hierarchical order + aggregation tiles -- what the device's fused kernels need)
        topo = Topology.from_batch(batch, need_weights=(ta.kind == _lib.SGAT), **kw)
        fused_used = fused_used or ta._can_fuse(topo, n_feat, None, True, batch.x)
        isolated = Net.__name__ == "FoutNet"       # FoutLayer: NaN rows for isolated nodes are dropped by the max-pool
        np.testing.assert_allclose(float(la), float(lb), rtol=2e-5, equal_nan=isolated)
        np.testing.assert_allclose(ta.last_pred.cpu().numpy(), tb.last_pred.cpu().numpy(), rtol=1e-4, atol=1e-5)
        ga, gb = ta.flat_g.cpu().numpy(), tb.flat_g.cpu().numpy()
        np.testing.assert_allclose(ga, gb, rtol=1e-4, atol=1e-5 * max(1.0, float(np.abs(gb).max())),
                                   err_msg="gradient, step %d" % it)
        # Adam normalises every gradient element by its own running magnitude: elements with a gradient at
        # rounding-noise level may move by a visible fraction of lr differently in the two paths
        np.testing.assert_allclose(ta.flat_p.cpu().numpy(), tb.flat_p.cpu().numpy(), rtol=1e-4, atol=2e-5)
    assert int(ta.step) == steps and int(tb.step) == steps
    return fused_used


def check_fused_predict(Net, n_feat, task, device, api=None, seed=0):
    """Inference through the fused kernel: equals the forward/head pair, is repeatable back to back (the exchange
    words carry the same tag in every inference launch) and leaves training untouched when interleaved with it."""
    torch.manual_seed(seed)
    batch = ragged_batch(seed, n_feat)
    n_out = 1 if task == "reg" else 3
    batch.y = (torch.tensor([k % 3 for k in range(batch.num_graphs)]) if task == "class"
               else torch.arange(batch.num_graphs, dtype=torch.float32) * 0.3 - 1.0)
    a = Net(n_feat, n_out, 1)
    b = copy.deepcopy(a)
    batch = batch.to(device)
    kw = {} if api is None else {"api": api}
    ta = FusedTrainer(a.to(device), lr=0.01, task=task, **kw)
    tb = FusedTrainer(b.to(device), lr=0.01, task=task, **kw)
    tb.fused_step = False
    for it in range(2):
        p1 = ta.predict(batch).cpu().numpy()
        p2 = ta.predict(batch).cpu().numpy()
        np.testing.assert_array_equal(p1, p2)
        np.testing.assert_allclose(p1, tb.predict(batch).cpu().numpy(), rtol=1e-4, atol=1e-5)
        la, lb = ta.train_step(batch), tb.train_step(batch)
        np.testing.assert_allclose(float(la), float(lb), rtol=2e-5, equal_nan=(Net.__name__ == "FoutNet"))
    assert int(ta.step) == 2
    # a stream of batches: the next batch's topology is built inside the current inference launch
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd import _lib
    need_w = ta.kind == _lib.SGAT
    t0 = Topology.from_batch(batch, need_weights=need_w, **kw)
    t1 = Topology.from_batch(batch, need_weights=need_w, build=False, **kw)
    pa = ta.predict(batch, topo=t0, next_topo=t1).cpu().numpy()
    pb = ta.predict(batch, topo=t1).cpu().numpy()          # t1 was built by the previous call
    np.testing.assert_array_equal(pa, pb)
    np.testing.assert_allclose(pa, tb.predict(batch).cpu().numpy(), rtol=1e-4, atol=1e-5)


class plan_overrides(object):
    """``with plan_overrides(trainer, force_wgs=1):`` every fused-step launch of THAT trainer inside the block is planned
    with these overrides of drgnn_step_plan (force_wgs / no_class / no_aggregate / no_split / no_paired); other trainers of the
    process are not affected (the overrides travel with the plan, there is no process-wide switch)."""

    def __init__(self, trainer, **ov):
        self.trainer, self.ov = trainer, ov

    def __enter__(self):
        self.saved = dict(self.trainer.plan_overrides)
        self.trainer.plan_overrides = dict(self.saved, **self.ov)
        return self

    def __exit__(self, *exc):
        self.trainer.plan_overrides = self.saved
        return False


def one_workgroup_layout(trainer, paired=True):
    """every GINet fused-step launch of ``trainer`` runs both branches of a graph in ONE workgroup, the layout the library
    takes by itself beyond the resident batch size.  paired: both branches share every phase (the default form of the
    product-first kernel); False: branch after branch (the form graphs too large for the paired LDS plan take)"""
    return plan_overrides(trainer, force_wgs=1, no_paired=0 if paired else 1)


def check_one_workgroup_layout(n_feat, task, device, api=None, seed=0, paired=True):
    """GINet: the one-workgroup-per-graph step (both branches in sequence) against the two-workgroup step on the same
    ragged batch -- loss, predictions, every gradient, two Adam steps, inference."""
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.topology import Topology
    api = api or _lib.get()
    kw = {"api": api}
    torch.manual_seed(seed)
    batch = ragged_batch(seed, n_feat)
    n_out = 1 if task == "reg" else 3
    cw = None
    if task == "class":
        batch.y = torch.tensor([k % 3 for k in range(batch.num_graphs)])
        cw = torch.tensor([0.2, 0.5, 0.3]).to(device)
    else:
        batch.y = torch.arange(batch.num_graphs, dtype=torch.float32) * 0.3 - 1.0
    a = GINet(n_feat, n_out, 1)
    a.dropout = 0.0
    b = copy.deepcopy(a)
    batch = batch.to(device)
    ta = FusedTrainer(a.to(device), lr=0.01, task=task, class_weights=cw, **kw)
    tb = FusedTrainer(b.to(device), lr=0.01, task=task, class_weights=cw, **kw)
    topo = Topology.from_batch(batch, need_weights=False, **kw)
    assert ta._can_fuse(topo, n_feat)
    assert ta._plan_for(topo, n_feat).wgs_per_graph == 2      # a handful of graphs: resident, two workgroups per graph
    with one_workgroup_layout(ta, paired):
        assert ta._plan_for(topo, n_feat).wgs_per_graph == 1
        assert tb._plan_for(topo, n_feat).wgs_per_graph == 2      # (the other trainer of the process keeps its own plan)
    for it in range(2):
        with one_workgroup_layout(ta, paired):
            pa = ta.predict(batch).cpu().numpy()
            la = ta.train_step(batch)
        pb = tb.predict(batch).cpu().numpy()
        lb = tb.train_step(batch)
        np.testing.assert_allclose(pa, pb, rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(float(la), float(lb), rtol=2e-6)
        np.testing.assert_allclose(ta.last_pred.cpu().numpy(), tb.last_pred.cpu().numpy(), rtol=1e-5, atol=1e-6)
        ga, gb = ta.flat_g.cpu().numpy(), tb.flat_g.cpu().numpy()
        np.testing.assert_allclose(ga, gb, rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(gb).max())), err_msg="gradient, step %d" % it)
        np.testing.assert_allclose(ta.flat_p.cpu().numpy(), tb.flat_p.cpu().numpy(), rtol=1e-4, atol=2e-5)
    assert int(ta.step) == 2 and int(tb.step) == 2
