"""Fused training-step kernel (host-emulation build) on ragged batches, odd feature widths and both
tasks, against the forward/backward pair of kernels.  CPU only."""
import pytest

from emu_api import emu
from step_check import check_fused_matches_pair
from deeprank_gnn_amd.ginet import GINet
from deeprank_gnn_amd.sGAT import sGAT
from deeprank_gnn_amd.foutnet import FoutNet


@pytest.mark.parametrize("Net", [GINet, sGAT, FoutNet])
@pytest.mark.parametrize("n_feat,task", [(5, "reg"), (16, "class"), (40, "reg")])
def test_fused_step_matches_launch_pair(Net, n_feat, task):
    assert check_fused_matches_pair(Net, n_feat, task, "cpu", api=emu(), seed=n_feat)


@pytest.mark.parametrize("Net,n_feat,task", [(GINet, 32, "reg"), (GINet, 5, "class"), (sGAT, 16, "reg"),
                                            (FoutNet, 40, "reg")])
def test_fused_inference(Net, n_feat, task):
    from step_check import check_fused_predict
    check_fused_predict(Net, n_feat, task, "cpu", api=emu(), seed=7 + n_feat)


@pytest.mark.parametrize("paired", [True, False])
@pytest.mark.parametrize("n_feat,task", [(32, "reg"), (5, "class"), (40, "reg"), (16, "reg")])
def test_ginet_one_workgroup_layout_matches_two_workgroup_layout(n_feat, task, paired):
    from step_check import check_one_workgroup_layout
    check_one_workgroup_layout(n_feat, task, "cpu", api=emu(), seed=3 + n_feat, paired=paired)


@pytest.mark.parametrize("Net", [GINet, sGAT, FoutNet])
def test_step_gradients_entry_point(Net):
    """drgnn_step_gradients (include/drgnn.h: the slab sum of the autograd boundary) in the host emulation: unweighted it is the
    gradient half of drgnn_step_update; with per-graph weights it is the weighted sum of the slabs (restated with numpy);
    the ranges handed over for clearing are cleared and the step index is committed."""
    import ctypes
    import numpy as np
    import torch
    from step_check import ragged_batch
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd.trainer import FusedTrainer
    api = emu()
    torch.manual_seed(2)
    batch = ragged_batch(4, 12)
    batch.y = torch.arange(batch.num_graphs, dtype=torch.float32) * 0.5 - 1.0
    net = Net(12, 1, 1)
    if hasattr(net, "dropout"):
        net.dropout = 0.0
    tr = FusedTrainer(net, lr=0.01, task="reg", api=api)
    topo = Topology.from_batch(batch, api=api, need_weights=(tr.kind == _lib.SGAT))
    c = tr._fused_prepare(batch, topo)
    assert c["plan"].family != _lib.STEP_FAMILY_NONE
    tr._fused_launch_step(c)
    tr._fused_launch_update(c, apply_adam=False)
    want = tr.flat_g.clone()
    B = c["B"]
    total = tr.flat_g.numel()
    out = torch.full((total,), 7.0)
    # the gradient descriptors of the trainer point into its own flat buffer: rebuild them over `out`
    from deeprank_gnn_amd.functional import _fill_grads, _split, H1, H2
    views, off = [], 0
    for p in net.parameters():
        views.append(out[off:off + p.numel()].view(p.shape))
        off += p.numel()
    index = {id(p): i for i, p in enumerate(net.parameters())}
    live = tuple(views[index[id(p)]] for p in tr.live)
    g1 = (_lib.ConvGrads * _lib.MAX_BRANCH)()
    g2 = (_lib.ConvGrads * _lib.MAX_BRANCH)()
    for b, (l1, l2) in enumerate(_split(tr.kind, live, tr.n_branch)):
        _fill_grads(g1[b], tr.kind, l1, 12, H1)
        _fill_grads(g2[b], tr.kind, l2, H1, H2)
    live_ids = {id(p) for p in tr.live} | {id(net.fc1.weight), id(net.fc1.bias), id(net.fc2.weight), id(net.fc2.bias)}
    dead = [(sum(q.numel() for q in list(net.parameters())[:i]), p.numel()) for i, p in enumerate(net.parameters()) if id(p) not in live_ids]
    zp = (ctypes.c_void_p * _lib.ZERO_RANGES)()
    zl = (ctypes.c_int64 * _lib.ZERO_RANGES)()
    for i, (o, n) in enumerate(dead):
        zp[i], zl[i] = out.data_ptr() + 4 * o, n
    head_grad = out.data_ptr() + 4 * tr.head_grad_offset
    tr.step2[0], tr.step2[1] = 3, 9
    api.step_gradients(c["desc"], c["partials"], B, g1, g2, c["hp"], c["readout"], tr.R, tr.H, tr.O, head_grad, None, zp, zl,
                       len(dead), tr.step2, c["slabs"], None)
    np.testing.assert_array_equal(out.numpy(), want.numpy())
    assert int(tr.step2[0]) == 9                                  # committed
    # weighted: every graph's slab times w[g] -- the contraction model(batch) / loss.backward() relies on
    w = torch.linspace(0.5, 2.0, B)
    api.step_gradients(c["desc"], c["partials"], B, g1, g2, c["hp"], c["readout"], tr.R, tr.H, tr.O, head_grad, w, zp, zl,
                       len(dead), None, c["slabs"], None)
    got = out.clone()
    # the same through B unweighted calls on one-hot weights: linearity in the weights
    acc = torch.zeros(total)
    for g in range(B):
        e = torch.zeros(B)
        e[g] = 1.0
        api.step_gradients(c["desc"], c["partials"], B, g1, g2, c["hp"], c["readout"], tr.R, tr.H, tr.O, head_grad, e, zp, zl,
                           len(dead), None, c["slabs"], None)
        acc += float(w[g]) * out
    np.testing.assert_allclose(got.numpy(), acc.numpy(), rtol=1e-5, atol=1e-6)
    assert float(np.abs(got.numpy()).sum()) > 0.0
