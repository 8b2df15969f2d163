"""The drop-in boundary on the fused step kernels: what an UNCHANGED reference trainer runs per mini-batch
(reference NeuralNet.py:489-506)

    optimizer.zero_grad(); pred = model(batch); loss = loss_fn(pred, y); loss.backward(); optimizer.step()

with ``model`` one of this package's GINet / sGAT / FoutNet (deeprank-gnn_amd/fused_autograd.py): ``model(batch)`` and
``loss.backward()`` are one launch each of the aggregation-first family (csrc/drgnn_step2.h / drgnn_step3.h), asserted here
through the engine's plan; the results are compared ELEMENT-WISE with the goldens recorded from the reference's own
ginet.py / sGAT.py / foutnet.py (tests/golden/gen/make_golden.py, make_width_golden.py), tolerance tests/elementwise.py
(1e-4 + 1e-4 |ref| per element, float64 arbiter <= 0.1 %), and with the native trainer's gradients where dropout is on.
"""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from helpers import CASES, STEP_ONLY_CASES, golden, params_of
from elementwise import Lazy64, check_step, new_stats, assert_arbiter_rate

pytestmark = pytest.mark.gpu
ALL_CASES = dict(CASES, **STEP_ONLY_CASES)


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _build(net_name, params, n_out):
    from test_gpu_parity import build
    return build(net_name, params, n_out)      # load_state_dict(strict=True), dropout 0, on the device


def _fw(net_name):
    return {"looped": False} if net_name == "FoutNet" else {}


def _loss(task, out, target):
    return F.mse_loss(out.reshape(-1), target) if task == "reg" else F.cross_entropy(out, target)


def _engine(net):
    from deeprank_gnn_amd.fused_autograd import engine_for
    return engine_for(net)


@pytest.mark.parametrize("fname", sorted(ALL_CASES))
def test_model_loss_backward_vs_reference_golden(fname):
    """model(batch) -> loss -> backward with nothing but the batch handed over: the engine builds the lean + tiles workspace,
    takes the aggregation-first family (asserted), and every gradient matches the reference's."""
    from deeprank_gnn_amd import _lib
    net_name, make_batch, task = ALL_CASES[fname]
    g = golden(fname)
    params = params_of(g)
    n_out = g["out"].shape[1]
    target_cpu = torch.from_numpy(g["target"])
    lazy = Lazy64(net_name, params, make_batch(), target=target_cpu, task=task, **_fw(net_name))
    batch = make_batch().to(_dev())
    net = _build(net_name, params, n_out)
    net.train()
    out = net(batch)
    eng = _engine(net)
    assert eng.last_path == ("jacobian" if n_out == 1 else "two-launch"), eng.last_path
    assert eng.last_plan.family == _lib.STEP_FAMILY_AGGREGATE
    n_feat = int(batch.x.shape[1])
    assert eng.last_plan.width == ((n_feat + 15) // 16) * 16
    topo = eng.topology_for(batch)
    assert topo.flags & _lib.TOPO_LEAN and topo.flags & _lib.TOPO_TILES      # the builder's short chains, tiles formed with it
    loss = _loss(task, out, target_cpu.to(_dev()))
    loss.backward()
    torch.cuda.synchronize()
    grads = {}
    base = None
    for name, p in net.named_parameters():
        assert p.grad is not None, name
        grads[name] = p.grad.cpu().numpy()
        # adopted, not copied: every .grad is a view of the ONE flat buffer the backward launch wrote
        base = p.grad.data_ptr() - 4 * eng.offset[name] if base is None else base
        assert p.grad.data_ptr() == base + 4 * eng.offset[name], name
    stats = new_stats()
    check_step(fname + " [drop-in]", lazy, loss.item(), out.detach().cpu().numpy(), grads, float(g["loss"]), g["out"],
               {name: g["grad/" + name] for name in grads}, stats)
    assert_arbiter_rate(stats, fname)
    # the same batch again: the kept workspace (no builder launch), the same numbers bit for bit
    for p in net.parameters():
        p.grad = None
    out2 = net(batch)
    assert eng.topology_for(batch) is topo
    _loss(task, out2, target_cpu.to(_dev())).backward()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out2.detach().cpu().numpy(), out.detach().cpu().numpy())
    for name, p in net.named_parameters():
        np.testing.assert_array_equal(p.grad.cpu().numpy(), grads[name], err_msg=name)
    # inference launch (eval, no_grad): the training launch's predictions (dropout 0)
    net.eval()
    with torch.no_grad():
        pred = net(batch)
    assert eng.last_path == "inference" and eng.last_plan.family == _lib.STEP_FAMILY_AGGREGATE
    np.testing.assert_allclose(pred.cpu().numpy(), out.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


def _legacy_grads(net, batch, loss_fn):
    """the launch pair (functional.net_body) + the head in torch: plain autograd through every stage"""
    for p in net.parameters():
        p.grad = None
    x = net.body(batch)
    x = F.relu(net.fc1(x))
    out = net.fc2(x)
    loss_fn(out).backward()
    return out.detach(), {n: p.grad.clone() for n, p in net.named_parameters()}


@pytest.mark.parametrize("net_name", ["GINet", "sGAT", "FoutNet"])
def test_any_loss_through_the_slabs(net_name):
    """The one-launch backward contracts d pred_g / d theta with WHATEVER d loss / d pred autograd hands over: losses the
    library has never heard of, against plain autograd through the launch pair + torch head on the same net."""
    import deeprank_gnn_amd.synthetic as synth
    batch = synth.make_batch(0, 16).to(_dev())
    torch.manual_seed(3)
    from test_gpu_parity import nets
    net = nets()[net_name](32, 1, 1).to(_dev())
    if hasattr(net, "dropout"):
        net.dropout = 0.0
    net.train()
    y = batch.y
    losses = {
        "l1": lambda o: F.l1_loss(o.reshape(-1), y),
        "huber": lambda o: F.smooth_l1_loss(o.reshape(-1), y, beta=0.5),
        "weighted_cubic": lambda o: ((o.reshape(-1) - y) ** 3 * torch.linspace(0.1, 2.0, o.shape[0], device=o.device)).sum(),
        "sum": lambda o: o.sum(),                       # (an expanded, non-contiguous upstream gradient)
    }
    for name, fn in losses.items():
        ref_out, ref = _legacy_grads(net, batch, fn)
        for p in net.parameters():
            p.grad = None
        out = net(batch)
        assert _engine(net).last_path == "jacobian"
        fn(out).backward()
        torch.cuda.synchronize()
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out.cpu().numpy(), rtol=1e-5, atol=1e-5)
        for n, p in net.named_parameters():
            r = ref[n].cpu().numpy()
            np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(r).max())),
                                       err_msg="%s %s %s" % (net_name, name, n))


@pytest.mark.parametrize("task,n_out", [("reg", 1), ("class", 2)])
def test_dropout_on_equals_the_native_step(task, n_out):
    """GINet with dropout 0.4 (ginet.py:138): the drop-in step draws the mask of the native step (same seed, same step index),
    so pred and every gradient equal FusedTrainer.compute_gradients' -- one output through the slabs, two outputs through the
    forward-only launch + the training launch fed with d loss / d pred."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.ginet import GINet
    from deeprank_gnn_amd.trainer import FusedTrainer
    from deeprank_gnn_amd.topology import Topology
    batch = synth.make_batch(0, 24).to(_dev())
    if task == "class":
        batch.y = (batch.y > 10).to(torch.int64)
    torch.manual_seed(11)
    net = GINet(32, n_out, 1).to(_dev())
    twin = copy.deepcopy(net)
    net.train()
    out = net(batch)
    eng = _engine(net)
    assert eng.last_path == ("jacobian" if n_out == 1 else "two-launch")
    _loss(task, out, batch.y).backward()
    tr = FusedTrainer(twin, lr=0.01, task=task, seed=eng.seed)
    tr.compute_gradients(batch, topo=Topology.from_batch(batch, need_weights=False))
    torch.cuda.synchronize()
    assert float((out.detach() - tr.last_pred).abs().max()) <= 1e-6
    # (a dropped unit is an exact zero in both: the masks agree or the predictions could not)
    for (n, p), (_, q) in zip(net.named_parameters(), twin.named_parameters()):
        r = q.grad.cpu().numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=1e-4, atol=1e-6 * max(1.0, float(np.abs(r).max())), err_msg=n)
    # the step index moved on with the backward: the next forward draws another mask
    with torch.no_grad():
        again = net(batch)
    assert float((again - out.detach()).abs().max()) > 1e-4


def test_two_forwards_before_a_backward_and_accumulation():
    """(l(a) + l(b)).backward() with both forwards in flight (the second keeps private slabs), and autograd's accumulation
    into existing .grad (zero_grad(set_to_none=False)): never a buffer added to itself."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.sGAT import sGAT
    a = synth.make_batch(0, 8).to(_dev())
    b = synth.make_batch(8, 8).to(_dev())
    torch.manual_seed(5)
    net = sGAT(32, 1, 1).to(_dev())
    net.train()

    def grads_of(fn):
        for p in net.parameters():
            p.grad = None
        fn()
        torch.cuda.synchronize()
        return {n: p.grad.clone() for n, p in net.named_parameters()}
    ga = grads_of(lambda: F.mse_loss(net(a).reshape(-1), a.y).backward())
    gb = grads_of(lambda: F.mse_loss(net(b).reshape(-1), b.y).backward())
    both = grads_of(lambda: (F.mse_loss(net(a).reshape(-1), a.y) + F.mse_loss(net(b).reshape(-1), b.y)).backward())
    for n in ga:
        np.testing.assert_allclose(both[n].cpu().numpy(), (ga[n] + gb[n]).cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=n)
    # accumulation: zeroed in place, then two backward passes
    opt = torch.optim.SGD(net.parameters(), lr=0.0)
    opt.zero_grad(set_to_none=False)
    F.mse_loss(net(a).reshape(-1), a.y).backward()
    F.mse_loss(net(b).reshape(-1), b.y).backward()
    torch.cuda.synchronize()
    for n, p in net.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), (ga[n] + gb[n]).cpu().numpy(), rtol=1e-5, atol=1e-6, err_msg=n)


def test_outside_the_fused_kernels_the_launch_pair_runs():
    """x.requires_grad (d loss / d x is the launch pair's), and a copy of the net starts with its own engine."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.foutnet import FoutNet
    batch = synth.make_batch(0, 4).to(_dev())
    torch.manual_seed(7)
    net = FoutNet(32, 1, 1).to(_dev())
    net.train()
    out = net(batch)
    assert _engine(net).last_path == "jacobian"
    twin = copy.deepcopy(net)
    assert twin.__dict__.get("_drgnn_engine") is None
    batch2 = batch.clone()
    batch2.x.requires_grad_(True)
    out2 = twin(batch2)
    assert _engine(twin).last_path is None
    out2.sum().backward()
    assert batch2.x.grad is not None and bool(torch.isfinite(batch2.x.grad).all())
    np.testing.assert_allclose(out2.detach().cpu().numpy(), out.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("net_name", ["GINet", "sGAT", "FoutNet"])
def test_reference_epoch_loop_eager_and_captured(net_name):
    """NeuralNet._epoch's loop body with torch.optim.Adam over model.parameters() (NeuralNet.py:183-184,489-506), five steps
    eager against five steps of the native trainer (its Adam follows torch's arithmetic), then the same body recorded in
    a hipGraph and replayed: the parameters after the replays equal the eager run's."""
    import deeprank_gnn_amd.synthetic as synth
    from test_gpu_parity import nets
    from deeprank_gnn_amd.trainer import FusedTrainer
    batches = [synth.make_batch(16 * i, 16).to(_dev()) for i in range(5)]
    torch.manual_seed(13)
    net = nets()[net_name](32, 1, 1).to(_dev())
    if hasattr(net, "dropout"):
        net.dropout = 0.0
    start = copy.deepcopy(net.state_dict())
    net.train()
    opt = torch.optim.Adam(net.parameters(), lr=0.01)
    for b in batches:
        opt.zero_grad()
        pred = net(b)
        assert _engine(net).last_path == "jacobian"
        loss = F.mse_loss(pred.reshape(-1), b.y)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    eager = {k: v.clone() for k, v in net.state_dict().items()}
    native = nets()[net_name](32, 1, 1).to(_dev())
    native.load_state_dict(start)
    if hasattr(native, "dropout"):
        native.dropout = 0.0
    tr = FusedTrainer(native, lr=0.01, task="reg")
    for b in batches:
        tr.train_step(b)
    torch.cuda.synchronize()
    for k, v in native.state_dict().items():
        np.testing.assert_allclose(eager[k].cpu().numpy(), v.cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg=k)
    # recorded: one graph per batch, the optimiser capturable
    cap = nets()[net_name](32, 1, 1).to(_dev())
    cap.load_state_dict(start)
    if hasattr(cap, "dropout"):
        cap.dropout = 0.0
    cap.train()
    copt = torch.optim.Adam(cap.parameters(), lr=0.01, capturable=True)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):      # warm-up on a side stream (optimiser state, engine buffers, workspaces)
        for b in batches:
            copt.zero_grad(set_to_none=True)
            F.mse_loss(cap(b).reshape(-1), b.y).backward()
            copt.step()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    cap.load_state_dict(start)
    copt = torch.optim.Adam(cap.parameters(), lr=0.01, capturable=True)
    with torch.cuda.stream(side):
        copt.zero_grad(set_to_none=True)
        F.mse_loss(cap(batches[0]).reshape(-1), batches[0].y).backward()
        copt.step()                     # (state initialised outside the capture)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    cap.load_state_dict(start)
    for st in copt.state.values():
        st["step"].zero_(); st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
    graphs = []
    for b in batches:
        gr = torch.cuda.CUDAGraph()
        copt.zero_grad(set_to_none=True)
        with torch.cuda.graph(gr):
            F.mse_loss(cap(b).reshape(-1), b.y).backward()
            copt.step()
        graphs.append(gr)
    cap.load_state_dict(start)
    for st in copt.state.values():
        st["step"].zero_(); st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
    for gr in graphs:
        gr.replay()
    torch.cuda.synchronize()
    # (capturable Adam forms its bias corrections on the device in float32, the eager one on the host: Adam's g / sqrt(v)
    # turns a last-bit difference of a tiny gradient into 1e-4 of an update; lr 0.01, five steps)
    for k, v in cap.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), eager[k].cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("net_name", ["sGAT", "FoutNet"])
def test_feature_count_not_a_multiple_of_four_on_an_odd_node_total(net_name):
    """7 features (rows padded to 8 floats in the tiles' X copy) on 5 x 45 = 225 nodes: the X copy starts at a multiple of 4 floats
    whatever the parity of the node total, so the step kernels' 128-bit row loads are aligned."""
    import deeprank_gnn_amd.synthetic as synth
    from test_gpu_parity import nets
    batch = synth.make_batch(4, 5, n_nodes=45, n_pairs=80, n_feat=7, n_c1=4, n_internal=10).to(_dev())
    assert batch.x.shape[0] % 2 == 1
    torch.manual_seed(9)
    net = nets()[net_name](7, 1, 1).to(_dev())
    net.train()
    fn = lambda o: F.mse_loss(o.reshape(-1), batch.y)           # noqa: E731
    ref_out, ref = _legacy_grads(net, batch, fn)
    for p in net.parameters():
        p.grad = None
    out = net(batch)
    assert _engine(net).last_path == "jacobian"
    fn(out).backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out.cpu().numpy(), rtol=1e-5, atol=1e-5)
    for n, p in net.named_parameters():
        r = ref[n].cpu().numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(r).max())), err_msg=n)


@pytest.mark.parametrize("net_name", ["GINet", "sGAT", "FoutNet"])
@pytest.mark.parametrize("n_feat,task", [(5, "reg"), (16, "class"), (40, "reg")])
def test_ragged_batches_against_the_launch_pair(net_name, n_feat, task):
    """Ragged batches (single-node graphs, no edges, self loops, duplicate edges, odd feature counts), regression and 3-class
    classification with class weights in the CALLER's loss (torch's CrossEntropyLoss(weight), NeuralNet.py:247-263): the
    drop-in boundary against plain autograd through the launch pair + torch head."""
    from step_check import ragged_batch
    from test_gpu_parity import nets
    batch = ragged_batch(7 + n_feat, n_feat)
    n_out = 1 if task == "reg" else 3
    if task == "class":
        batch.y = torch.tensor([k % 3 for k in range(batch.num_graphs)])
        crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.2, 0.5, 0.3], device=_dev()))
    else:
        batch.y = torch.arange(batch.num_graphs, dtype=torch.float32) * 0.3 - 1.0
        crit = torch.nn.MSELoss()
    batch = batch.to(_dev())
    torch.manual_seed(n_feat)
    net = nets()[net_name](n_feat, n_out, 1).to(_dev())
    if hasattr(net, "dropout"):
        net.dropout = 0.0
    net.train()
    fn = (lambda o: crit(o.reshape(-1), batch.y)) if task == "reg" else (lambda o: crit(o, batch.y))
    ref_out, ref = _legacy_grads(net, batch, fn)
    for p in net.parameters():
        p.grad = None
    out = net(batch)
    assert _engine(net).last_path == ("jacobian" if n_out == 1 else "two-launch")
    fn(out).backward()
    torch.cuda.synchronize()
    nan_ok = net_name == "FoutNet"          # (FoutLayer: NaN rows of isolated nodes, dropped by the max-pool)
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out.cpu().numpy(), rtol=1e-4, atol=1e-5, equal_nan=nan_ok)
    for n, p in net.named_parameters():
        r = ref[n].cpu().numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=2e-4, atol=2e-5 * max(1.0, float(np.nanmax(np.abs(r)))),
                                   err_msg=n, equal_nan=nan_ok)


def test_foreign_batch_objects_and_large_batches():
    """A batch that is NOT this package's Batch (attribute access only, as a torch_geometric Batch would be: no host offset
    tables -- Topology.from_batch derives them with one host round trip), and 160 graphs (GINet: both branches in one workgroup;
    no offsets in the kernel arguments): same predictions and gradients as our Batch / as two halves."""
    import types
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.ginet import GINet
    mine = synth.make_batch(0, 12).to(_dev())
    foreign = types.SimpleNamespace(x=mine.x, edge_index=mine.edge_index, edge_attr=mine.edge_attr, batch=mine.batch,
                                    cluster0=mine.cluster0, cluster1=mine.cluster1, y=mine.y, num_graphs=12)
    torch.manual_seed(21)
    net = GINet(32, 1, 1).to(_dev())
    net.dropout = 0.0
    net.train()

    def step(b):
        for p in net.parameters():
            p.grad = None
        out = net(b)
        assert _engine(net).last_path == "jacobian"
        F.mse_loss(out.reshape(-1), b.y).backward()
        torch.cuda.synchronize()
        return out.detach().clone(), {n: p.grad.clone() for n, p in net.named_parameters()}
    a_out, a = step(mine)
    b_out, b = step(foreign)
    assert torch.equal(a_out, b_out)
    for n in a:
        assert torch.equal(a[n], b[n]), n
    big = synth.make_batch(0, 160).to(_dev())      # (2 x 160 workgroups > 256 CUs: one workgroup per graph)
    o_big, g_big = step(big)
    assert _engine(net).last_plan.wgs_per_graph == 1
    halves = [synth.make_batch(0, 80).to(_dev()), synth.make_batch(80, 80).to(_dev())]
    outs, grads = zip(*[step(h) for h in halves])
    np.testing.assert_allclose(o_big.cpu().numpy(), torch.cat(outs).cpu().numpy(), rtol=1e-5, atol=1e-6)
    for n in g_big:      # (mean loss over 160 = mean of the two halves' mean losses)
        want = 0.5 * (grads[0][n] + grads[1][n])
        np.testing.assert_allclose(g_big[n].cpu().numpy(), want.cpu().numpy(), rtol=2e-4, atol=1e-6 * max(1.0, float(want.abs().max())), err_msg=n)


def test_eval_mode_with_gradients_and_parameter_surgery():
    """model.eval() with autograd on (dropout off, gradients wanted: the jacobian launch without a mask); in-place
    load_state_dict keeps the engine, replacing a parameter object or moving the net builds a new one."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.ginet import GINet
    batch = synth.make_batch(0, 8).to(_dev())
    torch.manual_seed(4)
    net = GINet(32, 1, 1).to(_dev())
    net.eval()
    out = net(batch)
    eng = _engine(net)
    assert eng.last_path == "jacobian"
    out.sum().backward()
    with torch.no_grad():
        again = net(batch)
    # no dropout in eval mode: the same numbers (training and inference instances of the kernel: to rounding)
    np.testing.assert_allclose(again.cpu().numpy(), out.detach().cpu().numpy(), rtol=1e-6, atol=1e-6)
    net.load_state_dict({k: v * 0.5 for k, v in net.state_dict().items()})
    assert _engine(net) is eng
    with torch.no_grad():
        halved = net(batch)
    assert not torch.equal(halved, again)            # (the launches read the parameters in place)
    net.fc2.bias = torch.nn.Parameter(torch.zeros_like(net.fc2.bias))
    assert _engine(net) is not eng
    with torch.no_grad():
        net(batch)
    assert _engine(net).last_path == "inference"


def test_classification_loop_recorded_in_a_graph():
    """GINet, two classes, dropout 0.4, CrossEntropyLoss + Adam: the two-launch form of the boundary (forward-only launch with the
    step's mask, training launch fed d loss / d pred) recorded in a hipGraph -- replays equal the eager steps (same mask stream:
    the step index lives on the device)."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.ginet import GINet
    batches = [synth.make_batch(16 * i, 16).to(_dev()) for i in range(4)]
    for b in batches:
        b.y = (b.y > 10).to(torch.int64)

    def fresh():
        torch.manual_seed(17)
        net = GINet(32, 2, 1).to(_dev())
        net.train()
        return net, torch.optim.Adam(net.parameters(), lr=0.01, capturable=True)

    def body(net, opt, b):
        opt.zero_grad(set_to_none=True)
        loss = F.cross_entropy(net(b), b.y)
        loss.backward()
        opt.step()
        return loss
    net, opt = fresh()
    eager_losses = [float(body(net, opt, b).detach()) for b in batches]
    assert _engine(net).last_path == "two-launch"
    eager = {k: v.clone() for k, v in net.state_dict().items()}
    cap, copt = fresh()
    start = copy.deepcopy(cap.state_dict())
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for b in batches:
            body(cap, copt, b)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()

    def reset():
        cap.load_state_dict(start)
        for st in copt.state.values():
            st["step"].zero_(); st["exp_avg"].zero_(); st["exp_avg_sq"].zero_()
        _engine(cap).step2.zero_()
    reset()
    graphs, losses = [], []
    for b in batches:
        gr = torch.cuda.CUDAGraph()
        copt.zero_grad(set_to_none=True)
        with torch.cuda.graph(gr):
            losses.append(body(cap, copt, b))
        graphs.append(gr)
    reset()
    got = []
    for gr, l in zip(graphs, losses):
        gr.replay()
        got.append(float(l.detach()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(got, eager_losses, rtol=1e-5)
    for k, v in cap.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), eager[k].cpu().numpy(), rtol=2e-4, atol=2e-5, err_msg=k)


@pytest.mark.parametrize("net_name", ["GINet", "sGAT", "FoutNet"])
def test_composed_forward_equals_the_fused_path(net_name):
    """The general path (composed.py: the reference's own composition of layers and pooling functions, ginet.py:99-141) on a
    net with the reference's options: the fused path's predictions and gradients."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.composed import composed_forward, default_layers
    from test_gpu_parity import nets
    batch = synth.make_batch(0, 6, n_nodes=60, n_pairs=120, n_feat=20, n_c1=5, n_internal=20).to(_dev())
    torch.manual_seed(31)
    net = nets()[net_name](20, 1, 1).to(_dev())
    if hasattr(net, "dropout"):
        net.dropout = 0.0
    net.train()
    assert default_layers(net)
    out = net(batch)
    assert _engine(net).last_path == "jacobian"
    F.mse_loss(out.reshape(-1), batch.y).backward()
    fused = {n: p.grad.clone() for n, p in net.named_parameters()}
    for p in net.parameters():
        p.grad = None
    x_before = batch.x.clone()
    out2 = composed_forward(net, batch)
    assert torch.equal(batch.x, x_before)                  # the caller's batch is left alone
    F.mse_loss(out2.reshape(-1), batch.y).backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out2.detach().cpu().numpy(), out.detach().cpu().numpy(), rtol=1e-4, atol=1e-5)
    for n, p in net.named_parameters():
        if p.grad is None:          # (GINet's dead attention parameters: no gradient at all through the layer functions' autograd?)
            assert float(fused[n].abs().max()) == 0.0, n
            continue
        r = fused[n].cpu().numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(r).max())), err_msg=n)


@pytest.mark.parametrize("net_name", ["sGAT", "FoutNet"])
def test_layer_options_the_reference_nets_do_not_use(net_name):
    """sGraphAttentionLayer(bias=False, undirected=False) (sGAT.py:50-53,86-87) / FoutLayer(bias=False) (foutnet.py:43-46) inside
    the nets: model(batch) takes the general path and matches the oracle's composition of the same layer functions."""
    import deeprank_gnn_amd.synthetic as synth
    from oracle import cpu_ref
    from deeprank_gnn_amd.sGAT import sGAT, sGraphAttentionLayer
    from deeprank_gnn_amd.foutnet import FoutNet, FoutLayer
    batch_cpu = synth.make_batch(0, 5, n_nodes=50, n_pairs=100, n_feat=12, n_c1=4, n_internal=20)
    torch.manual_seed(33)
    if net_name == "sGAT":
        net = sGAT(12, 1, 1)
        net.conv1 = sGraphAttentionLayer(12, 16, bias=False, undirected=False)
        net.conv2 = sGraphAttentionLayer(16, 32, bias=False, undirected=False)
        conv = lambda pre: (lambda x, ei, ea: cpu_ref.sgat_conv(x, ei, ea, P[pre + ".weight"], None, undirected=False))   # noqa: E731
    else:
        net = FoutNet(12, 1, 1)
        net.conv1 = FoutLayer(12, 16, bias=False)
        net.conv2 = FoutLayer(16, 32, bias=False)
        conv = lambda pre: (lambda x, ei, ea: cpu_ref.fout_conv(x, ei, P[pre + ".Wc"], P[pre + ".Wn"], None, looped=False))  # noqa: E731
    P = {k: v.detach().clone().requires_grad_(True) for k, v in net.state_dict().items()}
    x2, b2 = cpu_ref._branch(cpu_ref._as_ns(batch_cpu), conv("conv1"), conv("conv2"), None, "a.")
    feat = cpu_ref.scatter_mean(x2, b2)
    ref = F.linear(F.relu(F.linear(feat, P["fc1.weight"], P["fc1.bias"])), P["fc2.weight"], P["fc2.bias"])
    F.mse_loss(ref.reshape(-1), batch_cpu.y).backward()
    net = net.to(_dev())
    net.train()
    batch = batch_cpu.clone().to(_dev())
    out = net(batch)
    assert net.__dict__.get("_drgnn_engine") is None        # the general path: no fused step for these options
    F.mse_loss(out.reshape(-1), batch.y).backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(out.detach().cpu().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-4)
    for n, p in net.named_parameters():
        r = P[n].grad.numpy()
        np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=1e-3, atol=1e-4 * max(1.0, float(np.abs(r).max())), err_msg=n)


@pytest.mark.parametrize("net_name,task", [("GINet", "reg"), ("sGAT", "reg"), ("FoutNet", "reg"), ("GINet", "class")])
def test_reference_trainer_call_pattern_on_the_fixture(net_name, task):
    """BASELINE configs[0]'s shape through the boundary the way the reference trainer drives it (NeuralNet.py:153-154,
    183-184,239-263,489-523): a DataLoader over the fixture's ten 1ATN graphs (batch 8 + a ragged last batch of 2, shuffled),
    `data_batch.to(device)`, `model(data_batch)`, MSELoss / CrossEntropyLoss on `.reshape(-1)` / logits, `loss.backward()`,
    `torch.optim.Adam(model.parameters(), lr)`, `F.softmax(pred, dim=1)` for the class outputs; then `model.eval()` + no_grad
    over the same loader.  Three epochs: the loss falls, every step ran the fused kernels, and a twin net driven through the
    launch pair (fused path switched off) ends with the same parameters."""
    from helpers import fixture_graphs
    from deeprank_gnn_amd.data import DataLoader
    from deeprank_gnn_amd import fused_autograd
    from test_gpu_parity import nets
    graphs = fixture_graphs(target="irmsd" if task == "reg" else "binclass")
    n_feat = int(graphs[0].x.shape[1])
    n_out = 1 if task == "reg" else 2

    def run(fused):
        torch.manual_seed(41)
        net = nets()[net_name](n_feat, n_out, 1).to(_dev())
        if hasattr(net, "dropout"):
            net.dropout = 0.0
        opt = torch.optim.Adam(net.parameters(), lr=0.005)
        crit = torch.nn.MSELoss() if task == "reg" else torch.nn.CrossEntropyLoss()
        gen = torch.Generator().manual_seed(3)
        loader = DataLoader(graphs, batch_size=8, shuffle=True, generator=gen)
        saved = fused_autograd.StepEngine.run
        if not fused:
            fused_autograd.StepEngine.run = lambda self, data, topo=None: None        # (every call takes the launch pair)
        try:
            losses, paths = [], set()
            for epoch in range(3):
                net.train()
                running = 0.0
                for data_batch in loader:
                    data_batch = data_batch.to(_dev())
                    opt.zero_grad()
                    pred = net(data_batch)
                    if fused:
                        paths.add(fused_autograd.engine_for(net).last_path)
                    y = data_batch.y if task == "reg" else data_batch.y.to(torch.int64)
                    loss = crit(pred.reshape(-1), y) if task == "reg" else crit(pred, y)
                    running += float(loss.detach())
                    loss.backward()
                    opt.step()
                    if task == "class":
                        assert bool(torch.isfinite(F.softmax(pred, dim=1)).all())
                losses.append(running)
            net.eval()
            outs = []
            with torch.no_grad():
                for data_batch in DataLoader(graphs, batch_size=8, shuffle=False):
                    outs.append(net(data_batch.to(_dev())))
            return net, losses, paths, torch.cat(outs)
        finally:
            fused_autograd.StepEngine.run = saved
    net_a, losses_a, paths, out_a = run(True)
    assert paths == {"jacobian" if n_out == 1 else "two-launch"}, paths
    assert losses_a[-1] < losses_a[0] and np.isfinite(losses_a).all()
    net_b, losses_b, _, out_b = run(False)
    np.testing.assert_allclose(losses_a, losses_b, rtol=2e-3)
    np.testing.assert_allclose(out_a.cpu().numpy(), out_b.cpu().numpy(), rtol=5e-3, atol=5e-3)
    for (n, p), (_, q) in zip(net_a.named_parameters(), net_b.named_parameters()):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=5e-3, atol=5e-4, err_msg=n)


@pytest.mark.parametrize("net_name", ["GINet", "sGAT", "FoutNet"])
def test_boundary_edge_cases(net_name):
    """One graph per batch, one feature, a batch whose graphs have no edges at all, and 64 + 1 graphs (the host offset tables
    stop travelling in the kernel arguments): the fused boundary against the launch pair (FoutNet: NaN rows of isolated nodes
    are dropped by the max-pool; an all-isolated graph gives a NaN prediction in both)."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.data import Batch
    from test_gpu_parity import nets
    cases = {
        "one graph": lambda: synth.make_batch(3, 1, n_nodes=30, n_pairs=50, n_feat=8, n_c1=3, n_internal=10),
        "one feature": lambda: synth.make_batch(0, 5, n_nodes=30, n_pairs=50, n_feat=1, n_c1=3, n_internal=10),
        "65 graphs": lambda: synth.make_batch(0, 65, n_nodes=24, n_pairs=40, n_feat=8, n_c1=3, n_internal=8),
    }

    def no_edges():
        gs = [synth.make_graph(i, n_nodes=12, n_pairs=20, n_feat=8, n_c1=3, n_internal=4) for i in range(3)]
        for g in gs[:2]:
            g.edge_index = g.edge_index[:, :0]
            g.edge_attr = g.edge_attr[:0]
        return Batch.from_data_list(gs)
    cases["graphs without edges"] = no_edges
    for what, make in cases.items():
        batch = make().to(_dev())
        n_feat = int(batch.x.shape[1])
        torch.manual_seed(5)
        net = nets()[net_name](n_feat, 1, 1).to(_dev())
        if hasattr(net, "dropout"):
            net.dropout = 0.0
        net.train()
        fn = lambda o: F.mse_loss(o.reshape(-1), batch.y)        # noqa: E731
        ref_out, ref = _legacy_grads(net, batch, fn)
        for p in net.parameters():
            p.grad = None
        out = net(batch)
        assert _engine(net).last_path == "jacobian", (what, _engine(net).last_path)
        fn(out).backward()
        torch.cuda.synchronize()
        nan_ok = net_name == "FoutNet"
        np.testing.assert_allclose(out.detach().cpu().numpy(), ref_out.cpu().numpy(), rtol=1e-4, atol=1e-5, equal_nan=nan_ok, err_msg=what)
        for n, p in net.named_parameters():
            r = ref[n].cpu().numpy()
            scale = float(np.nanmax(np.abs(r))) if np.isfinite(r).any() else 1.0
            np.testing.assert_allclose(p.grad.cpu().numpy(), r, rtol=2e-4, atol=2e-5 * max(1.0, scale), equal_nan=nan_ok,
                                       err_msg="%s %s" % (what, n))


@pytest.mark.parametrize("net_name,n_feat,n_nodes", [("GINet", 48, 380), ("GINet", 64, 330), ("sGAT", 32, 340), ("FoutNet", 48, 300),
                                                       # (sGAT / FoutNet with the S rows left in memory as well)
                                                       ("sGAT", 48, 380), ("FoutNet", 64, 390), ("sGAT", 64, 330), ("FoutNet", 44, 370),
                                                       ("sGAT", 32, 395), ("FoutNet", 30, 390)])
def test_graphs_beyond_the_staged_layout_through_the_drop_in_boundary(net_name, n_feat, n_nodes):
    """model(batch) / loss.backward() on graphs beyond the staged kernels' LDS budget: the from-memory instances
    (plan.from_memory) -- for the GINet shapes also beyond what the builder stages an x tile for, so the engine has the
    aggregation tiles formed by the stand-alone launch behind the build -- against the oracle."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.data import Batch
    from oracle import cpu_ref
    shape = dict(n_nodes=n_nodes, n_pairs=(5 * n_nodes) // 2, n_c1=max(4, n_nodes // 12), n_internal=(7 * n_nodes) // 4)
    batch_cpu = Batch.from_data_list([synth.make_graph(i, n_feat=n_feat, **shape) for i in range(10)])
    params = cpu_ref.init_params(net_name, n_feat, 1, 1, seed=29)
    kw = _fw(net_name)
    ref_pred, ref_loss, ref_grads = cpu_ref.loss_and_grads(net_name, params, batch_cpu, batch_cpu.y, **kw)
    net = _build(net_name, params, 1)
    net.train()
    batch = batch_cpu.clone().to(_dev())
    out = net(batch)
    eng = _engine(net)
    assert eng.last_path == "jacobian" and eng.last_plan.family == _lib.STEP_FAMILY_AGGREGATE and eng.last_plan.from_memory, \
        eng.last_reason
    topo = eng.topology_for(batch)
    assert bool(getattr(topo, "_tiles_separately", False)) == (not _lib.get().topology_tiles_ok(n_nodes, 5 * n_nodes, n_feat))
    loss = F.mse_loss(out.reshape(-1), batch.y)
    loss.backward()
    torch.cuda.synchronize()
    stats = new_stats()
    check_step("%s F=%d, %d nodes [drop-in, from memory]" % (net_name, n_feat, n_nodes), Lazy64(net_name, params, batch_cpu, **kw),
               loss.item(), out.detach().cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in net.named_parameters()}, ref_loss,
               ref_pred.numpy(), {k: v.numpy() for k, v in ref_grads.items()}, stats)
    assert_arbiter_rate(stats, net_name)
    net.eval()
    with torch.no_grad():
        pred = net(batch)
    assert eng.last_path == "inference" and eng.last_plan.from_memory
    np.testing.assert_allclose(pred.cpu().numpy(), out.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_classification_on_graphs_beyond_the_staged_layout():
    """Several outputs on such graphs: the forward-only from-memory instance, then the training instance fed d loss / d pred
    (two launches); cross-entropy against the oracle."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.data import Batch
    from oracle import cpu_ref
    n_feat, n_nodes = 32, 360
    shape = dict(n_nodes=n_nodes, n_pairs=(5 * n_nodes) // 2, n_c1=max(4, n_nodes // 12), n_internal=(7 * n_nodes) // 4)
    batch_cpu = Batch.from_data_list([synth.make_graph(i, n_feat=n_feat, **shape) for i in range(9)])
    target = torch.tensor([0, 2, 1, 1, 0, 2, 2, 0, 1])
    for net_name in ("GINet", "FoutNet"):
        params = cpu_ref.init_params(net_name, n_feat, 3, 1, seed=31)
        kw = _fw(net_name)
        ref_pred, ref_loss, ref_grads = cpu_ref.loss_and_grads(net_name, params, batch_cpu, target, task="class", **kw)
        net = _build(net_name, params, 3)
        net.train()
        batch = batch_cpu.clone().to(_dev())
        out = net(batch)
        eng = _engine(net)
        assert eng.last_path == "two-launch" and eng.last_plan.from_memory, eng.last_reason
        loss = F.cross_entropy(out, target.to(_dev()))
        loss.backward()
        torch.cuda.synchronize()
        stats = new_stats()
        check_step("%s 3 classes, %d nodes [drop-in, from memory]" % (net_name, n_nodes),
                   Lazy64(net_name, params, batch_cpu, target=target, task="class", **kw), loss.item(), out.detach().cpu().numpy(),
                   {k: p.grad.cpu().numpy() for k, p in net.named_parameters()}, ref_loss, ref_pred.numpy(),
                   {k: v.numpy() for k, v in ref_grads.items()}, stats)
        assert_arbiter_rate(stats, net_name)
