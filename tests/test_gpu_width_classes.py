"""The aggregation-first step kernels (csrc/drgnn_step2.h / drgnn_step3.h) -- the family bench.py times -- pinned DIRECTLY on
the reference-generated goldens, for every padded feature width they are instantiated for (16 / 32 / 48 / 64), in every launch
layout, for training and inference launches.

Round 4 pinned these kernels through the oracle only (itself pinned on the goldens: two hops), and only at width 32; the
goldens reached the autograd path alone (missing #2 / #3).  Here ``FusedTrainer.compute_gradients`` -- asserting
that its launch plan IS the aggregation-first family -- is compared element-wise with ``out`` / ``loss`` / ``grad/*`` recorded
from the reference's own ginet.py / sGAT.py / foutnet.py (tests/golden/gen/make_golden.py, make_width_golden.py):

    fix8_*    28 features -> width 32 (the reference's fixture graphs, 1ATN)       syn4_* / iso3_*   12 -> 16
    mid4_*    44 features -> width 48                                               wide4_*           52 -> 64
    pretrained_treg   the reference's SHIPPED regression model, 48 features -> 48 (paper_pretrained_models/
                      scoring_of_docking_models/treg_yfnat_b128_e20_lr0.001_20.pth.tar), loaded strict=True

Reference path: ginet.py:50-73,99-141, sGAT.py:62-93,114-138, foutnet.py:56-82,103-125 (:71-73 for the NaN rows of iso3),
NeuralNet.py:489-506 for the step.  Tolerance: tests/elementwise.py (1e-4 + 1e-4 |ref| per element, float64 arbiter <= 0.1 %).
"""
import numpy as np
import pytest
import torch

from helpers import CASES, STEP_ONLY_CASES, golden, params_of
from oracle import cpu_ref
from elementwise import Lazy64, check, check_step, new_stats, assert_arbiter_rate

pytestmark = pytest.mark.gpu
ALL_CASES = dict(CASES, **STEP_ONLY_CASES)
# launch layouts per kind: plan overrides and the workgroups per graph they must give
LAYOUTS = {"GINet": [("two", {}, 2), ("one", {"force_wgs": 1}, 1)],
           "sGAT": [("split", {}, 2), ("whole", {"no_split": 1}, 1)],
           "FoutNet": [("split", {}, 2), ("whole", {"no_split": 1}, 1)]}


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _fw(net_name):
    return {"looped": False} if net_name == "FoutNet" else {}


def _trainer(net_name, params, n_out, task):
    from test_gpu_parity import build
    from deeprank_gnn_amd.trainer import FusedTrainer
    net = build(net_name, params, n_out)          # load_state_dict(strict=True), dropout 0
    return net, FusedTrainer(net, lr=0.01, task=task)


@pytest.mark.parametrize("fname", sorted(ALL_CASES))
def test_fused_step_vs_reference_golden(fname):
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.topology import Topology
    net_name, make_batch, task = ALL_CASES[fname]
    g = golden(fname)
    params = params_of(g)
    n_out = g["out"].shape[1]
    target_cpu = torch.from_numpy(g["target"])
    batch_cpu = make_batch()
    batch_cpu.y = target_cpu
    lazy = Lazy64(net_name, params, make_batch(), target=target_cpu, task=task, **_fw(net_name))
    batch = batch_cpu.clone().to(_dev())
    n_feat = int(batch.x.shape[1])
    ref_grads = {k[len("grad/"):]: v for k, v in g.items() if k.startswith("grad/")}
    for name, ov, wgs in LAYOUTS[net_name]:
        net, tr = _trainer(net_name, params, n_out, task)
        tr.plan_overrides = dict(ov)
        topo = Topology.from_batch(batch, need_weights=(net_name == "sGAT"))
        c = tr._fused_prepare(batch, topo)
        plan = c["plan"]
        assert plan.family == _lib.STEP_FAMILY_AGGREGATE, (fname, name, plan.family)
        assert plan.width == ((n_feat + 15) // 16) * 16 and plan.wgs_per_graph == wgs, (plan.width, plan.wgs_per_graph)
        loss = tr.compute_gradients(batch, topo=topo)
        torch.cuda.synchronize()
        assert tr.faults() == 0
        grads = {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}
        stats = new_stats()
        check_step("%s [%s]" % (fname, name), lazy, float(loss), tr.last_pred.cpu().numpy(), grads, float(g["loss"]), g["out"],
                   ref_grads, stats)
        assert stats["arbiter"] <= max(3, stats["elements"] // 1000), stats
        # the inference launch of the same family on the same workspace: the training launch's predictions (dropout 0)
        assert tr._plan_for(topo, n_feat, None, False, batch.x).family == _lib.STEP_FAMILY_AGGREGATE
        pred = tr.predict(batch, topo=topo)
        torch.cuda.synchronize()
        check(fname + " inference", pred.cpu().numpy(), g["out"], lazy.pred, new_stats())
        if net_name == "GINet" and name == "two":       # the exchange words of an inference launch are left clean: again
            np.testing.assert_array_equal(tr.predict(batch, topo=topo).cpu().numpy(), pred.cpu().numpy())


def test_shipped_regression_model_inference_known_answer():
    """treg_yfnat (48 features): eval-mode predictions recorded from the reference's GINet, batched and graph by graph,
    through the aggregation-first inference kernels (k_step3_co_topo<48, ., ., false>)."""
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.topology import Topology
    from helpers import treg_graphs
    g = golden("pretrained_treg.npz")
    params = params_of(g)
    net, tr = _trainer("GINet", params, 1, "reg")
    net.eval()
    graphs = treg_graphs()
    batch = Batch.from_data_list(graphs).to(_dev())
    topo = Topology.from_batch(batch, need_weights=False)
    plan = tr._plan_for(topo, 48, None, False, batch.x)
    assert plan.family == _lib.STEP_FAMILY_AGGREGATE and plan.width == 48
    np.testing.assert_allclose(tr.predict(batch, topo=topo).cpu().numpy(), g["pred_batched"], rtol=1e-4, atol=1e-4)
    one = torch.cat([tr.predict(Batch.from_data_list([gr]).to(_dev())) for gr in graphs])
    np.testing.assert_allclose(one.cpu().numpy(), g["pred_single"], rtol=1e-4, atol=1e-4)
    # the drop-in module (autograd path) gives the same numbers
    with torch.no_grad():
        np.testing.assert_allclose(net(batch).cpu().numpy(), g["pred_batched"], rtol=1e-4, atol=1e-4)


# (sGAT / FoutNet keep the S AND the x rows of a graph in LDS: 200-node graphs fit that way up to width 48 -- 157 - 160 KB since
# conv1's activations share their place with the pooled level's [S | T] and Z2.  At 49 - 64 features the 64-wide kernels stage both
# tiles for graphs up to ~150 nodes (the 120-node cases) and, round 6, read the x rows from memory beyond that (drgnn_step2.h, XG:
# the 200-node cases; 52 = rows shorter than the padded width, 50 = the tiles' padded x copy) -- every reference net shape up to
# 64 features runs a fused step)
SHAPES = [("GINet", 48, 200, 128), ("GINet", 16, 200, 64), ("GINet", 64, 200, 64),
          ("sGAT", 16, 200, 64), ("sGAT", 48, 200, 128), ("sGAT", 64, 120, 64),
          ("FoutNet", 16, 200, 64), ("FoutNet", 48, 200, 128), ("FoutNet", 64, 120, 64),
          ("sGAT", 64, 200, 64), ("FoutNet", 64, 200, 64), ("FoutNet", 52, 200, 64), ("sGAT", 50, 200, 64),
          # graphs beyond the LDS budget of the staged forms, up to what the builder forms tiles for: GINet's one-workgroup kernel
          # with the S rows left in memory (drgnn_step3.h, SG), sGAT / FoutNet with the x rows left in memory at every width
          ("GINet", 32, 340, 32), ("GINet", 48, 300, 32), ("GINet", 16, 400, 24), ("GINet", 64, 250, 32), ("GINet", 30, 330, 24),
          ("sGAT", 32, 340, 32), ("FoutNet", 32, 340, 32), ("sGAT", 48, 300, 32), ("FoutNet", 48, 300, 32), ("FoutNet", 16, 400, 24),
          ("sGAT", 30, 330, 24)]


@pytest.mark.parametrize("net_name,n_feat,n_nodes,B", SHAPES)
def test_fused_step_other_widths_match_oracle_elementwise(net_name, n_feat, n_nodes, B):
    """The reference's shipped regression models are 48 features at batch 128: 128 SYN-shaped graphs (200 nodes,
    ~1000 edges) with 48 features -- and 64 graphs at the widths 16 / 64 -- through the launches that shape takes by itself
    (GINet at batch 128: both branches in one workgroup, k_step3b_co_topo; sGAT / FoutNet: one workgroup per graph), rebuilt
    (with the next topology co-built, lean) and out of a cached topology, element-wise against the oracle; cached == rebuilt
    bit for bit where the layouts agree."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.resident import ResidentGraphSet
    from deeprank_gnn_amd.topology import Topology
    from topo_check import check_against_oracle
    dev = _dev()
    shape = dict(n_nodes=n_nodes, n_pairs=(5 * n_nodes) // 2, n_c1=max(4, n_nodes // 12), n_internal=(7 * n_nodes) // 4)
    graphs = [synth.make_graph(i, n_feat=n_feat, **shape) for i in range(B)]
    from deeprank_gnn_amd.data import Batch
    batch_cpu = Batch.from_data_list(graphs)
    params = cpu_ref.init_params(net_name, n_feat, 1, 1, seed=17)
    kw = _fw(net_name)
    ref_pred, ref_loss, ref_grads = cpu_ref.loss_and_grads(net_name, params, batch_cpu, batch_cpu.y, **kw)
    lazy = Lazy64(net_name, params, batch_cpu, **kw)
    net, tr = _trainer(net_name, params, 1, "reg")
    batch = batch_cpu.clone().to(dev)
    need_w = net_name == "sGAT"
    topo = Topology.from_batch(batch, need_weights=need_w)
    nxt = Topology.from_batch(batch, need_weights=need_w, build=False)
    c = tr._fused_prepare(batch, topo, True, nxt)
    assert c["plan"].family == _lib.STEP_FAMILY_AGGREGATE, (net_name, n_feat, c["plan"].family)
    assert c["plan"].width == ((n_feat + 15) // 16) * 16
    loss = tr.compute_gradients(batch, topo=topo, next_topo=nxt)
    torch.cuda.synchronize()
    assert tr.faults() == 0
    stats = new_stats()
    grads = {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}
    check_step("%s %d graphs of %d nodes F=%d" % (net_name, B, n_nodes, n_feat), lazy, float(loss), tr.last_pred.cpu().numpy(), grads, ref_loss,
               ref_pred.numpy(), {k: v.numpy() for k, v in ref_grads.items()}, stats)
    assert_arbiter_rate(stats, "%s F=%d" % (net_name, n_feat))
    assert nxt.status()[0] == 0
    check_against_oracle(nxt, batch_cpu, weights=need_w)
    loss2 = tr.compute_gradients(batch, topo=nxt)          # the co-built (lean) workspace under the same kind of launch
    assert float(loss2) == float(loss)
    # cached topology of the same graphs: the same bits
    rs = ResidentGraphSet(graphs, dev)
    cache = rs.topology_cache(need_weights=need_w)
    net2, tr2 = _trainer(net_name, params, 1, "reg")
    cc = tr2._cached_prepare(cache, list(range(B)))
    assert cc["plan"].family == _lib.STEP_FAMILY_AGGREGATE and cc["plan"].width == ((n_feat + 15) // 16) * 16
    tr2.train_step_cached(cache, list(range(B)), apply_adam=False)
    torch.cuda.synchronize()
    if cc["plan"].wgs_per_graph == c["plan"].wgs_per_graph:
        assert float(tr2.loss) == float(loss)
        for k, p in net2.named_parameters():
            np.testing.assert_array_equal(p.grad.detach().cpu().numpy(), grads[k], err_msg=k)
    else:       # (no builder in a cached launch: it may be resident with two workgroups per graph where the rebuilt one is not)
        np.testing.assert_allclose(float(tr2.loss), float(loss), rtol=1e-5)
    pc = tr2.predict_cached(cache, list(range(B)))
    check("cached inference", pc.cpu().numpy(), ref_pred.numpy(), lazy.pred, new_stats())
    beyond = n_nodes > 200 or (net_name != "GINet" and n_feat > 48 and n_nodes >= 200)
    if not beyond:
        assert not c["plan"].from_memory
        return
    # Graphs beyond the staged kernels' LDS: the instances that leave the node-sized tile in memory (plan.from_memory).  sGAT /
    # FoutNet took it above; GINet's is the one-workgroup kernel, which a batch this small reaches only when told to (two
    # workgroups per graph hold less each): the same numbers again through it, training and inference.
    if net_name != "GINet":
        assert c["plan"].from_memory and cc["plan"].from_memory, (net_name, n_feat, n_nodes)
        return
    net3, tr3 = _trainer(net_name, params, 1, "reg")
    tr3.plan_overrides = {"force_wgs": 1}
    topo3 = Topology.from_batch(batch, need_weights=False)
    c3 = tr3._fused_prepare(batch, topo3, True, None)
    assert c3["plan"].family == _lib.STEP_FAMILY_AGGREGATE and c3["plan"].wgs_per_graph == 1 and c3["plan"].from_memory, \
        (n_feat, n_nodes, c3["plan"].family, c3["plan"].wgs_per_graph, c3["plan"].from_memory)
    loss3 = tr3.compute_gradients(batch, topo=topo3)
    torch.cuda.synchronize()
    assert tr3.faults() == 0
    stats3 = new_stats()
    check_step("%s %d graphs of %d nodes F=%d, S rows from memory" % (net_name, B, n_nodes, n_feat), lazy, float(loss3),
               tr3.last_pred.cpu().numpy(), {k: p.grad.detach().cpu().numpy() for k, p in net3.named_parameters()}, ref_loss,
               ref_pred.numpy(), {k: v.numpy() for k, v in ref_grads.items()}, stats3)
    assert_arbiter_rate(stats3, "%s F=%d from memory" % (net_name, n_feat))
    p3 = tr3.predict(batch, topo=topo3)
    check("inference, S rows from memory", p3.cpu().numpy(), ref_pred.numpy(), lazy.pred, new_stats())
    net4, tr4 = _trainer(net_name, params, 1, "reg")
    tr4.plan_overrides = {"force_wgs": 1}
    c4 = tr4._cached_prepare(cache, list(range(B)))
    assert c4["plan"].from_memory and c4["plan"].wgs_per_graph == 1
    tr4.train_step_cached(cache, list(range(B)), apply_adam=False)
    torch.cuda.synchronize()
    assert float(tr4.loss) == float(loss3)                 # cached topology, gathered: the same bits
    for k, p in net4.named_parameters():
        np.testing.assert_array_equal(p.grad.detach().cpu().numpy(), net3.get_parameter(k).grad.detach().cpu().numpy(), err_msg=k)


def test_two_trainers_keep_their_own_forced_layouts_in_one_process():
    """The launch plan's overrides belong to a trainer, not to the process: two trainers with different
    forced layouts, stepped alternately, each run their own layout and give the numbers of a trainer stepped alone."""
    import copy
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.sGAT import sGAT
    from deeprank_gnn_amd.topology import Topology
    from deeprank_gnn_amd.trainer import FusedTrainer
    dev = _dev()
    batch = synth.make_batch(0, 16).to(dev)
    torch.manual_seed(1)
    base = sGAT(32, 1, 1)
    base.dropout = 0.0
    want = {"split": ({}, _lib.STEP_FAMILY_AGGREGATE, 2), "whole": ({"no_split": 1}, _lib.STEP_FAMILY_AGGREGATE, 1),
            "product": ({"no_aggregate": 1}, _lib.STEP_FAMILY_NONE, 0)}       # (the launch pair: no fused kernel of that family on the device)
    solo = {}
    for name, (ov, fam, wgs) in want.items():
        tr = FusedTrainer(copy.deepcopy(base).to(dev), lr=0.01, task="reg")
        tr.plan_overrides = dict(ov)
        topo = Topology.from_batch(batch, need_weights=True)
        for _ in range(3):
            tr.train_step(batch, topo=topo)
        solo[name] = tr.flat_p.detach().cpu().numpy().copy()
    trs = {}
    for name, (ov, fam, wgs) in want.items():
        trs[name] = FusedTrainer(copy.deepcopy(base).to(dev), lr=0.01, task="reg")
        trs[name].plan_overrides = dict(ov)
    topo = Topology.from_batch(batch, need_weights=True)
    for _ in range(3):
        for name, (ov, fam, wgs) in want.items():
            plan = trs[name]._fused_prepare(batch, topo)["plan"]
            assert (plan.family, plan.wgs_per_graph) == (fam, wgs), (name, plan.family, plan.wgs_per_graph)
            trs[name].train_step(batch, topo=topo)
    torch.cuda.synchronize()
    for name in want:
        np.testing.assert_array_equal(trs[name].flat_p.detach().cpu().numpy(), solo[name], err_msg=name)
    np.testing.assert_allclose(solo["split"], solo["product"], rtol=1e-3, atol=1e-4)


def test_stale_tiles_are_not_used():
    """a Topology built with tiles bakes the neighbour sums of ITS x in; stepping other node features (a new
    tensor, or the same one modified in place) must not mix the old sums with the new rows."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd.topology import Topology
    dev = _dev()
    batch_cpu = synth.make_batch(0, 8)
    params = cpu_ref.init_params("FoutNet", 32, 1, 1, seed=2)
    batch = batch_cpu.clone().to(dev)
    topo = Topology.from_batch(batch, need_weights=False)
    for mode in ("in_place", "new_tensor"):
        net, tr = _trainer("FoutNet", params, 1, "reg")
        b2 = batch_cpu.clone()
        b2.x = b2.x * 1.5 + 0.25
        ref_pred, ref_loss, _ = cpu_ref.loss_and_grads("FoutNet", params, b2, b2.y, looped=False)
        if mode == "in_place":
            batch.x.mul_(1.5).add_(0.25)
            stepped = batch
        else:
            stepped = b2.clone().to(dev)
        loss = tr.compute_gradients(stepped, topo=topo)
        torch.cuda.synchronize()
        np.testing.assert_allclose(float(loss), float(ref_loss), rtol=1e-4)
        np.testing.assert_allclose(tr.last_pred.cpu().numpy(), ref_pred.numpy(), rtol=1e-4, atol=1e-4)
        batch = batch_cpu.clone().to(dev)
        topo = Topology.from_batch(batch, need_weights=False)


def test_stand_alone_tiles_launch_gives_the_builders_bits():
    """drgnn_topology_tiles (the launch that forms aggregation tiles from a BUILT workspace: the other flavour of a shared cache,
    and the tiles of a set whose largest graph the builder cannot stage) == the tiles the builder forms itself, bit for bit,
    plain and edge-weighted, with rows shorter than the padded width too."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.resident import ResidentGraphSet
    dev = _dev()
    for n_feat in (32, 30):
        graphs = [synth.make_graph(i, n_feat=n_feat) for i in range(12)]
        rs = ResidentGraphSet(graphs, dev)
        for weighted in (False, True):
            cache = rs.topology_cache(need_weights=weighted)
            t = cache.topo
            assert t.flags & _lib.TOPO_TILES
            again = torch.full_like(t.tiles, float("nan"))
            rs.api.topology_tiles(t.ws_i32, t.ws_f32, t.n_nodes, t.n_edges, t.n_graphs, rs.x, n_feat, weighted, again,
                                  _lib.current_stream(rs.x))
            torch.cuda.synchronize()
            n = rs.api.topology_tiles_elems(t.n_nodes, n_feat)
            assert torch.equal(again[:n].view(torch.int32), t.tiles[:n].view(torch.int32)), (n_feat, weighted)


def test_one_large_graph_does_not_take_the_fused_kernels_from_the_rest_of_the_set():
    """A resident set whose LARGEST graph is beyond what the builder stages an x tile for (380 nodes at 48 features; the limit is
    310) still gets aggregation tiles -- formed by the stand-alone launch -- so its mini-batches keep the fused kernels: the ones
    of ordinary graphs the staged instances, the one with the large graph the from-memory instances.  Each against the oracle."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.resident import ResidentGraphSet
    dev = _dev()
    n_feat = 48

    def shape(n):
        return dict(n_nodes=n, n_pairs=(5 * n) // 2, n_c1=max(4, n // 12), n_internal=(7 * n) // 4)
    graphs = [synth.make_graph(i, n_feat=n_feat, **shape(150)) for i in range(23)] + [synth.make_graph(99, n_feat=n_feat, **shape(380))]
    assert not _lib.get().topology_tiles_ok(380, 1900, n_feat)
    rs = ResidentGraphSet(graphs, dev)
    for net_name in ("GINet", "sGAT"):
        need_w = net_name == "sGAT"
        cache = rs.topology_cache(need_weights=need_w)
        assert cache.topo.flags & _lib.TOPO_TILES and cache.topo.tiles is not None
        params = cpu_ref.init_params(net_name, n_feat, 1, 1, seed=23)
        for ids, big in ((list(range(0, 12)), False), (list(range(12, 24)), True)):
            net, tr = _trainer(net_name, params, 1, "reg")
            c = tr._cached_prepare(cache, ids)
            assert c["plan"].family == _lib.STEP_FAMILY_AGGREGATE and bool(c["plan"].from_memory) == big, (net_name, big)
            loss = tr.train_step_cached(cache, ids, apply_adam=False)
            torch.cuda.synchronize()
            assert tr.faults() == 0
            batch_cpu = Batch.from_data_list([graphs[i] for i in ids])
            kw = _fw(net_name)
            ref_pred, ref_loss, ref_grads = cpu_ref.loss_and_grads(net_name, params, batch_cpu, batch_cpu.y, **kw)
            stats = new_stats()
            check_step("%s cached, %s graphs" % (net_name, "with the 380-node graph" if big else "150-node"),
                       Lazy64(net_name, params, batch_cpu, **kw), float(loss), tr.last_pred.cpu().numpy(),
                       {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}, ref_loss, ref_pred.numpy(),
                       {k: v.numpy() for k, v in ref_grads.items()}, stats)
            assert_arbiter_rate(stats, net_name)
    # ... and the native epoch loop over that set (cached topology, GINet): mini-batch by mini-batch the same launches
    cache = rs.topology_cache(need_weights=False)
    params = cpu_ref.init_params("GINet", n_feat, 1, 1, seed=23)
    order = list(range(24))
    net_a, tr_a = _trainer("GINet", params, 1, "reg")
    done = tr_a.train_epoch(rs, order, 12, cached=True)
    assert done is not None
    losses, pred = done
    net_b, tr_b = _trainer("GINet", params, 1, "reg")
    want = [float(tr_b.train_step_cached(cache, order[k:k + 12])) for k in (0, 12)]
    torch.cuda.synchronize()
    assert [float(v) for v in losses.cpu()] == want
    for (k, pa), (_, pb) in zip(net_a.named_parameters(), net_b.named_parameters()):
        assert torch.equal(pa, pb), k


@pytest.mark.parametrize("net_name,n_feat,n_nodes", [("sGAT", 48, 380), ("FoutNet", 64, 390), ("sGAT", 64, 300), ("FoutNet", 32, 395),
                                                      ("GINet", 48, 390), ("GINet", 20, 400)])
def test_largest_graphs_of_the_fused_kernels_on_a_cached_set(net_name, n_feat, n_nodes):
    """The far end of the fused kernels' reach (~400 nodes, 2000 edges per graph) out of a resident set's cached topology: the
    set's tiles come from the stand-alone launch (the builder stages no x tile for graphs this large), the step is the
    from-memory instance (sGAT / FoutNet: neither the x nor the S rows staged), training and inference against the oracle."""
    import deeprank_gnn_amd.synthetic as synth
    from deeprank_gnn_amd import _lib
    from deeprank_gnn_amd.data import Batch
    from deeprank_gnn_amd.resident import ResidentGraphSet
    dev = _dev()
    B = 12
    shape = dict(n_nodes=n_nodes, n_pairs=(5 * n_nodes) // 2, n_c1=max(4, n_nodes // 12), n_internal=(7 * n_nodes) // 4)
    graphs = [synth.make_graph(i, n_feat=n_feat, **shape) for i in range(B)]
    batch_cpu = Batch.from_data_list(graphs)
    params = cpu_ref.init_params(net_name, n_feat, 1, 1, seed=37)
    kw = _fw(net_name)
    ref_pred, ref_loss, ref_grads = cpu_ref.loss_and_grads(net_name, params, batch_cpu, batch_cpu.y, **kw)
    lazy = Lazy64(net_name, params, batch_cpu, **kw)
    rs = ResidentGraphSet(graphs, dev)
    cache = rs.topology_cache(need_weights=(net_name == "sGAT"))
    assert cache.topo.flags & _lib.TOPO_TILES
    net, tr = _trainer(net_name, params, 1, "reg")
    c = tr._cached_prepare(cache, list(range(B)))
    assert c["plan"].family == _lib.STEP_FAMILY_AGGREGATE and c["plan"].from_memory, (net_name, n_feat, n_nodes)
    loss = tr.train_step_cached(cache, list(range(B)), apply_adam=False)
    torch.cuda.synchronize()
    assert tr.faults() == 0
    stats = new_stats()
    check_step("%s %d nodes F=%d, cached, from memory" % (net_name, n_nodes, n_feat), lazy, float(loss), tr.last_pred.cpu().numpy(),
               {k: p.grad.detach().cpu().numpy() for k, p in net.named_parameters()}, ref_loss, ref_pred.numpy(),
               {k: v.numpy() for k, v in ref_grads.items()}, stats)
    assert_arbiter_rate(stats, net_name)
    pc = tr.predict_cached(cache, list(range(B)))
    check("cached inference", pc.cpu().numpy(), ref_pred.numpy(), lazy.pred, new_stats())
    # the same mini-batch handed to the trainer as a Batch (no workspace given): the same launches, the same bits
    net2, tr2 = _trainer(net_name, params, 1, "reg")
    batch = batch_cpu.clone().to(dev)
    loss2 = tr2.compute_gradients(batch)
    torch.cuda.synchronize()
    assert float(loss2) == float(loss)
    for k, p in net2.named_parameters():
        np.testing.assert_array_equal(p.grad.detach().cpu().numpy(), net.get_parameter(k).grad.detach().cpu().numpy(), err_msg=k)
    np.testing.assert_array_equal(tr2.predict(batch).cpu().numpy(), pc.cpu().numpy())
