"""drgnn_train_epoch (native epoch loop over the resident set) against stepping the same mini-batches one by one
through FusedTrainer.train_step on host-collated batches.  Shared by the emulated (CPU) and the MI355X test."""
import copy

import numpy as np
import torch

from deeprank_gnn_amd.data import Batch
from deeprank_gnn_amd.resident import ResidentGraphSet
from deeprank_gnn_amd.trainer import FusedTrainer


def check_epoch(Net, graphs, n_feat, task, device, batch_size, api=None, epochs=2, seed=0, exact=True, cached=False):
    """``cached``: the native loop steps the mini-batches out of the set's topology cache (built once, no builder in the
    loop) -- must give the SAME bits as rebuilding every mini-batch's topology."""
    torch.manual_seed(seed)
    n_out = 1 if task == "reg" else 3
    net = Net(n_feat, n_out, 1).to(device)
    if hasattr(net, "dropout"):
        net.dropout = 0.4                       # the dropout stream follows the step counter in both paths
    rs = ResidentGraphSet(graphs, device, api=api)
    if task == "class":
        labels = torch.arange(len(graphs)) % n_out
        rs.set_targets(labels)
        graphs = [copy.copy(g) for g in graphs]
        for g, lab in zip(graphs, labels.tolist()):
            g.y = torch.tensor([lab])
    tr_a = FusedTrainer(net, lr=1e-2, task=task, api=api, seed=7)
    tr_b = FusedTrainer(copy.deepcopy(net), lr=1e-2, task=task, api=api, seed=7)
    rng = np.random.default_rng(seed)
    held_loss = tr_a.loss
    for _ in range(epochs):
        order = rng.permutation(len(graphs)).tolist()
        got = tr_a.train_epoch(rs, order, batch_size, cached=cached)
        assert got is not None, "the native epoch loop refused a configuration that fits"
        losses, pred = got
        want_l, want_p = [], []
        for lo in range(0, len(order), batch_size):
            b = Batch.from_data_list([graphs[i] for i in order[lo:lo + batch_size]]).to(device)
            want_l.append(float(tr_b.train_step(b)))
            want_p.append(tr_b.last_pred.detach().cpu().clone())
        want_p = torch.cat(want_p)
        # trainer.loss after an epoch: the LAST mini-batch's loss, written by that mini-batch's update launch itself
        # (drgnn_epoch_plan.last_loss: no copy, no lazy state) into the one buffer the single-step launches write -- a reference
        # taken BEFORE the epoch sees it (recorded hipGraphs and callers keep the buffer's address)
        buf = tr_a.loss
        assert buf.data_ptr() == tr_a._loss_buf.data_ptr() and buf is held_loss
        assert float(held_loss) == float(losses[-1])
        if exact:
            assert losses.cpu().tolist() == want_l
            assert torch.equal(pred.cpu(), want_p)
        else:
            np.testing.assert_allclose(losses.cpu().numpy(), want_l, rtol=1e-5)
            np.testing.assert_allclose(pred.cpu().numpy(), want_p.numpy(), rtol=1e-4, atol=1e-5)
    # inference pass over a permutation (dropout off): native loop vs predict() on host-collated mini-batches
    order = rng.permutation(len(graphs)).tolist()
    got = tr_a.predict_epoch(rs, order, batch_size, cached=cached)
    assert got is not None
    want = torch.cat([tr_b.predict(Batch.from_data_list([graphs[i] for i in order[lo:lo + batch_size]]).to(device)).cpu()
                      for lo in range(0, len(order), batch_size)])
    if exact:
        assert torch.equal(got.cpu(), want)
    else:
        np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    assert int(tr_a.step) == int(tr_b.step) == epochs * ((len(graphs) + batch_size - 1) // batch_size)
    if exact:
        assert torch.equal(tr_a.flat_p.cpu(), tr_b.flat_p.cpu())
        assert torch.equal(tr_a.exp_avg_sq.cpu(), tr_b.exp_avg_sq.cpu())
    else:
        np.testing.assert_allclose(tr_a.flat_p.cpu().numpy(), tr_b.flat_p.cpu().numpy(), rtol=1e-4, atol=1e-6)
    if cached:
        # single steps out of the cache == single steps on the collated mini-batch
        ids = order[:batch_size]
        b = Batch.from_data_list([graphs[i] for i in ids]).to(device)
        la = float(tr_a.train_step_cached(rs.topology_cache(need_weights=True), ids))
        lb = float(tr_b.train_step(b))
        pa = tr_a.predict_cached(rs.topology_cache(need_weights=True), ids).cpu()
        pb = tr_b.predict(b).cpu()
        if exact:
            assert la == lb
            assert torch.equal(tr_a.flat_p.cpu(), tr_b.flat_p.cpu())
            assert torch.equal(pa, pb)
        else:
            np.testing.assert_allclose(la, lb, rtol=1e-5)
            np.testing.assert_allclose(pa.numpy(), pb.numpy(), rtol=1e-4, atol=1e-5)
    return tr_a
