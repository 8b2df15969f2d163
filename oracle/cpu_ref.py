"""CPU oracle for the Deeprank-GNN message-passing hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module, and only as the checker / timed CPU baseline -- never as part of
the product path (``deeprank_gnn_amd`` never imports it and raises if the HIP library
is missing).

What it is: a plain PyTorch (fp32, CPU) restatement of the reference algorithm for the
path, keeping the reference's algorithmic SHAPE (edge-level GEMMs, the dead attention
branch of GINetConvLayer, the per-graph Python loop of get_preloaded_cluster, the
per-node Python loop of FoutLayer) so that timing it is an honest stand-in for "the
reference on CPU".  Each function cites the reference file:line it follows.

The arithmetic of the un-vendored third-party ops the reference calls (torch_scatter,
torch_sparse.coalesce, torch_geometric pooling helpers; reference setup.py:42-47 lists
them without versions, CI resolves the torch-1.8.0 wheels, build.yml:36-48) is restated
from their published semantics, see SURVEY.md Appendix A.

Pinning ("how do we know the oracle is right"): tests/test_oracle_golden.py checks it
against tests/golden/*.npz, which were produced in the build container by running the
reference's own, unmodified layer code (tests/golden/gen/make_golden.py).  The
reference's own tests hold no numerical assertions for this path; the
fixture's stored clustering/mcl/depth_1 pins the index semantics of pooling.
"""
import math
import types

import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------- #
# third-party op semantics
# --------------------------------------------------------------------------- #


def scatter_sum(src, index, dim_size=None):
    """torch_scatter.scatter_sum along dim 0: out.scatter_add_(0, index, src)."""
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    out = torch.zeros((dim_size,) + tuple(src.shape[1:]), dtype=src.dtype)
    return out.scatter_add_(0, idx, src)


def scatter_mean(src, index, dim_size=None):
    """torch_scatter.scatter_mean: sum / clamp(count, min=1); empty segment -> 0."""
    total = scatter_sum(src, index, dim_size)
    count = scatter_sum(torch.ones(index.numel(), dtype=src.dtype), index, total.size(0))
    count = count.clamp(min=1)
    return total / count.view(-1, *([1] * (src.dim() - 1)))


class _SegMax(torch.autograd.Function):
    """torch_scatter.scatter_max, CPU tie rule: sources are visited in order and replace
    the running maximum on strict '>', i.e. the FIRST occurrence of the maximum wins.
    Empty segment -> value 0 and arg = number of sources."""

    @staticmethod
    def forward(ctx, src, index, dim_size):
        n, h = src.shape
        idx = index.view(-1, 1).expand(n, h)
        # the CPU kernel compares with '>' : a NaN source never replaces the running maximum
        clean = torch.where(torch.isnan(src), torch.full_like(src, -math.inf), src)
        best = torch.full((dim_size, h), -math.inf, dtype=src.dtype)
        best = best.scatter_reduce(0, idx, clean, reduce="amax", include_self=True)
        pos = torch.arange(n).view(-1, 1).expand(n, h)
        is_best = (clean == best.gather(0, idx)) & (clean > -math.inf)
        cand = torch.where(is_best, pos, torch.full_like(pos, n))
        arg = torch.full((dim_size, h), n, dtype=torch.long)
        arg = arg.scatter_reduce(0, idx, cand, reduce="amin", include_self=True)
        out = torch.where(arg == n, torch.zeros_like(best), best)
        ctx.save_for_backward(arg)
        ctx.n = n
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, grad_out, _):
        (arg,) = ctx.saved_tensors
        grad = torch.zeros(ctx.n + 1, grad_out.size(1), dtype=grad_out.dtype)
        grad.scatter_add_(0, arg, grad_out)
        return grad[: ctx.n], None, None


def scatter_max(src, index, dim_size=None):
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() else 0
    return _SegMax.apply(src, index, dim_size)


def consecutive_cluster(ids):
    """torch_geometric consecutive_cluster: id -> rank among the sorted distinct ids;
    perm[c] = index of one member of new cluster c."""
    uniq, rank = torch.unique(ids, sorted=True, return_inverse=True)
    perm = torch.empty(uniq.numel(), dtype=torch.long)
    perm.scatter_(0, rank, torch.arange(ids.numel()))
    return rank, perm


def pool_edge(cluster, edge_index, edge_attr=None):
    """torch_geometric pool_edge: relabel both endpoints, drop self loops, then
    torch_sparse.coalesce (sort by row*n+col, merge duplicates, attributes SUMMED)."""
    n = cluster.numel()
    row = cluster[edge_index[0]]
    col = cluster[edge_index[1]]
    keep = row != col
    row, col = row[keep], col[keep]
    attr = None if edge_attr is None else edge_attr[keep]
    if row.numel() == 0:
        return torch.stack([row, col]), attr
    key = row * n + col
    order = torch.argsort(key, stable=True)
    key = key[order]
    uniq, slot = torch.unique_consecutive(key, return_inverse=True)
    new_row = torch.div(uniq, n, rounding_mode="floor")
    new_index = torch.stack([new_row, uniq - new_row * n])
    if attr is not None:
        attr = scatter_sum(attr[order], slot, uniq.numel())
    return new_index, attr


def max_pool_x(cluster, x, batch):
    """torch_geometric max_pool_x: consecutive_cluster -> per-cluster max -> batch[perm]."""
    cluster, perm = consecutive_cluster(cluster)
    pooled, _ = scatter_max(x, cluster)
    return pooled, batch[perm]


# --------------------------------------------------------------------------- #
# reference functions on the path
# --------------------------------------------------------------------------- #


def get_preloaded_cluster(cluster, batch):
    """reference community_pooling.py:25-30 -- make per-graph cluster ids globally
    unique with a running offset; IN PLACE; one masked update per graph."""
    n_graph = int(batch.max()) + 1
    for g in range(1, n_graph):
        prev_top = cluster[batch == g - 1].max()
        cluster[batch == g] += prev_top + 1
    return cluster


def community_pooling(cluster, data):
    """reference community_pooling.py:161-251 (the ``Batch`` branch, :222-234)."""
    cluster, perm = consecutive_cluster(cluster)
    x, arg = scatter_max(data.x, cluster)
    edge_index, edge_attr = pool_edge(cluster, data.edge_index, data.edge_attr)
    out = types.SimpleNamespace(x=x, edge_index=edge_index, edge_attr=edge_attr,
                                batch=data.batch[perm], argmax=arg, cluster=cluster)
    if getattr(data, "internal_edge_index", None) is not None:
        out.internal_edge_index, out.internal_edge_attr = pool_edge(
            cluster, data.internal_edge_index, getattr(data, "internal_edge_attr", None))
    if getattr(data, "pos", None) is not None:
        out.pos = scatter_mean(data.pos, cluster)
    out.cluster0 = getattr(data, "cluster0", None)
    out.cluster1 = getattr(data, "cluster1", None)
    return out


def ginet_conv(x, edge_index, edge_attr, w_fc, w_edge, w_att, b_fc=None, b_edge=None, b_att=None):
    """reference ginet.py:50-73 (GINetConvLayer.forward), op for op, including the
    attention branch whose softmax over a size-1 axis is identically 1.
    ``b_*``: the three Linear biases of GINetConvLayer(bias=True) (ginet.py:26-37; the nets build bias=False)."""
    row, col = edge_index[0], edge_index[1]
    if edge_attr.dim() == 1:
        edge_attr = edge_attr.unsqueeze(-1)
    msg_col = F.linear(x[col], w_fc, b_fc)
    msg_row = F.linear(x[row], w_fc, b_fc)
    edge_term = F.linear(edge_attr, w_edge, b_edge)
    score = F.linear(torch.cat([msg_row, msg_col, edge_term], dim=1), w_att, b_att)
    score = F.softmax(F.leaky_relu(score), dim=1)
    return scatter_sum(score * msg_col, row, x.size(0))


def sgat_conv(x, edge_index, edge_attr, weight, bias, undirected=True):
    """reference sGAT.py:62-93 (sGraphAttentionLayer.forward).  ``undirected=False`` adds the second
    ``scatter_mean(alpha, col, dim=0, out=out)`` (:86-87): torch_scatter adds the column sums INTO the row means and then
    divides the WHOLE buffer by the column counts (clamped to 1) -- SURVEY Appendix A."""
    row, col = edge_index[0], edge_index[1]
    if edge_attr.dim() == 1:
        edge_attr = edge_attr.unsqueeze(-1)
    pair = torch.cat([x[row], x[col]], dim=-1)
    msg = edge_attr * torch.mm(pair, weight)
    out = scatter_mean(msg, row, x.size(0))
    if not undirected:
        n = x.size(0)
        count = scatter_sum(torch.ones(col.numel(), dtype=msg.dtype), col, n).clamp(min=1)
        out = (out + scatter_sum(msg, col, n)) / count.view(-1, 1)
    return out if bias is None else out + bias


def fout_conv(x, edge_index, w_center, w_neigh, bias, looped=True):
    """reference foutnet.py:56-82 (FoutLayer.forward).  ``looped=True`` keeps the
    reference's per-node Python loop (:71-73; a node without out-edges yields NaN);
    ``looped=False`` is the arithmetically equivalent vectorised form used only as the
    second CPU-baseline line."""
    n = x.size(0)
    center = torch.mm(x, w_center)
    neigh = torch.mm(x, w_neigh)
    if looped:
        rows = []
        for node in range(n):
            nbr = edge_index[1, edge_index[0] == node]
            rows.append(neigh[nbr].mean(dim=0))
        gathered = torch.stack(rows) if rows else neigh.new_zeros(0, neigh.size(1))
    else:
        row, col = edge_index[0], edge_index[1]
        total = scatter_sum(neigh[col], row, n)
        deg = scatter_sum(torch.ones(row.numel(), dtype=x.dtype), row, n)
        gathered = total / deg.view(-1, 1)          # 0/0 -> NaN like mean of empty
    out = center + gathered
    return out if bias is None else out + bias


def _as_ns(data):
    ns = types.SimpleNamespace()
    for k in ("x", "edge_index", "edge_attr", "batch", "pos", "cluster0", "cluster1",
              "internal_edge_index", "internal_edge_attr"):
        v = getattr(data, k, None)
        setattr(ns, k, v.clone() if torch.is_tensor(v) else v)
    return ns


def _branch(data, conv1, conv2, trace, tag):
    """conv -> relu -> community_pooling -> conv -> relu -> max_pool_x
    (reference ginet.py:103-114, sGAT.py:119-130, foutnet.py:108-117)."""
    z1 = conv1(data.x, data.edge_index, data.edge_attr)
    data.x = F.relu(z1)
    cl0 = get_preloaded_cluster(data.cluster0, data.batch)
    pooled = community_pooling(cl0, data)
    xp = pooled.x
    z2 = conv2(xp, pooled.edge_index, pooled.edge_attr)
    pooled.x = F.relu(z2)
    cl1 = get_preloaded_cluster(pooled.cluster1, pooled.batch)
    x2, batch2 = max_pool_x(cl1, pooled.x, pooled.batch)
    if trace is not None:
        trace[tag + "z1"] = z1
        trace[tag + "cluster0"] = pooled.cluster
        trace[tag + "xp"] = xp
        trace[tag + "arg0"] = pooled.argmax
        trace[tag + "pool_edge_index"] = pooled.edge_index
        trace[tag + "pool_edge_attr"] = pooled.edge_attr
        trace[tag + "pool_batch"] = pooled.batch
        trace[tag + "z2"] = z2
        trace[tag + "cluster1"] = consecutive_cluster(cl1)[0]
        trace[tag + "x2"] = x2
        trace[tag + "batch2"] = batch2
    return x2, batch2


def ginet_forward(params, data, dropout=0.0, training=False, trace=None, drop_mask=None):
    """reference ginet.py:99-141 (GINet.forward).  ``params`` maps state_dict names to
    tensors.  The second branch convolves over the SAME edge_index.
    ``drop_mask`` ([B, 128] of 0 / 1): F.dropout's arithmetic (ginet.py:138) with the Bernoulli
    draw given instead of sampled -- ``hid * mask / (1 - p)`` -- so that a dropout-on launch of
    the kernels (drgnn_head_desc.drop_mask) has a deterministic reference."""
    def conv(prefix):
        return lambda x, ei, ea: ginet_conv(x, ei, ea, params[prefix + ".fc.weight"],
                                            params[prefix + ".fc_edge_attr.weight"],
                                            params[prefix + ".fc_attention.weight"])
    d_a, d_b = _as_ns(data), _as_ns(data)
    xa, ba = _branch(d_a, conv("conv1"), conv("conv2"), trace, "a.")
    xb, bb = _branch(d_b, conv("conv1_ext"), conv("conv2_ext"), trace, "b.")
    ra = scatter_mean(xa, ba)
    rb = scatter_mean(xb, bb)
    feat = torch.cat([ra, rb], dim=1)
    if trace is not None:
        trace["readout"] = feat
    hid = F.relu(F.linear(feat, params["fc1.weight"], params["fc1.bias"]))
    if drop_mask is not None:
        hid = hid * drop_mask.to(hid.dtype) / (1.0 - dropout)
    else:
        hid = F.dropout(hid, dropout, training=training)
    return F.linear(hid, params["fc2.weight"], params["fc2.bias"])


def sgat_forward(params, data, trace=None):
    """reference sGAT.py:114-138 (sGAT.forward); act = relu (the Tanhshrink on :116 is
    overwritten on :117)."""
    def conv(prefix):
        return lambda x, ei, ea: sgat_conv(x, ei, ea, params[prefix + ".weight"],
                                           params[prefix + ".bias"])
    d = _as_ns(data)
    x2, b2 = _branch(d, conv("conv1"), conv("conv2"), trace, "a.")
    feat = scatter_mean(x2, b2)
    if trace is not None:
        trace["readout"] = feat
    hid = F.relu(F.linear(feat, params["fc1.weight"], params["fc1.bias"]))
    return F.linear(hid, params["fc2.weight"], params["fc2.bias"])


def fout_forward(params, data, looped=True, trace=None):
    """reference foutnet.py:103-125 (FoutNet.forward)."""
    def conv(prefix):
        return lambda x, ei, ea: fout_conv(x, ei, params[prefix + ".Wc"],
                                           params[prefix + ".Wn"], params[prefix + ".bias"],
                                           looped=looped)
    d = _as_ns(data)
    x2, b2 = _branch(d, conv("conv1"), conv("conv2"), trace, "a.")
    feat = scatter_mean(x2, b2)
    if trace is not None:
        trace["readout"] = feat
    hid = F.relu(F.linear(feat, params["fc1.weight"], params["fc1.bias"]))
    return F.linear(hid, params["fc2.weight"], params["fc2.bias"])


FORWARD = {"GINet": ginet_forward, "sGAT": sgat_forward, "FoutNet": fout_forward}


# --------------------------------------------------------------------------- #
# parameter construction with the reference's shapes and init rules
# --------------------------------------------------------------------------- #


def _uniform(shape, fan, gen):
    bound = 1.0 / math.sqrt(fan)
    return (torch.rand(shape, generator=gen) * 2.0 - 1.0) * bound


def init_params(net, n_feat, n_out=1, n_edge_feat=1, seed=0):
    """Random parameters with the reference's names, shapes and init bounds
    (ginet.py:35-48,87-95; sGAT.py:47-60,106-110; foutnet.py:40-54,95-99;
    nn.Linear default init for the FC head)."""
    gen = torch.Generator().manual_seed(seed)
    p = {}

    def linear(name, fin, fout):
        p[name + ".weight"] = _uniform((fout, fin), fin, gen)
        p[name + ".bias"] = _uniform((fout,), fin, gen)

    if net == "GINet":
        for name, fin, fout in (("conv1", n_feat, 16), ("conv2", 16, 32),
                                ("conv1_ext", n_feat, 16), ("conv2_ext", 16, 32)):
            p[name + ".fc.weight"] = _uniform((fout, fin), fin, gen)
            p[name + ".fc_edge_attr.weight"] = _uniform((n_edge_feat, n_edge_feat), fin, gen)
            p[name + ".fc_attention.weight"] = _uniform((1, 2 * fout + n_edge_feat), fin, gen)
        linear("fc1", 64, 128)
        linear("fc2", 128, n_out)
    elif net == "sGAT":
        for name, fin, fout in (("conv1", n_feat, 16), ("conv2", 16, 32)):
            p[name + ".weight"] = _uniform((2 * fin, fout), 2 * fin, gen)
            p[name + ".bias"] = _uniform((fout,), 2 * fin, gen)
        linear("fc1", 32, 64)
        linear("fc2", 64, n_out)
    elif net == "FoutNet":
        for name, fin, fout in (("conv1", n_feat, 16), ("conv2", 16, 32)):
            p[name + ".Wc"] = _uniform((fin, fout), fin, gen)
            p[name + ".Wn"] = _uniform((fin, fout), fin, gen)
            p[name + ".bias"] = _uniform((fout,), fin, gen)
        linear("fc1", 32, 64)
        linear("fc2", 64, n_out)
    else:
        raise ValueError(net)
    return p


def loss_and_grads(net, params, data, target, task="reg", **fw):
    """One training-step worth of math on CPU: forward, loss (MSE for regression,
    cross-entropy for classification: reference NeuralNet.py:239-263), backward.
    Returns (pred, loss, {name: grad})."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    pred = FORWARD[net](leaves, data, **fw)
    if task == "reg":
        loss = F.mse_loss(pred.reshape(-1), target)
    else:
        loss = F.cross_entropy(pred, target)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaves.items()}
    return pred.detach(), loss.detach(), grads
