"""graclus / normalized_cut / max_pool (README custom-net recipe) against oracle/graclus_ref.py and structural
properties.  Shared by the emulated (CPU) and the MI355X test."""
import numpy as np
import torch

import deeprank_gnn_amd.community_pooling as cp
from deeprank_gnn_amd.data import Batch, Data
from oracle import graclus_ref


def random_graphs(seed, count=5):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(count):
        n = int(rng.integers(1, 40))
        m = int(rng.integers(0, 3 * n))
        pairs = rng.integers(0, n, size=(m, 2))
        if k == 1 and m:
            pairs[0] = [0, 0]                                       # a self loop
        ei = np.concatenate([pairs, pairs[:, ::-1]], axis=0).T      # symmetric, like the loader's edge_index
        g = Data(x=torch.from_numpy(rng.standard_normal((n, 6)).astype(np.float32)),
                 edge_index=torch.from_numpy(np.ascontiguousarray(ei)).long(),
                 edge_attr=torch.from_numpy(rng.uniform(0.1, 2.0, (2 * m, 1)).astype(np.float32)),
                 pos=torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32)))
        out.append(g)
    return out


def check_graclus(device, api=None, seed=0):
    cp._API = api
    try:
        rng = np.random.default_rng(seed)
        graphs = random_graphs(seed)
        for weighted in (False, True):
            for use_perm in (False, True):
                # one graph at a time against the oracle
                for g in graphs:
                    n = g.x.size(0)
                    perm = rng.permutation(n) if use_perm else None
                    w = g.edge_attr.reshape(-1) if weighted else None
                    want = graclus_ref.graclus(g.edge_index.numpy(), None if w is None else w.numpy(), n, perm)
                    got = cp.graclus(g.edge_index.to(device), None if w is None else w.to(device), n,
                                     perm=None if perm is None else torch.from_numpy(perm).to(device))
                    assert got.cpu().tolist() == want.tolist()
                    _properties(g.edge_index.numpy(), want, n)
                # the block-diagonal batch, split by `batch`: per-graph labels shifted by the node offsets
                b = Batch.from_data_list(graphs)
                perms = [rng.permutation(g.x.size(0)) for g in graphs] if use_perm else None
                w = b.edge_attr.reshape(-1) if weighted else None
                got = cp.graclus(b.edge_index.to(device), None if w is None else w.to(device), perm=None if perms is None
                                 else torch.from_numpy(np.concatenate(perms)).to(device), batch=b.batch.to(device))
                want, off = [], 0
                for i, g in enumerate(graphs):
                    n = g.x.size(0)
                    lab = graclus_ref.graclus(g.edge_index.numpy(), g.edge_attr.reshape(-1).numpy() if weighted else None,
                                              n, None if perms is None else perms[i])
                    want += (lab + off).tolist()
                    off += n
                assert got.cpu().tolist() == want
        # normalized_cut
        g = graphs[0]
        nc = cp.normalized_cut(g.edge_index.to(device), g.edge_attr.to(device), g.x.size(0))
        np.testing.assert_allclose(nc.cpu().numpy(), graclus_ref.normalized_cut(g.edge_index.numpy(), g.edge_attr.numpy(),
                                                                              g.x.size(0)), rtol=1e-6)
    finally:
        cp._API = None


def _properties(edge_index, labels, n):
    """A maximal matching: clusters of size <= 2, pairs are adjacent, no edge joins two singletons."""
    members = {}
    for i, l in enumerate(labels.tolist()):
        members.setdefault(l, []).append(i)
    adj = {(int(a), int(b)) for a, b in zip(edge_index[0], edge_index[1]) if a != b}
    single = set()
    for l, m in members.items():
        assert len(m) <= 2 and l == min(m)
        if len(m) == 2:
            assert (m[0], m[1]) in adj or (m[1], m[0]) in adj
        else:
            single.add(m[0])
    for a, b in adj:
        assert not (a in single and b in single)


def check_custom_net_recipe(device, api=None):
    """The README pipeline: normalized_cut -> graclus -> max_pool -> ... -> graclus -> max_pool_x -> scatter_mean, on a
    batch; max_pool against a plain torch restatement of the PyG pooling (unique / amax / coalesce-add) on the same labels."""
    cp._API = api
    try:
        graphs = random_graphs(3, count=4)
        b = Batch.from_data_list(graphs).to(device)
        w = cp.normalized_cut(b.edge_index, b.edge_attr, b.x.size(0))
        cluster = cp.graclus(b.edge_index, w, b.x.size(0), batch=b.batch)
        pooled = cp.max_pool(cluster, b)
        # oracle pooling of the same labels
        hb = Batch.from_data_list(graphs)
        uniq, inv = torch.unique(cluster.cpu(), return_inverse=True)
        x_ref = torch.full((uniq.numel(), hb.x.size(1)), float("-inf"))
        x_ref = x_ref.scatter_reduce(0, inv[:, None].expand(-1, hb.x.size(1)), hb.x, reduce="amax")
        assert torch.equal(pooled.x.cpu(), x_ref)
        r, c = inv[hb.edge_index[0]], inv[hb.edge_index[1]]
        keep = r != c
        key = r[keep] * uniq.numel() + c[keep]
        ukey, kinv = torch.unique(key, return_inverse=True)
        attr = torch.zeros(ukey.numel()).index_add_(0, kinv, hb.edge_attr.reshape(-1)[keep])
        assert torch.equal(pooled.edge_index.cpu(), torch.stack([ukey // uniq.numel(), ukey % uniq.numel()]))
        np.testing.assert_allclose(pooled.edge_attr.cpu().reshape(-1).numpy(), attr.numpy(), rtol=1e-6)
        first = torch.zeros(uniq.numel(), dtype=torch.long).scatter_reduce(0, inv, hb.batch, reduce="amin", include_self=False)
        assert torch.equal(pooled.batch.cpu(), first)
        # second level
        w2 = cp.normalized_cut(pooled.edge_index, pooled.edge_attr, pooled.x.size(0))
        cluster2 = cp.graclus(pooled.edge_index, w2, pooled.x.size(0), batch=pooled.batch)
        x2, batch2 = cp.max_pool_x(cluster2, pooled.x, pooled.batch)
        out = cp.scatter_mean(x2, batch2, dim=0)
        assert out.shape == (4, 6) and torch.isfinite(out).all()
    finally:
        cp._API = None
