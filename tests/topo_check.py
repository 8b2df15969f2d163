"""Expands a Topology workspace into plain global tensors and compares with the oracle."""
import numpy as np
import torch

from oracle import cpu_ref


def expand(topo):
    a = {k: topo.array(k).cpu().numpy() for k in
         ("NPTR", "EPTR", "ROWPTR0", "COL0", "EID0", "COLPTR0", "ROWIDX0", "TSLOT0", "CL0", "NC0",
          "MPTR0", "MEM0", "ROWPTR1", "COL1", "NE1", "COLPTR1", "ROWIDX1", "TSLOT1", "CL1", "NC1",
          "MPTR1", "MEM1", "HORD", "HMP0", "HSPLIT")}
    if topo.ws_f32 is not None:
        a["W0"] = topo.weights("W0").cpu().numpy()
        a["W1"] = topo.weights("W1").cpu().numpy()
    return a


def check_against_oracle(topo, batch, level1=True, weights=True):
    """A workspace built with TOPO_LEAN (flags & 2) holds no CSC0, no depth-0 member lists and -- without edge weights -- no
    TSLOT1: those arrays are not compared."""
    a = expand(topo)
    lean = bool(int(getattr(topo, "flags", 0)) & 2)
    B = topo.n_graphs
    nptr, eptr = a["NPTR"], a["EPTR"]
    ei = batch.edge_index.cpu()
    bvec = batch.batch.cpu()
    ea = None if getattr(batch, "edge_attr", None) is None else batch.edge_attr.cpu().reshape(-1)
    if not weights or topo.ws_f32 is None:      # structure-only build: no W0 / W1 to compare
        ea = None
    # offsets
    counts = torch.bincount(bvec, minlength=B).numpy() if bvec.numel() else np.zeros(B, int)
    np.testing.assert_array_equal(nptr[:B + 1], np.concatenate([[0], np.cumsum(counts)]))
    ecounts = torch.bincount(bvec[ei[0]], minlength=B).numpy() if ei.numel() else np.zeros(B, int)
    np.testing.assert_array_equal(eptr[:B + 1], np.concatenate([[0], np.cumsum(ecounts)]))

    # oracle global quantities
    cl0 = cpu_ref.get_preloaded_cluster(batch.cluster0.cpu().clone(), bvec)
    cons0, perm0 = cpu_ref.consecutive_cluster(cl0)
    pei, pea = cpu_ref.pool_edge(cons0, ei, None if ea is None else ea.view(-1, 1))
    pbatch = bvec[perm0]
    c0counts = torch.bincount(pbatch, minlength=B).numpy() if pbatch.numel() else np.zeros(B, int)
    np.testing.assert_array_equal(a["NC0"][:B], c0counts)
    cptr0 = np.concatenate([[0], np.cumsum(c0counts)])

    got_rows, got_cols, got_w = [], [], []
    for g in range(B):
        n0, n1, e0, e1 = nptr[g], nptr[g + 1], eptr[g], eptr[g + 1]
        N, E = n1 - n0, e1 - e0
        rb = n0 + g
        # ---- CSR0: neighbours of node i in edge-id order
        rp = a["ROWPTR0"][rb:rb + N + 1]
        assert rp[0] == 0 and rp[-1] == E
        rows = (ei[0, e0:e1] - n0).numpy()
        cols = (ei[1, e0:e1] - n0).numpy()
        eid = a["EID0"][e0:e1]
        order = np.lexsort((np.arange(E), rows))          # stable by row, then edge id
        np.testing.assert_array_equal(eid, order)
        np.testing.assert_array_equal(a["COL0"][e0:e1], cols[order])
        np.testing.assert_array_equal(np.repeat(np.arange(N), np.diff(rp)), rows[order])
        if ea is not None:
            np.testing.assert_array_equal(a["W0"][e0:e1], ea[e0:e1].numpy()[order])
        # ---- CSC0: entries of column j ordered by edge id
        if not lean:
            cp = a["COLPTR0"][rb:rb + N + 1]
            eorder = np.lexsort((np.arange(E), cols))
            slot_of_edge = np.empty(E, dtype=np.int64)
            slot_of_edge[order] = np.arange(E)
            np.testing.assert_array_equal(a["TSLOT0"][e0:e1], slot_of_edge[eorder])
            np.testing.assert_array_equal(a["ROWIDX0"][e0:e1], rows[eorder])
            np.testing.assert_array_equal(np.repeat(np.arange(N), np.diff(cp)), cols[eorder])
        # ---- depth-0 clusters
        C = a["NC0"][g]
        np.testing.assert_array_equal(a["CL0"][n0:n1] + cptr0[g], cons0[n0:n1].numpy())
        if not lean:
            mp = a["MPTR0"][rb:rb + C + 1]
            mem = a["MEM0"][n0:n1]
            assert mp[0] == 0 and mp[-1] == N
            loc = a["CL0"][n0:n1]
            np.testing.assert_array_equal(mem, np.lexsort((np.arange(N), loc)))
            np.testing.assert_array_equal(np.repeat(np.arange(C), np.diff(mp)), loc[mem])
        # ---- pooled graph
        E1 = a["NE1"][g]
        rp1 = a["ROWPTR1"][rb:rb + C + 1]
        assert rp1[0] == 0 and rp1[-1] == E1
        r1 = np.repeat(np.arange(C), np.diff(rp1))
        c1 = a["COL1"][e0:e0 + E1]
        got_rows.append(r1 + cptr0[g])
        got_cols.append(c1 + cptr0[g])
        if ea is not None:
            got_w.append(a["W1"][e0:e0 + E1])
        cp1 = a["COLPTR1"][rb:rb + C + 1]
        corder = np.lexsort((np.arange(E1), c1))
        if not lean or ea is not None:
            np.testing.assert_array_equal(a["TSLOT1"][e0:e0 + E1], corder)
        np.testing.assert_array_equal(a["ROWIDX1"][e0:e0 + E1], r1[corder])
        np.testing.assert_array_equal(np.repeat(np.arange(C), np.diff(cp1)), c1[corder])
    got_rows = np.concatenate(got_rows) if got_rows else np.zeros(0, int)
    got_cols = np.concatenate(got_cols) if got_cols else np.zeros(0, int)
    np.testing.assert_array_equal(np.stack([got_rows, got_cols]), pei.numpy().reshape(2, -1))
    if ea is not None and pea is not None:
        # (the builder adds the raw weights of a pooled edge EXACTLY (64-bit fixed point); the oracle's scatter-add runs in float32:
        # the difference is the oracle's own rounding, ~sqrt(addends) ulp)
        np.testing.assert_allclose(np.concatenate(got_w), pea.numpy().reshape(-1), rtol=2e-5, atol=1e-6)

    if level1 and getattr(batch, "cluster1", None) is not None:
        cl1 = cpu_ref.get_preloaded_cluster(batch.cluster1.cpu().clone(), pbatch)
        cons1, perm1 = cpu_ref.consecutive_cluster(cl1)
        b2 = pbatch[perm1]
        c1counts = torch.bincount(b2, minlength=B).numpy() if b2.numel() else np.zeros(B, int)
        np.testing.assert_array_equal(a["NC1"][:B], c1counts)
        cptr1 = np.concatenate([[0], np.cumsum(c1counts)])
        for g in range(B):
            n0 = nptr[g]
            C, C1 = a["NC0"][g], a["NC1"][g]
            rb = n0 + g
            loc = a["CL1"][n0:n0 + C]
            np.testing.assert_array_equal(loc + cptr1[g], cons1[cptr0[g]:cptr0[g + 1]].numpy())
            mp = a["MPTR1"][rb:rb + C1 + 1]
            mem = a["MEM1"][n0:n0 + C]
            assert mp[0] == 0 and mp[-1] == C
            np.testing.assert_array_equal(mem, np.lexsort((np.arange(C), loc)))
            # ---- hierarchical node order: nodes by (depth-1 cluster, depth-0 cluster, node id); the depth-0 clusters in
            # MEM1 order own consecutive runs of positions (HMP0); split = the prefix of depth-1 clusters closest to N / 2
            N = nptr[g + 1] - n0
            if len(loc) == C and C > 0 and (int(getattr(topo, "flags", 1)) & 1):      # (built with DRGNN_TOPO_HIER)
                cl0 = a["CL0"][n0:n0 + N]
                want = np.lexsort((np.arange(N), cl0, loc[cl0]))
                np.testing.assert_array_equal(a["HORD"][n0:n0 + N], want)
                sizes = np.bincount(cl0, minlength=C)
                hmp = np.concatenate([[0], np.cumsum(sizes[mem])])
                np.testing.assert_array_equal(a["HMP0"][rb:rb + C + 1], hmp)
                pos = hmp[mp]                                   # first position of every depth-1 cluster (+ the end)
                k = int(np.argmin(np.abs(2 * pos - N)))
                np.testing.assert_array_equal(a["HSPLIT"][4 * g:4 * g + 4], [k, mp[k], hmp[mp[k]], C1])
    return a


def check_tiles(topo, batch, weights):
    """Level-0 aggregation tiles (TOPO_TILES) against a plain restatement: S_i = sum over the edges (i, j) of [w_e] x_j, D, C."""
    assert topo.tiles is not None and (int(topo.flags) & 4)
    S, D, C = (t.cpu().numpy() for t in topo.tile_arrays())
    x = batch.x.cpu().numpy().astype(np.float64)
    ei = batch.edge_index.cpu().numpy()
    n = x.shape[0]
    w = batch.edge_attr.cpu().numpy().reshape(-1).astype(np.float64) if weights else np.ones(ei.shape[1])
    want = np.zeros_like(x)
    np.add.at(want, ei[0], w[:, None] * x[ei[1]])
    deg = np.bincount(ei[0], minlength=n).astype(np.float64)
    np.testing.assert_allclose(S, want, rtol=1e-5, atol=1e-5)
    if weights:
        d = 1.0 / np.maximum(deg, 1.0)
        asum = np.bincount(ei[0], weights=w, minlength=n)
        np.testing.assert_allclose(D, d, rtol=1e-6)
        np.testing.assert_allclose(C, asum * d, rtol=1e-5, atol=1e-6)
    else:
        np.testing.assert_allclose(D, np.where(deg > 0, 1.0 / np.maximum(deg, 1.0), 0.0), rtol=1e-6)
        np.testing.assert_array_equal(C, np.ones(n, dtype=np.float32))
    F = x.shape[1]
    if F % 4:
        # the padded copy of the node features (what sGAT / FoutNet multiply with their self weights when rows are not multiples of
        # 16 bytes): behind D and C at a multiple of 4 floats -- 16-byte aligned rows whatever the parity of n (include/drgnn.h)
        TF = (F + 3) // 4 * 4
        off = n * TF + (2 * n + 3) // 4 * 4
        assert (topo.tiles.data_ptr() + 4 * off) % 16 == 0
        X = topo.tiles[off:off + n * TF].view(n, TF).cpu().numpy()
        np.testing.assert_array_equal(X[:, :F], batch.x.cpu().numpy())
        np.testing.assert_array_equal(X[:, F:], np.zeros((n, TF - F), dtype=np.float32))
    # IHORD is the inverse of HORD, graph by graph
    nptr = topo.array("NPTR").cpu().numpy()
    hord, ihord = topo.array("HORD").cpu().numpy(), topo.array("IHORD").cpu().numpy()
    for g in range(topo.n_graphs):
        h = hord[nptr[g]:nptr[g + 1]]
        np.testing.assert_array_equal(ihord[nptr[g]:nptr[g + 1]][h], np.arange(len(h)))
