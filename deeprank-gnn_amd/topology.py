"""Per-mini-batch topology workspace (CSR/CSC, consecutive clusters, pooled graph).

Host-side owner of the buffers that ``drgnn_topology_build`` fills (include/drgnn.h).
It stands in for what the reference recomputes inside every forward pass:
``get_preloaded_cluster`` (community_pooling.py:25-30), ``consecutive_cluster`` and
``pool_edge`` [torch_geometric] inside ``community_pooling`` (community_pooling.py:197-201),
and the COO edge scans of the conv layers.  Nothing here depends on learned parameters,
so one ``Topology`` serves both GINet branches, forward and backward.
"""
import torch

from . import _lib

__all__ = ["Topology"]


def _contig(t, dtype):
    if t is None:
        return None
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


class Topology(object):
    def __init__(self, api, n_nodes, n_edges, n_graphs, device, has_weights):
        self.api = api
        self.n_nodes, self.n_edges, self.n_graphs = int(n_nodes), int(n_edges), int(n_graphs)
        self.off_i32, self.off_f32 = api.topology_layout(self.n_nodes, self.n_edges, self.n_graphs)
        self.ws_i32 = torch.empty(max(self.off_i32[-1], 4), dtype=torch.int32, device=device)
        self.ws_f32 = (torch.empty(max(self.off_f32[-1], 4), dtype=torch.float32, device=device)
                       if has_weights else None)
        self.max_nodes = 0
        self.max_edges = 0
        self.max_c0 = 0
        self.host_node_ptr = self.host_edge_ptr = None      # host copies of the offsets, when derived here (from_batch)
        self.has_level1 = False
        self.flags = _lib.TOPO_HIER      # what the last build put into the workspace (drgnn_topology_request.flags)
        # level-0 aggregation tiles (TOPO_TILES): the node features they are formed from and the output buffer
        # ([S n x F | D n | C n], include/drgnn.h); None: not available for this batch
        # CONTRACT: with TOPO_TILES a build reads the VALUES of x (S = sums of x rows over the neighbours); a launch may use
        # the tiles only for that same tensor, unmodified since (FusedTrainer checks storage and version and forms the
        # tiles again otherwise).  Everything else in the workspace depends on the index tensors only.
        self.x = None
        self.tiles = None
        self._tiles_x_version = None
        self.n_feat = 0
        self._finalized = False

    # ---------------------------------------------------------------------------
    @classmethod
    def from_batch(cls, data, api=None, with_level1=True, check=False, need_weights=True, graph_only=False,
                   build=True, flags=None, with_tiles=True, host_tables=True):
        """Build from a ``Batch``-like object (attribute access only).  ``need_weights=False``
        skips everything that involves ``edge_attr`` (GINet's attention is identically 1 and
        FoutLayer never reads it, so only sGAT needs the pooled, summed edge attributes)."""
        api = api or _lib.get()
        edge_index = _contig(data.edge_index, torch.int64)
        batch = _contig(data.batch, torch.int64)
        if api is _lib._API:
            _lib.require_device(edge_index, batch)
        device = batch.device
        n_nodes = batch.numel()
        n_edges = edge_index.size(1) if edge_index.dim() == 2 else 0
        edge_attr = getattr(data, "edge_attr", None) if need_weights else None
        if edge_attr is not None:
            if edge_attr.dim() == 2 and edge_attr.size(1) != 1:
                raise ValueError("only one edge feature is supported (the reference's layers "
                                 "broadcast edge_attr [E,1] over the channels, sGAT.py:76)")
            edge_attr = _contig(edge_attr.reshape(-1), torch.float32)
        cluster0 = _contig(getattr(data, "cluster0", None), torch.int64)
        if cluster0 is None and not graph_only:
            raise ValueError("the batch has no cluster0 (pre-computed communities, DataSet.py:342-357)")
        if graph_only:
            cluster0 = None
            with_level1 = False
        cluster1 = _contig(getattr(data, "cluster1", None), torch.int64) if with_level1 else None
        d = getattr(data, "__dict__", {})
        n_graphs = d.get("_num_graphs")
        if n_graphs is None:
            n_graphs = getattr(data, "num_graphs", None)
        if n_graphs is None:
            n_graphs = int(batch.max()) + 1 if n_nodes else 0
        node_ptr = _contig(d.get("_node_ptr"), torch.int32)
        edge_ptr = _contig(d.get("_edge_ptr"), torch.int32)
        c1_ptr = _contig(d.get("_c1_ptr"), torch.int32) if cluster1 is not None else None
        for t in (node_ptr, edge_ptr, c1_ptr):
            if t is not None and t.device != device:
                node_ptr = edge_ptr = c1_ptr = None      # stale host copies: let the device derive them
                break
        topo = cls(api, n_nodes, n_edges, n_graphs, device, edge_attr is not None)
        max_nodes = int(d.get("_max_nodes", 0))
        max_edges = int(d.get("_max_edges", 0))
        max_c0 = int(d.get("_max_c0", 0))
        if max_nodes == 0 and n_nodes and n_graphs:
            # A batch object that is not this package's (a torch_geometric Batch: attribute access only): the per-graph sizes are
            # derived here -- nodes, edges and distinct depth-0 clusters per graph, ONE host round trip for all of it -- so that it
            # gets the same bounds (LDS layout of the fused kernels) and, with ``host_tables``, the same offset tables (lean
            # builder chains, offsets in the kernel arguments) as a batch collated by data.Batch.from_data_list.  Tables that do
            # not add up (batch vector / edge list not grouped by graph, cluster1 of the wrong length) are left to the device,
            # which derives what it can and flags the rest (DRGNN_S_*).
            cn = torch.bincount(batch, minlength=n_graphs)
            be = batch[edge_index[0]] if n_edges else None
            ce = torch.bincount(be, minlength=n_graphs) if n_edges else torch.zeros_like(cn)
            cc = torch.zeros_like(cn)
            if cluster0 is not None:
                lo = cluster0.min()
                span = cluster0.max() - lo + 1
                keys = torch.unique(batch * span + (cluster0 - lo))
                cc = torch.bincount(torch.div(keys, span, rounding_mode="floor"), minlength=n_graphs)
            grouped = (batch[1:] >= batch[:-1]).all() if n_nodes > 1 else torch.ones((), dtype=torch.bool, device=device)
            if n_edges > 1:
                grouped = grouped & (be[1:] >= be[:-1]).all()
            flag = grouped.to(cn.dtype).reshape(1).expand(n_graphs)
            counts = torch.stack([cn[:n_graphs], ce[:n_graphs], cc[:n_graphs], flag]).cpu()
            max_nodes, max_edges = int(counts[0].max()), int(counts[1].max())
            if cluster0 is not None:
                max_c0 = int(counts[2].max())
            if host_tables and node_ptr is None and bool(counts[3, 0]) and int(counts[0].sum()) == n_nodes and \
                    int(counts[1].sum()) == n_edges and cn.numel() == n_graphs:
                ptr = torch.zeros((3, n_graphs + 1), dtype=torch.int32)
                ptr[:, 1:] = torch.cumsum(counts[:3], dim=1).to(torch.int32)
                dev_ptr = ptr.to(device)
                node_ptr, edge_ptr = dev_ptr[0], dev_ptr[1]
                if cluster1 is not None and int(counts[2].sum()) == cluster1.numel():
                    c1_ptr = dev_ptr[2]
                topo.host_node_ptr, topo.host_edge_ptr = ptr[0].numpy(), ptr[1].numpy()
        topo.max_nodes, topo.max_edges = max_nodes, max_edges
        topo.max_c0 = max_c0
        scratch = None
        lds_limit = 160 * 1024
        need = api.topology_lds_bytes(max_nodes, max_edges)
        if max_nodes == 0 or need > lds_limit:
            scratch = torch.empty(api.topology_scratch_elems(n_nodes, n_edges, n_graphs),
                                  dtype=torch.int32, device=device)
        topo.has_level1 = cluster1 is not None
        topo._inputs = (edge_index, edge_attr, batch, cluster0, cluster1, node_ptr, edge_ptr, c1_ptr, scratch)
        x = getattr(data, "x", None) if (with_tiles and cluster1 is not None) else None
        if x is not None and x.dim() == 2 and x.dtype == torch.float32 and x.device == device and scratch is None:
            x = x.contiguous()
            F = int(x.shape[1])
            # (float4 loads of the rows where rows are multiples of 16 bytes; any other feature count is staged word by word)
            if x.shape[0] == n_nodes and F > 0 and (F % 4 != 0 or x.data_ptr() % 16 == 0) and \
                    (api.topology_tiles_ok(max_nodes, max_edges, F) or (F <= 64 and max_nodes < 1024 and max_edges <= 2048)):
                topo.x, topo.n_feat = x, F
                topo.tiles = torch.empty(max(api.topology_tiles_elems(n_nodes, F), 4), dtype=torch.float32, device=device)
                # a graph beyond what the builder stages an x tile for: the tiles are formed from the built workspace by a
                # launch of their own (drgnn_topology_tiles: the same bits) behind every rebuild(); a launch that CO-builds
                # this workspace leaves it without tiles
                topo._tiles_separately = not api.topology_tiles_ok(max_nodes, max_edges, F)
        if build:
            topo.rebuild(flags)
            if check:
                topo.check()
        return topo

    def request(self, flags=None):
        """The builder's arguments as a ``drgnn_topology_request`` (to have a body launch of the
        PREVIOUS mini-batch build this topology in the same launch)."""
        edge_index, edge_attr, batch, cluster0, cluster1, node_ptr, edge_ptr, c1_ptr, scratch = self._inputs
        r = _lib.TopologyRequest()
        p = _lib._ptr
        r.edge_index, r.edge_attr, r.batch, r.cluster0, r.cluster1 = p(edge_index), p(edge_attr), p(batch), p(cluster0), p(cluster1)
        r.node_ptr, r.edge_ptr, r.c1_ptr = p(node_ptr), p(edge_ptr), p(c1_ptr)
        r.n_nodes, r.n_edges, r.n_graphs = self.n_nodes, self.n_edges, self.n_graphs
        r.len_cluster1 = 0 if cluster1 is None else cluster1.numel()
        r.max_nodes, r.max_edges = self.max_nodes, self.max_edges
        r.ws_i32, r.ws_f32, r.scratch_i32 = p(self.ws_i32), p(self.ws_f32), p(scratch)
        r.flags = self.full_flags() if flags is None else int(flags)
        self._tiles_pending = False
        if (r.flags & _lib.TOPO_TILES) and getattr(self, "_tiles_separately", False):
            r.flags &= ~_lib.TOPO_TILES
            self._tiles_pending = True
        if r.flags & _lib.TOPO_TILES:
            if self.tiles is None:
                raise ValueError("this topology has no aggregation tiles (no float32 x the builder's LDS holds a tile of)")
            r.x, r.tiles, r.n_feat = p(self.x), p(self.tiles), self.n_feat
            # the tiles bake values of x in: what x was when the builder read it (trainer._usable_flags compares)
            self._tiles_x_version = self.x._version
        self.flags = int(r.flags)
        self._finalized = False
        return r

    def full_flags(self):
        """Everything the builder can put into this workspace (tiles it cannot stage are formed on explicit request only:
        ``rebuild(flags | TOPO_TILES)``)."""
        own = self.tiles is not None and not getattr(self, "_tiles_separately", False)
        return _lib.TOPO_HIER | (_lib.TOPO_TILES if own else 0)

    def rebuild(self, flags=None):
        """(Re)run the builder into this object's existing buffers, on torch's current stream --
        e.g. on a side stream while the previous mini-batch is still training (the build only
        depends on the inputs, never on parameters).  The input tensors captured at
        construction are re-read (with TOPO_TILES also the node features ``self.x``), so refreshing them in place refreshes
        the topology.
        ``flags``: TOPO_* request flags (default: everything, with the hierarchical order)."""
        edge_index, edge_attr, batch, cluster0, cluster1, node_ptr, edge_ptr, c1_ptr, scratch = self._inputs
        if flags is None:
            flags = self.full_flags()
        if int(flags) != _lib.TOPO_HIER:
            stream = _lib.current_stream(batch)
            self.api.topology_build_request(self.request(flags), stream)
            if getattr(self, "_tiles_pending", False):
                self.api.topology_tiles(self.ws_i32, self.ws_f32, self.n_nodes, self.n_edges, self.n_graphs, self.x, self.n_feat,
                                        self.ws_f32 is not None, self.tiles, stream)
                self._tiles_x_version = self.x._version
                self.flags |= _lib.TOPO_TILES
                self._tiles_pending = False
            return self
        self.flags = _lib.TOPO_HIER          # (drgnn_topology_build always builds the hierarchical order)
        self.api.topology_build(edge_index, edge_attr, batch, cluster0, cluster1, node_ptr, edge_ptr, c1_ptr,
                                self.n_nodes, self.n_edges, 0 if cluster1 is None else cluster1.numel(),
                                self.n_graphs, self.max_nodes, self.max_edges, self.ws_i32, self.ws_f32,
                                scratch, _lib.current_stream(batch))
        self._finalized = False
        return self

    @classmethod
    def single_graph(cls, edge_index, edge_attr, n_nodes, api=None, cluster=None):
        """Workspace for ONE graph given as bare tensors (the stand-alone layers and the
        scatter-style functions): CSR/CSC only, or with ``cluster`` also its member lists."""
        import types
        dev = edge_index.device if edge_index is not None else cluster.device
        if edge_index is None:
            edge_index = torch.zeros((2, 0), dtype=torch.int64, device=dev)
        shadow = types.SimpleNamespace(
            edge_index=edge_index, edge_attr=edge_attr,
            batch=torch.zeros(n_nodes, dtype=torch.int64, device=dev), cluster0=cluster, cluster1=None)
        shadow.__dict__["_num_graphs"] = 1 if n_nodes > 0 else 0
        shadow.__dict__["_node_ptr"] = torch.tensor([0, n_nodes], dtype=torch.int32, device=dev)
        shadow.__dict__["_edge_ptr"] = torch.tensor([0, edge_index.size(1)], dtype=torch.int32, device=dev)
        shadow.__dict__["_max_nodes"] = n_nodes
        shadow.__dict__["_max_edges"] = int(edge_index.size(1))
        return cls.from_batch(shadow, api=api, with_level1=False, graph_only=cluster is None)

    def totals(self):
        """(C0_total, E1_total, C1_total) -- finalizes and synchronises."""
        self.finalize()
        B = self.n_graphs
        vals = torch.stack([self.array("CPTR0")[B], self.array("E1PTR")[B], self.array("CPTR1")[B]]).tolist()
        return int(vals[0]), int(vals[1]), int(vals[2])

    # ---------------------------------------------------------------------------
    def array(self, name):
        """View of one int32 array of the workspace (enum drgnn_topo_i32)."""
        k = _lib.TI[name]
        return self.ws_i32[self.off_i32[k]:self.off_i32[k + 1]]

    def tile_arrays(self):
        """(S [n, F], D [n], C [n]) views of the level-0 aggregation tiles (include/drgnn.h, DRGNN_TOPO_TILES; the rows of S
        are stored padded to a multiple of 4 floats)."""
        n, F = self.n_nodes, self.n_feat
        TF = (F + 3) // 4 * 4
        t = self.tiles
        return t[:n * TF].view(n, TF)[:, :F], t[n * TF:n * TF + n], t[n * TF + n:n * TF + 2 * n]

    def weights(self, name):
        k = _lib.TF[name]
        return self.ws_f32[self.off_f32[k]:self.off_f32[k + 1]]

    def status(self):
        return self.api.topology_status(self.ws_i32, self.n_nodes, self.n_edges, self.n_graphs,
                                        _lib.current_stream(self.ws_i32))

    def check(self):
        """Synchronising validity check of the index tensors the workspace was built from."""
        st = self.status()
        if st[0]:
            msgs = [m for bit, m in _lib.STATUS_BITS.items() if st[0] & bit]
            raise _lib.DrgnnError("malformed batch (graph %d): %s" % (st[1], "; ".join(msgs)))

    def finalize(self):
        if not self._finalized:
            self.api.topology_finalize(self.ws_i32, self.n_nodes, self.n_edges, self.n_graphs,
                                       _lib.current_stream(self.ws_i32))
            self._finalized = True
        return self
