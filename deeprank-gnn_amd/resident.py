"""The dataset resident in HBM, mini-batches assembled on the device.

The reference re-opens the HDF5 file and re-reads every graph, per graph and per epoch
(reference DataSet.py:231-366, ``h5py.File`` at :241), then collates each mini-batch on the host
(torch_geometric DataLoader, reference NeuralNet.py:153-154) and copies it to the device (:434,491).
An MI355X holds 288 GB: the whole graph set is uploaded ONCE, graph-major with local node ids, and a
mini-batch is a list of graph numbers handed to ``drgnn_collate`` (csrc/drgnn_collate.h) -- one launch,
no host tensor work.  The ``Batch`` it returns has the fields, dtypes and values of
``Batch.from_data_list`` (data.py; pinned by tests/golden/collate.npz) for the keys the path reads:
``x, edge_index, edge_attr, batch, cluster0, cluster1, y, mol`` plus the per-graph offset tables.
``pos`` and the internal edges, which no shipped net reads, stay on the host copy.
"""
import numpy as np
import torch

from . import _lib
from .data import Batch

__all__ = ["ResidentGraphSet"]


def _counts_to_ptr(counts):
    ptr = np.zeros(len(counts) + 1, dtype=np.int64)
    np.cumsum(np.asarray(counts, dtype=np.int64), out=ptr[1:])
    return ptr


class ResidentGraphSet(object):
    """``graphs``: a sequence of ``Data`` (e.g. a ``GraphDataSet``); every graph is read once."""

    def _open(self, device, api):
        self.api = api or _lib.get()
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        if self.api is _lib._API and self.device.type != "cuda":
            raise _lib.DrgnnError("ResidentGraphSet keeps the data in HBM: it needs an MI355X device, got %s. "
                                  "There is no CPU path." % self.device)

    def __init__(self, graphs, device, api=None, indices=None):
        self._open(device, api)
        order = range(len(graphs)) if indices is None else list(indices)
        items = [graphs[i] for i in order]
        if not items:
            raise ValueError("cannot build a resident set from zero graphs")
        first = items[0]
        has_attr = first.edge_attr is not None
        has_c0 = getattr(first, "cluster0", None) is not None
        has_c1 = getattr(first, "cluster1", None) is not None
        for g in items:
            if g.edge_attr is not None and g.edge_attr.dim() == 2 and g.edge_attr.size(1) != 1:
                raise ValueError("only one edge feature is supported (the reference's layers broadcast "
                                 "edge_attr [E,1] over the channels, sGAT.py:76)")

        def cat(parts, dtype, dim=0):
            return torch.cat([p.to(dtype) for p in parts], dim=dim).contiguous()
        y = None
        if first.y is not None:
            y = torch.cat([g.y.reshape(-1)[:1] for g in items])
        self._adopt(
            mols=[getattr(g, "mol", None) for g in items],
            n_nodes=[g.num_nodes for g in items], n_edges=[g.num_edges for g in items],
            n_c1=[int(g.cluster1.numel()) if has_c1 else 0 for g in items],
            x=cat([g.x if g.x.dim() == 2 else g.x.reshape(-1, 1) for g in items], torch.float32),
            edge_index=cat([g.edge_index.reshape(2, -1) for g in items], torch.int64, dim=1),     # local ids
            edge_attr=cat([g.edge_attr.reshape(-1) for g in items], torch.float32) if has_attr else None,
            cluster0=cat([g.cluster0 for g in items], torch.int64) if has_c0 else None,
            cluster1=cat([g.cluster1 for g in items], torch.int64) if has_c1 else None, y=y)

    def _adopt(self, mols, n_nodes, n_edges, n_c1, x, edge_index, edge_attr, cluster0, cluster1, y):
        """Takes the concatenated host arrays, uploads them and fills the drgnn_graph_set descriptor."""
        dev = self.device
        self.mols = list(mols)
        self.n_nodes = np.asarray(n_nodes, dtype=np.int64)
        self.n_edges = np.asarray(n_edges, dtype=np.int64)
        self.n_c1 = np.asarray(n_c1, dtype=np.int64)
        self.node_ptr, self.edge_ptr, self.c1_ptr = (_counts_to_ptr(c) for c in (self.n_nodes, self.n_edges, self.n_c1))
        self.has_attr, self.has_c0, self.has_c1 = edge_attr is not None, cluster0 is not None, cluster1 is not None
        self.has_y = y is not None
        self.x = x.to(torch.float32).contiguous().to(dev)
        self.n_feat = int(self.x.size(1))
        self.edge_index = edge_index.to(torch.int64).contiguous().to(dev)
        self.edge_attr = edge_attr.to(torch.float32).contiguous().to(dev) if self.has_attr else None
        self.cluster0 = cluster0.to(torch.int64).contiguous().to(dev) if self.has_c0 else None
        self.cluster1 = cluster1.to(torch.int64).contiguous().to(dev) if self.has_c1 else None
        self.y = None
        self.y_host = None      # the targets on the host as well (a pass's targets in visiting order are a host gather: no device work)
        if self.has_y:
            self.y_host = (y.to(torch.float32) if y.is_floating_point() else y.to(torch.int64)).contiguous().cpu()
            self.y = self.y_host.to(dev)
        self._ptr_dev = [torch.from_numpy(p).to(dev) for p in (self.node_ptr, self.edge_ptr, self.c1_ptr)]
        gs = _lib.GraphSet()
        gs.n_graphs, gs.n_nodes, gs.n_edges = len(self.mols), int(self.node_ptr[-1]), int(self.edge_ptr[-1])
        gs.len_cluster1 = int(self.c1_ptr[-1])
        gs.n_feat = self.n_feat
        gs.y_bytes = 0 if self.y is None else self.y.element_size()
        p = _lib._ptr
        gs.node_ptr, gs.edge_ptr = p(self._ptr_dev[0]), p(self._ptr_dev[1])
        gs.c1_ptr = p(self._ptr_dev[2]) if self.has_c1 else None
        gs.x, gs.edge_index, gs.edge_attr = p(self.x), p(self.edge_index), p(self.edge_attr)
        gs.cluster0, gs.cluster1, gs.y = p(self.cluster0), p(self.cluster1), p(self.y)
        self._desc = gs

    def __len__(self):
        return len(self.mols)

    # -- one-file image of the set: start-up without per-graph parsing ------------------------------------
    def save(self, path):
        """Write the set as ONE ``.npz`` of concatenated arrays (the layout that is uploaded), so that later runs
        skip the per-graph reads of the HDF5 / npz store (reference DataSet.py:231-366 does them per epoch)."""
        arrays = {"node_ptr": self.node_ptr, "edge_ptr": self.edge_ptr, "c1_ptr": self.c1_ptr,
                  "x": self.x.cpu().numpy(), "edge_index": self.edge_index.cpu().numpy(),
                  "mol": np.asarray(["" if m is None else str(m) for m in self.mols])}
        for name in ("edge_attr", "cluster0", "cluster1", "y"):
            t = getattr(self, name)
            if t is not None:
                arrays[name] = t.cpu().numpy()
        np.savez(path, **arrays)

    @classmethod
    def load(cls, path, device, api=None):
        """Inverse of ``save``: the arrays go to the device as they are (no per-graph objects)."""
        with np.load(path, allow_pickle=False) as z:
            a = {k: z[k] for k in z.files}
        self = cls.__new__(cls)
        self._open(device, api)

        def opt(name):
            return torch.from_numpy(a[name]) if name in a else None
        self._adopt(mols=[str(m) for m in a["mol"]], n_nodes=np.diff(a["node_ptr"]), n_edges=np.diff(a["edge_ptr"]),
                    n_c1=np.diff(a["c1_ptr"]), x=torch.from_numpy(a["x"]), edge_index=torch.from_numpy(a["edge_index"]),
                    edge_attr=opt("edge_attr"), cluster0=opt("cluster0"), cluster1=opt("cluster1"), y=opt("y"))
        return self

    # -- cached topology (declared mode) ---------------------------------------------------------------------
    def topology_cache(self, need_weights=False):
        """The topology of EVERY graph of the set, built once (``TopologyCache``).  Per-graph topology is independent of
        the mini-batch (ids inside a graph's segment are local), so the fused step can read graph ``ids[g]``'s segments in
        place: no builder workgroups, no per-step index work (``FusedTrainer.train_step_cached`` / ``train_epoch(...,
        cached=True)``).  Built by the same builder as a mini-batch's workspace (resident-set mode, ids = 0..G-1)."""
        key = bool(need_weights and self.has_attr)
        cache = getattr(self, "_topo_cache", {}).get(key)
        if cache is None:
            if not hasattr(self, "_topo_cache"):
                self._topo_cache = {}
            cache = self._topo_cache[key] = TopologyCache(self, key)
        return cache

    def topology_cache_bytes(self, need_weights=False):
        """HBM bytes ``topology_cache`` takes for this set (workspace + edge weights + aggregation tiles); 0 once it exists."""
        key = bool(need_weights and self.has_attr)
        if getattr(self, "_topo_cache", {}).get(key) is not None:
            return 0
        G = len(self)
        N, E = int(self.node_ptr[-1]), int(self.edge_ptr[-1])
        off_i, off_f = self.api.topology_layout(N, E, G)
        return 4 * (int(off_i[-1]) + (int(off_f[-1]) if key else 0) + int(self.api.topology_tiles_elems(N, self.n_feat)) + 3 * (G + 1))

    # -- native container: the uploaded image, optionally with the cached topology ------------------------------
    def save_native(self, path, with_topology=True, need_weights=None):
        """Write the set as ONE native container (container.py): sections ``set/*`` = the concatenated arrays exactly as
        they are uploaded, and (``with_topology``) ``topo/*`` = the cached per-graph topology (int32 CSR / CSC of both
        levels, consecutive clusters, member lists, pooled CSR; f32 edge weights when the set has edge attributes), so a
        later run uploads everything with a few copies: no per-graph parsing, no topology build at all."""
        from .container import write_container
        sec = {"set/node_ptr": self.node_ptr, "set/edge_ptr": self.edge_ptr, "set/c1_ptr": self.c1_ptr,
               "set/x": self.x.cpu().numpy(), "set/edge_index": self.edge_index.cpu().numpy()}
        for name in ("edge_attr", "cluster0", "cluster1", "y"):
            t = getattr(self, name)
            if t is not None:
                sec["set/" + name] = t.cpu().numpy()
        meta = {"kind": "resident_set", "mols": ["" if m is None else str(m) for m in self.mols],
                "n_graphs": len(self), "n_feat": self.n_feat, "abi": 1}
        if with_topology and self.has_c0 and self.has_c1:
            want_w = self.has_attr if need_weights is None else bool(need_weights and self.has_attr)
            cache = self.topology_cache(need_weights=want_w)
            st = cache.topo.status()
            if st[0]:
                raise _lib.DrgnnError("the set's topology is flagged malformed (status %d, graph %d): not saved" % (st[0], st[1]))
            sec["topo/ws_i32"] = cache.topo.ws_i32.cpu().numpy()
            if cache.topo.ws_f32 is not None:
                sec["topo/ws_f32"] = cache.topo.ws_f32.cpu().numpy()
            if getattr(cache.topo, "tiles", None) is not None:
                sec["topo/tiles"] = cache.topo.tiles.cpu().numpy()      # the set's level-0 aggregation tiles (TOPO_TILES)
            meta["topology"] = {"flags": int(cache.topo.flags),"n_nodes": cache.topo.n_nodes, "n_edges": cache.topo.n_edges, "n_graphs": cache.topo.n_graphs,
                                "with_weights": cache.topo.ws_f32 is not None,
                                "off_i32": [int(v) for v in cache.topo.off_i32], "off_f32": [int(v) for v in cache.topo.off_f32],
                                "arrays_i32": sorted(_lib.TI, key=_lib.TI.get), "arrays_f32": sorted(_lib.TF, key=_lib.TF.get)}
        write_container(path, sec, meta=meta)

    @classmethod
    def load_native(cls, path, device, api=None):
        """Inverse of ``save_native``.  A stored topology is adopted as the set's cache when its layout is the one this
        library lays out for the same shape (else it is ignored and rebuilt on demand)."""
        from .container import read_container
        from .topology import Topology
        meta, a = read_container(path)
        if meta.get("kind") != "resident_set":
            raise ValueError("%s holds %r, not a resident set" % (path, meta.get("kind")))
        self = cls.__new__(cls)
        self._open(device, api)

        def opt(name):
            return torch.from_numpy(a["set/" + name]) if "set/" + name in a else None
        self._adopt(mols=[(m or None) for m in meta["mols"]], n_nodes=np.diff(a["set/node_ptr"]),
                    n_edges=np.diff(a["set/edge_ptr"]), n_c1=np.diff(a["set/c1_ptr"]), x=torch.from_numpy(a["set/x"]),
                    edge_index=torch.from_numpy(a["set/edge_index"]), edge_attr=opt("edge_attr"),
                    cluster0=opt("cluster0"), cluster1=opt("cluster1"), y=opt("y"))
        t = meta.get("topology")
        if t is not None and "topo/ws_i32" in a:
            topo = Topology(self.api, t["n_nodes"], t["n_edges"], t["n_graphs"], self.device, t["with_weights"])
            same = ([int(v) for v in topo.off_i32] == t["off_i32"] and [int(v) for v in topo.off_f32] == t["off_f32"]
                    and t["n_graphs"] == len(self) and topo.ws_i32.numel() == a["topo/ws_i32"].size)
            if same:
                topo.ws_i32.copy_(torch.from_numpy(a["topo/ws_i32"]))
                if topo.ws_f32 is not None:
                    topo.ws_f32.copy_(torch.from_numpy(a["topo/ws_f32"]))
                topo.max_nodes, topo.max_edges = int(self.n_nodes.max()), int(self.n_edges.max())
                topo.max_c0, topo.has_level1, topo._inputs = int(self.n_c1.max()), True, None
                topo.flags = int(t.get("flags", _lib.TOPO_HIER)) & ~_lib.TOPO_TILES
                if "topo/tiles" in a and a["topo/tiles"].size >= self.api.topology_tiles_elems(t["n_nodes"], self.n_feat) and \
                        (int(t.get("flags", 0)) & _lib.TOPO_TILES):
                    topo.tiles = torch.from_numpy(a["topo/tiles"]).to(self.device)
                    topo.n_feat = self.n_feat
                    topo.flags |= _lib.TOPO_TILES
                self._topo_cache = {bool(t["with_weights"]): TopologyCache(self, bool(t["with_weights"]), topo=topo)}
                if t["with_weights"]:
                    # a weighted workspace serves the nets that ignore the weights as well; its aggregation tiles are WEIGHTED
                    # sums -- the other flavour is formed from the same workspace on first use (TopologyCache.tiles_for /
                    # desc_for), not by building a second workspace
                    self._topo_cache[False] = self._topo_cache[True]
        return self

    def set_targets(self, y):
        """Replace the targets (e.g. class labels mapped to class indices); ``y``: [G] tensor."""
        y = y.reshape(-1)
        if y.numel() != len(self):
            raise ValueError("expected %d targets, got %d" % (len(self), y.numel()))
        self.y_host = (y.to(torch.float32) if y.is_floating_point() else y.to(torch.int64)).contiguous().cpu()
        self.y = self.y_host.to(self.device)
        self.has_y = True
        self._desc.y = self.y.data_ptr()
        self._desc.y_bytes = self.y.element_size()
        for cache in getattr(self, "_topo_cache", {}).values():
            cache.refresh_targets()

    def upload_ids(self, ids):
        """Graph numbers of one or more mini-batches as a device int32 tensor (slice it per batch)."""
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        if ids.size and (ids.min() < 0 or ids.max() >= len(self)):
            raise IndexError("graph number out of range [0, %d)" % len(self))
        host = torch.from_numpy(ids.astype(np.int32))
        if self.device.type != "cuda":
            return host.to(self.device)
        # Pinned + non-blocking: a pageable copy would make the host wait for everything already enqueued on the stream (the
        # previous epoch), i.e. serialise the epochs the trainer pipelines.  The staging buffers are a small ring owned by the
        # set (an event per slot says when its last copy has left): allocating pinned memory per epoch costs tens of
        # milliseconds each time and stalls the launches behind it.
        ring = self.__dict__.setdefault("_id_staging", {"slots": [], "next": 0})
        n = int(host.numel())
        if not ring["slots"] or ring["slots"][0][0].numel() < n:
            cap = max(n, 1024)
            ring["slots"] = [[torch.empty(cap, dtype=torch.int32).pin_memory(), None] for _ in range(4)]
            ring["next"] = 0
        slot = ring["slots"][ring["next"]]
        ring["next"] = (ring["next"] + 1) % len(ring["slots"])
        if slot[1] is not None:
            slot[1].synchronize()                  # (four uploads ago: long done)
        slot[0][:n].copy_(host)
        # The copy runs on a stream of its own: enqueued on the caller's stream it would sit BEHIND the previous epoch and cost
        # every epoch a 4 us blit on the critical path (profiles/r05_epoch_boundary.txt); here it is done long before the
        # caller's stream gets to the event.  The tensor is allocated from the side stream's pool (a block just freed on the
        # caller's stream may still be read by launches that have not run yet) and handed over with record_stream.
        main = torch.cuda.current_stream(self.device)
        side = ring.get("stream")
        if side is None:
            side = ring["stream"] = torch.cuda.Stream(self.device)
        if torch.cuda.is_current_stream_capturing():
            side = main
        with torch.cuda.stream(side):
            out = torch.empty(n, dtype=torch.int32, device=self.device)
            out.copy_(slot[0][:n], non_blocking=True)
            slot[1] = torch.cuda.Event()
            slot[1].record()
        if side is not main:
            main.wait_event(slot[1])
            out.record_stream(main)
        return out

    def batch_offsets(self, ids_dev, batch_size):
        """Slot offset tables of every mini-batch of the visiting order ``ids_dev`` (``upload_ids``), one launch:
        int32 [n_batches, 3, batch_size + 1] = node / edge / cluster1 offsets (drgnn_batch_offsets)."""
        n = int(ids_dev.numel())
        nb = (n + batch_size - 1) // batch_size
        ptrs = torch.zeros((nb, 3, batch_size + 1), dtype=torch.int32, device=self.device)
        self.api.batch_offsets(self._desc, ids_dev.contiguous(), n, int(batch_size), ptrs, _lib.current_stream(ptrs))
        return ptrs

    def build_topology(self, ids, ids_dev, ptrs, need_weights=False, topo=None):
        """Topology workspace, node features and targets of the mini-batch ``ids`` straight from the resident
        set (no collate: the builder reads the graphs' index data in place).  ``ptrs``: this mini-batch's [3, B+1]
        slice of ``batch_offsets``; ``topo``: an allocated workspace of the right size to build into.
        Returns (Topology, x [N, F], y [B])."""
        from .topology import Topology
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        B = int(ids.size)
        nn, ne, nc = self.n_nodes[ids], self.n_edges[ids], self.n_c1[ids]
        N, E, C = int(nn.sum()), int(ne.sum()), int(nc.sum())
        want_w = bool(need_weights and self.has_attr)
        if topo is None:
            topo = Topology(self.api, N, E, B, self.device, want_w)
        topo.max_nodes, topo.max_edges, topo.max_c0 = int(nn.max()), int(ne.max()), int(nc.max())
        topo.has_level1 = self.has_c1
        x = torch.empty((N, self.n_feat), dtype=torch.float32, device=self.device)
        y = torch.empty((B,), dtype=self.y.dtype, device=self.device) if self.y is not None else None
        r = _lib.TopologyRequest()
        p = _lib._ptr
        r.node_ptr, r.edge_ptr = p(ptrs[0]), p(ptrs[1])
        r.c1_ptr = p(ptrs[2]) if self.has_c1 else None
        r.n_nodes, r.n_edges, r.len_cluster1, r.n_graphs = N, E, C, B
        r.max_nodes, r.max_edges = topo.max_nodes, topo.max_edges
        r.ws_i32, r.ws_f32 = p(topo.ws_i32), p(topo.ws_f32)
        scratch = None
        if self.api.topology_lds_bytes(topo.max_nodes, max(topo.max_edges, 1)) > 160 * 1024:
            scratch = torch.empty(self.api.topology_scratch_elems(N, E, B), dtype=torch.int32, device=self.device)
        r.scratch_i32 = p(scratch)
        import ctypes
        r.set = ctypes.cast(ctypes.pointer(self._desc), ctypes.c_void_p)
        r.ids, r.x_out, r.y_out = p(ids_dev), p(x), p(y)
        r.flags = _lib.TOPO_HIER
        # the level-0 aggregation tiles of the mini-batch (formed from the set's x by the builder), where it can
        if scratch is None and self.has_c1 and (self.n_feat % 4 != 0 or self.x.data_ptr() % 16 == 0) and \
                self.api.topology_tiles_ok(topo.max_nodes, topo.max_edges, self.n_feat):
            need = self.api.topology_tiles_elems(N, self.n_feat)
            if topo.tiles is None or topo.tiles.numel() < need or topo.n_feat != self.n_feat:
                topo.tiles = torch.empty(max(need, 4), dtype=torch.float32, device=self.device)
            topo.n_feat = self.n_feat
            r.tiles, r.n_feat = p(topo.tiles), self.n_feat
            r.flags |= _lib.TOPO_TILES
        else:
            topo.tiles = None
        topo.flags = int(r.flags)
        self.api.topology_build_request(r, _lib.current_stream(x))
        topo._inputs = None
        topo.x = x
        topo._tiles_x_version = x._version      # (the tiles are sums of exactly these rows: the builder gathered both)
        return topo, x, y

    def batch(self, ids, ids_dev=None):
        """The mini-batch of graphs ``ids`` (host sequence of graph numbers, slot order).  ``ids_dev``: the same
        numbers already on the device (a slice of ``upload_ids`` of a whole epoch), else they are uploaded."""
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        B = int(ids.size)
        if B == 0:
            raise ValueError("cannot batch an empty list of graphs")
        if ids.min() < 0 or ids.max() >= len(self):
            raise IndexError("graph number out of range [0, %d)" % len(self))
        if ids_dev is None:
            ids_dev = torch.from_numpy(ids.astype(np.int32)).to(self.device)
        elif ids_dev.numel() != B or ids_dev.dtype != torch.int32 or ids_dev.device != self.device:
            raise ValueError("ids_dev must be the %d graph numbers as int32 on %s" % (B, self.device))
        nn, ne, nc = self.n_nodes[ids], self.n_edges[ids], self.n_c1[ids]
        N, E, C = int(nn.sum()), int(ne.sum()), int(nc.sum())
        dev = self.device
        x = torch.empty((N, self.n_feat), dtype=torch.float32, device=dev)
        edge_index = torch.empty((2, E), dtype=torch.int64, device=dev)
        edge_attr = torch.empty((E, 1), dtype=torch.float32, device=dev) if self.has_attr else None
        owner = torch.empty((N,), dtype=torch.int64, device=dev)
        cluster0 = torch.empty((N,), dtype=torch.int64, device=dev) if self.has_c0 else None
        cluster1 = torch.empty((C,), dtype=torch.int64, device=dev) if self.has_c1 else None
        y = torch.empty((B,), dtype=self.y.dtype, device=dev) if self.y is not None else None
        ptrs = torch.empty((3, B + 1), dtype=torch.int32, device=dev)
        self.api.collate(self._desc, ids_dev.contiguous(), B, N, E, x, edge_index, edge_attr, owner, cluster0,
                         cluster1, y, ptrs[0], ptrs[1], ptrs[2] if self.has_c1 else None,
                         _lib.current_stream(x))
        out = Batch(batch=owner, x=x, edge_index=edge_index, edge_attr=edge_attr, y=y)
        if self.has_c0:
            out.cluster0 = cluster0
        if self.has_c1:
            out.cluster1 = cluster1
        out.mol = [self.mols[i] for i in ids.tolist()]
        d = out.__dict__
        d["_num_graphs"] = B
        d["_node_ptr"], d["_edge_ptr"] = ptrs[0], ptrs[1]
        d["_max_nodes"], d["_max_edges"] = int(nn.max()), int(ne.max())
        d["_host_node_ptr"] = _counts_to_ptr(nn).astype(np.int32)
        d["_host_edge_ptr"] = _counts_to_ptr(ne).astype(np.int32)
        if self.has_c1:
            d["_c1_ptr"] = ptrs[2]
            d["_max_c0"] = int(nc.max())
        return out


class TopologyCache(object):
    """One topology workspace over a whole ``ResidentGraphSet`` (drgnn_topology_cache, include/drgnn.h)."""

    def __init__(self, gset, with_weights, topo=None):
        from .topology import Topology
        self.set = gset
        api, dev = gset.api, gset.device
        G = len(gset)
        N, E = int(gset.node_ptr[-1]), int(gset.edge_ptr[-1])
        if N + G >= 2 ** 31 - 1 or E >= 2 ** 31 - 1:
            raise _lib.DrgnnError("the set is too large for one int32-indexed workspace: split it")
        if not (gset.has_c0 and gset.has_c1):
            raise ValueError("a topology cache needs cluster0 and cluster1 on every graph")
        self.with_weights = bool(with_weights)
        self.max_nodes, self.max_edges = int(gset.n_nodes.max()), int(gset.n_edges.max())
        self.max_c0 = int(gset.n_c1.max())
        if topo is None:
            topo = Topology(api, N, E, G, dev, self.with_weights)
            topo.max_nodes, topo.max_edges, topo.max_c0 = self.max_nodes, self.max_edges, self.max_c0
            topo.has_level1 = True
            self._ptrs32 = torch.from_numpy(np.stack([gset.node_ptr, gset.edge_ptr, gset.c1_ptr]).astype(np.int32)).to(dev)
            ids = torch.arange(G, dtype=torch.int32, device=dev)
            r = _lib.TopologyRequest()
            p = _lib._ptr
            r.node_ptr, r.edge_ptr, r.c1_ptr = p(self._ptrs32[0]), p(self._ptrs32[1]), p(self._ptrs32[2])
            r.n_nodes, r.n_edges, r.len_cluster1, r.n_graphs = N, E, int(gset.c1_ptr[-1]), G
            r.max_nodes, r.max_edges = self.max_nodes, self.max_edges
            r.ws_i32, r.ws_f32 = p(topo.ws_i32), p(topo.ws_f32)
            scratch = None
            if api.topology_lds_bytes(self.max_nodes, max(self.max_edges, 1)) > 160 * 1024:
                scratch = torch.empty(api.topology_scratch_elems(N, E, G), dtype=torch.int32, device=dev)
            r.scratch_i32 = p(scratch)
            import ctypes
            r.set = ctypes.cast(ctypes.pointer(gset._desc), ctypes.c_void_p)
            r.ids, r.x_out, r.y_out = p(ids), None, None
            r.flags = _lib.TOPO_HIER
            # the set's level-0 aggregation tiles: formed ONCE here, read by every training step on the cache
            x_ok = gset.n_feat % 4 != 0 or gset.x.data_ptr() % 16 == 0
            separately = False
            if scratch is None and x_ok and api.topology_tiles_ok(self.max_nodes, self.max_edges, gset.n_feat):
                topo.tiles = torch.empty(max(api.topology_tiles_elems(N, gset.n_feat), 4), dtype=torch.float32, device=dev)
                topo.n_feat = gset.n_feat
                r.tiles, r.n_feat = p(topo.tiles), gset.n_feat
                r.flags |= _lib.TOPO_TILES
            elif x_ok and gset.n_feat <= 64:
                # the set's LARGEST graph is beyond what the builder stages an x tile for (or beyond its LDS altogether): the
                # tiles come from the built workspace in a launch of their own (drgnn_topology_tiles: the builder's sums in the
                # builder's order, the same bits), so that the mini-batches of smaller graphs keep their fused kernels
                separately = True
            topo.flags = int(r.flags)
            api.topology_build_request(r, _lib.current_stream(gset.x))
            if separately:
                topo.tiles = torch.empty(max(api.topology_tiles_elems(N, gset.n_feat), 4), dtype=torch.float32, device=dev)
                topo.n_feat = gset.n_feat
                api.topology_tiles(topo.ws_i32, topo.ws_f32, N, E, G, gset.x, gset.n_feat, self.with_weights, topo.tiles,
                                   _lib.current_stream(gset.x))
                topo.flags |= _lib.TOPO_TILES
            topo._inputs = None
            self._keep = (ids, scratch)
        self.topo = topo
        d = _lib.TopologyCacheDesc()
        d.n_graphs, d.n_nodes, d.n_edges = G, N, E
        d.ws_i32, d.ws_f32, d.x = _lib._ptr(topo.ws_i32), _lib._ptr(topo.ws_f32), _lib._ptr(gset.x)
        d.flags = int(getattr(topo, "flags", 0))
        d.tiles = _lib._ptr(getattr(topo, "tiles", None) if (d.flags & _lib.TOPO_TILES) else None)
        self._desc = d
        self.refresh_targets()

    def refresh_targets(self):
        y = self.set.y
        for d in [self._desc] + list(getattr(self, "_flavour", {}).values()):
            d.y = None if y is None else y.data_ptr()
            d.y_bytes = 0 if y is None else y.element_size()

    def tiles_for(self, weighted):
        """The set's level-0 aggregation tiles in the flavour a net needs -- weighted sums for sGAT, plain sums for GINet /
        FoutNet -- or None.  The cache's own tiles have the flavour of its workspace; the other one is formed on first use
        from the built workspace (drgnn_topology_tiles)."""
        own = getattr(self.topo, "tiles", None)
        if own is None or not (int(self.topo.flags) & _lib.TOPO_TILES):
            return None
        weighted = bool(weighted and self.with_weights)
        if weighted == self.with_weights:
            return own
        alt = getattr(self, "_alt_tiles", None)
        if alt is None:
            s, t = self.set, self.topo
            alt = self._alt_tiles = torch.empty_like(own)
            s.api.topology_tiles(t.ws_i32, t.ws_f32, t.n_nodes, t.n_edges, t.n_graphs, s.x, s.n_feat, weighted, alt,
                                 _lib.current_stream(s.x))
        return alt

    def desc_for(self, weighted):
        """drgnn_topology_cache descriptor whose tiles have the flavour a net needs."""
        tiles = self.tiles_for(weighted)
        if tiles is getattr(self.topo, "tiles", None):
            return self._desc
        fl = self.__dict__.setdefault("_flavour", {})
        d = fl.get(bool(weighted))
        if d is None:
            d = _lib.TopologyCacheDesc()
            for name, _ in _lib.TopologyCacheDesc._fields_:
                setattr(d, name, getattr(self._desc, name))
            d.tiles = _lib._ptr(tiles)
            fl[bool(weighted)] = d
        return d

    def bounds(self, ids):
        """(max_nodes, max_edges, max_c0) over the graphs ``ids`` (host numbers)."""
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        s = self.set
        # numpy indexing wraps negative numbers silently while the kernel reads set_node_ptr[id] as is: refuse here,
        # whatever the batch size
        if ids.size == 0 or ids.min() < 0 or ids.max() >= len(s):
            raise IndexError("graph number out of range [0, %d)" % len(s))
        return int(s.n_nodes[ids].max()), int(s.n_edges[ids].max()), int(s.n_c1[ids].max())
