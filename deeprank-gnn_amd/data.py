"""Minimal graph containers for the hot path: ``Data``, ``Batch``, ``DataLoader``.

The reference receives its mini-batches as torch_geometric ``Batch`` objects built by
the PyG ``DataLoader`` (reference NeuralNet.py:153-154) from the per-graph ``Data``
objects assembled in HDF5DataSet.load_one_graph (reference DataSet.py:231-366).
torch_geometric is an un-vendored dependency that is absent on the target, so the
subset of its behaviour the path observes is provided here:

* block-diagonal batching: every tensor attribute is concatenated along dim 0, except
  attributes whose name contains ``index`` or ``face``, which are concatenated along
  the last dim and shifted by the running node count (so ``edge_index`` AND
  ``internal_edge_index`` are shifted; ``cluster0`` / ``cluster1`` are NOT);
* ``batch`` = graph id of every node (int64); non-tensor attributes (``mol``) become
  Python lists; scalars become 1-D tensors.

On top of that the ``Batch`` keeps ``num_graphs`` as a plain int so that the device
path never needs ``int(batch.max()) + 1`` (a host sync in the reference).

Objects of the real torch_geometric classes are accepted everywhere these are: the
models only use attribute access (duck typing).
"""
import copy

import torch

__all__ = ["Data", "Batch", "DataLoader"]


def _is_index_key(key):
    return ("index" in key) or ("face" in key)


class Data(object):
    """One graph: a bag of named attributes (tensors or Python objects)."""

    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **extra):
        store = self.__dict__
        store["x"] = x
        store["edge_index"] = edge_index
        store["edge_attr"] = edge_attr
        store["y"] = y
        store["pos"] = pos
        store.update(extra)

    # -- mapping-style access ------------------------------------------------
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith("_")]

    def __getitem__(self, key):
        return self.__dict__.get(key)

    def __setitem__(self, key, value):
        self.__dict__[key] = value

    def __contains__(self, key):
        return self.__dict__.get(key) is not None

    def __iter__(self):
        for k in self.keys():
            yield k, self.__dict__[k]

    # -- sizes ---------------------------------------------------------------
    @property
    def num_nodes(self):
        for key in ("x", "pos", "batch", "cluster0"):
            v = self.__dict__.get(key)
            if torch.is_tensor(v):
                return v.size(0)
        ei = self.__dict__.get("edge_index")
        if torch.is_tensor(ei) and ei.numel():
            return int(ei.max()) + 1
        return 0

    @property
    def num_edges(self):
        ei = self.__dict__.get("edge_index")
        return 0 if ei is None else ei.size(-1)

    @property
    def num_features(self):
        x = self.__dict__.get("x")
        if x is None:
            return 0
        return 1 if x.dim() == 1 else x.size(1)

    num_node_features = num_features

    # -- tensor plumbing -----------------------------------------------------
    def apply(self, fn):
        for k, v in self.__dict__.items():
            if torch.is_tensor(v):
                self.__dict__[k] = fn(v)
        return self

    def to(self, device, non_blocking=False):
        return self.apply(lambda t: t.to(device, non_blocking=non_blocking))

    def cpu(self):
        return self.to("cpu")

    def cuda(self, device=None):
        return self.to("cuda" if device is None else device)

    def clone(self):
        """Deep copy: tensors are cloned, everything else deep-copied (what the
        reference relies on at ginet.py:101)."""
        twin = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            twin.__dict__[k] = v.clone() if torch.is_tensor(v) else copy.deepcopy(v)
        return twin

    def __repr__(self):
        parts = []
        for k, v in self:
            parts.append("%s=%s" % (k, list(v.shape) if torch.is_tensor(v) else type(v).__name__))
        return "%s(%s)" % (self.__class__.__name__, ", ".join(parts))


class Batch(Data):
    """Several graphs stacked block-diagonally."""

    def __init__(self, batch=None, **kw):
        super().__init__(**kw)
        self.__dict__["batch"] = batch

    @property
    def num_graphs(self):
        ng = self.__dict__.get("_num_graphs")
        if ng is None:
            b = self.__dict__.get("batch")
            ng = 0 if b is None or b.numel() == 0 else int(b.max()) + 1
            self.__dict__["_num_graphs"] = ng
        return ng

    @classmethod
    def from_data_list(cls, graphs):
        if len(graphs) == 0:
            raise ValueError("cannot batch an empty list of graphs")
        names = list(graphs[0].keys())
        pieces = {k: [] for k in names}
        owner = []
        shift = 0
        for gid, g in enumerate(graphs):
            n = g.num_nodes
            for k in names:
                v = g[k]
                if torch.is_tensor(v) and _is_index_key(k) and shift:
                    v = v + shift
                pieces[k].append(v)
            owner.append(torch.full((n,), gid, dtype=torch.long))
            shift += n
        out = cls()
        for k in names:
            first = pieces[k][0]
            if torch.is_tensor(first):
                vals = [v.reshape(1) if v.dim() == 0 else v for v in pieces[k]]
                out[k] = torch.cat(vals, dim=-1 if _is_index_key(k) else 0)
            elif isinstance(first, (int, float, bool)):
                out[k] = torch.tensor(pieces[k])
            else:
                out[k] = pieces[k]
        out["batch"] = torch.cat(owner, dim=0)
        out.__dict__["_num_graphs"] = len(graphs)
        out._record_layout(graphs)
        return out

    def _record_layout(self, graphs):
        """Per-graph offsets and size bounds, known for free at collate time; they let the
        device path skip deriving them (and size its LDS) without any host sync.  Stored
        under underscore names: moved by ``.to()``, invisible to ``keys()``."""
        nodes = [g.num_nodes for g in graphs]
        edges = [g.num_edges for g in graphs]

        def ptr(counts):
            t = torch.zeros(len(counts) + 1, dtype=torch.int32)
            if counts:
                t[1:] = torch.tensor(counts, dtype=torch.int64).cumsum(0).to(torch.int32)
            return t
        d = self.__dict__
        d["_node_ptr"] = ptr(nodes)
        d["_edge_ptr"] = ptr(edges)
        # host copies (numpy; survive .to(device)): the native step passes them with the launch arguments
        d["_host_node_ptr"] = d["_node_ptr"].numpy().copy()
        d["_host_edge_ptr"] = d["_edge_ptr"].numpy().copy()
        d["_max_nodes"] = max(nodes) if nodes else 0
        d["_max_edges"] = max(edges) if edges else 0
        c1 = [g["cluster1"] for g in graphs]
        if all(torch.is_tensor(c) for c in c1):
            lens = [int(c.numel()) for c in c1]
            d["_c1_ptr"] = ptr(lens)
            d["_max_c0"] = max(lens) if lens else 0


class DataLoader(torch.utils.data.DataLoader):
    """``torch.utils.data.DataLoader`` whose collate step is ``Batch.from_data_list``
    (the role of torch_geometric.data.DataLoader at reference NeuralNet.py:11,153)."""

    def __init__(self, dataset, batch_size=1, shuffle=False, **kw):
        kw.pop("collate_fn", None)
        super().__init__(dataset, batch_size=batch_size, shuffle=shuffle,
                         collate_fn=Batch.from_data_list, **kw)
