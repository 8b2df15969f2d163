"""Graph store -> ``Data`` objects, with the tensor layout the hot path expects.

Counterpart of HDF5DataSet.load_one_graph (reference DataSet.py:231-366): same feature
stacking order (order of the ``node_feature`` list, 1-D features become one column,
:251-256), same edge symmetrisation (``[pairs ; flipped pairs]``, :266-269), same
``edge_feature_transform`` default ``tanh(-d/2+2)+1`` (:96,281), same target and
cluster handling (:318-357).

Storage: the reference keeps graphs in HDF5 (h5py, absent on the target image).  The same group / dataset tree
is accepted as
  * ``.drgs``  the native container (container.py): a lossless mirror of the HDF5 tree, read and written here without
               h5py; ``tools/hdf5_to_native.py`` / ``tools/native_to_hdf5.py`` convert from / to ``.hdf5`` where h5py
               exists (the build container's conda python);
  * ``.npz``   keys ``"<mol>/<dataset path>"`` plus ``__mols__`` (numeric datasets only; the test fixtures);
  * ``.hdf5``  directly, when h5py is importable.
``database`` may be one path or a list of paths (reference DataSet.py:116-118: ``database=glob('*.hdf5')``); ``index``
selects molecules PER FILE, as the reference's ``create_index_molecules`` does (DataSet.py:388-398).
"""
import numpy as np
import torch

from .data import Data

__all__ = ["GraphStore", "GraphDataSet", "default_edge_transform"]


def default_edge_transform(d):
    return np.tanh(-d / 2.0 + 2.0) + 1.0


class GraphStore(object):
    """``{mol: {dataset path: ndarray}}`` view of a graph file (the reference's HDF5 tree, Graph.py:61-139)."""

    def __init__(self, path):
        self.path = path
        self._mols = {}
        name = str(path)
        if name.endswith(".npz"):
            with np.load(path) as z:
                order = [str(m) for m in z["__mols__"]]
                for m in order:
                    self._mols[m] = {}
                for key in z.files:
                    if key == "__mols__":
                        continue
                    mol, _, rest = key.partition("/")
                    self._mols[mol][rest] = z[key]
        elif name.endswith(".drgs"):
            from .container import read_container
            meta, sections = read_container(path, prefix="tree/")
            for m in meta.get("mols", []):
                self._mols[m] = {}
            for key, arr in sections.items():
                mol, _, rest = key[len("tree/"):].partition("/")
                self._mols.setdefault(mol, {})[rest] = arr
        else:
            try:
                import h5py
            except ImportError as exc:  # pragma: no cover - depends on the image
                raise ImportError("reading %s needs h5py; convert it with tools/hdf5_to_native.py (writes the native "
                                  ".drgs container this class reads without h5py)" % (path,)) from exc
            with h5py.File(path, "r") as f:
                for mol in f.keys():
                    tree = {}

                    def visit(name, obj, tree=tree):
                        if isinstance(obj, h5py.Dataset) and obj.dtype.kind in "fiubS":
                            tree[name] = obj[()]
                    f[mol].visititems(visit)
                    self._mols[mol] = tree

    def save_native(self, path):
        """Write the whole tree as a native container (lossless: names, dtypes, shapes, byte strings)."""
        from .container import write_container
        sections = {}
        for mol, tree in self._mols.items():
            for k, v in tree.items():
                sections["tree/%s/%s" % (mol, k)] = np.asarray(v)
        write_container(path, sections, meta={"kind": "tree", "mols": self.mols(),
                                              "schema": "deeprank_gnn Graph.nx2h5 (reference Graph.py:61-139)"})

    def mols(self):
        return list(self._mols.keys())

    def has(self, mol, path):
        return path in self._mols[mol]

    def get(self, mol, path):
        return self._mols[mol][path]

    def set(self, mol, path, array):
        """Add / replace one dataset in memory (used by PreCluster to attach clustering/<method>/depth_k)."""
        self._mols[mol][path] = np.asarray(array)

    def save_npz(self, path):
        flat = {"__mols__": np.array(self.mols())}
        for mol, tree in self._mols.items():
            for k, v in tree.items():
                flat["%s/%s" % (mol, k)] = v
        np.savez_compressed(path, **flat)

    def children(self, mol, prefix):
        prefix = prefix.rstrip("/") + "/"
        return sorted({k[len(prefix):].split("/")[0] for k in self._mols[mol] if k.startswith(prefix)})


class GraphDataSet(torch.utils.data.Dataset):
    """Indexable dataset of ``Data`` graphs (role of reference HDF5DataSet)."""

    def __init__(self, database, node_feature="all", edge_feature=("dist",), target=None,
                 clustering_method="mcl", edge_feature_transform=default_edge_transform,
                 index=None):
        # one store per file; (store number, mol) entries in file order -- reference create_index_molecules
        # (DataSet.py:368-407): `index` picks molecules of EVERY file by position
        dbs = list(database) if isinstance(database, (list, tuple)) else [database]
        if not dbs:
            raise ValueError("no database given")
        self.stores = [d if isinstance(d, GraphStore) else GraphStore(d) for d in dbs]
        self.entries = []
        for k, st in enumerate(self.stores):
            mols = st.mols()
            if index is not None:
                mols = [mols[i] for i in index]
            self.entries += [(k, m) for m in mols]
        if not self.entries:
            raise ValueError("the database holds no molecules")
        self.store = self.stores[0]
        self.mols = [m for _, m in self.entries]
        first = self.stores[self.entries[0][0]], self.entries[0][1]
        if node_feature == "all":
            node_feature = first[0].children(first[1], "node_data")
        for feat in node_feature:
            if not first[0].has(first[1], "node_data/" + feat):
                raise KeyError("node feature %r not found; available: %s"
                               % (feat, first[0].children(first[1], "node_data")))
        self.node_feature = list(node_feature)
        self.edge_feature = None if edge_feature is None else list(edge_feature)
        self.target = target
        self.clustering_method = clustering_method
        self.edge_feature_transform = edge_feature_transform

    def __len__(self):
        return len(self.mols)

    def len(self):
        return len(self.mols)

    def __getitem__(self, i):
        k, mol = self.entries[i]
        return self.load_one_graph(mol, self.stores[k])

    get = __getitem__

    def store_of(self, i):
        """(GraphStore, mol) of entry ``i``."""
        k, mol = self.entries[i]
        return self.stores[k], mol

    def _stack(self, mol, group, names, st=None):
        st = st or self.store
        cols = []
        for feat in names:
            v = np.asarray(st.get(mol, group + "/" + feat))
            cols.append(v.reshape(-1, 1) if v.ndim == 1 else v)
        return np.hstack(cols)

    def _edges(self, mol, index_key, data_group, st=None):
        st = st or self.store
        pairs = np.asarray(st.get(mol, index_key)).reshape(-1, 2)
        both = np.vstack((pairs, pairs[:, ::-1])).T
        edge_index = torch.tensor(np.ascontiguousarray(both), dtype=torch.long)
        edge_attr = None
        if self.edge_feature is not None:
            vals = self._stack(mol, data_group, self.edge_feature, st)
            vals = self.edge_feature_transform(np.vstack((vals, vals)))
            edge_attr = torch.tensor(vals, dtype=torch.float).contiguous()
        return edge_index, edge_attr

    def load_one_graph(self, mol, st=None):
        st = st or self.store
        x = torch.tensor(self._stack(mol, "node_data", self.node_feature, st), dtype=torch.float)
        edge_index, edge_attr = self._edges(mol, "edge_index", "edge_data", st)
        iei, iea = self._edges(mol, "internal_edge_index", "internal_edge_data", st)
        y = None
        if self.target is not None and st.has(mol, "score/" + self.target):
            y = torch.tensor([st.get(mol, "score/" + self.target)[()]], dtype=torch.float)
        pos = torch.tensor(np.asarray(st.get(mol, "node_data/pos")), dtype=torch.float)
        g = Data(x=x, edge_index=edge_index, edge_attr=edge_attr, y=y, pos=pos)
        g.internal_edge_index = iei
        g.internal_edge_attr = iea
        g.mol = mol
        base = "clustering/%s/" % self.clustering_method
        if st.has(mol, base + "depth_0") and st.has(mol, base + "depth_1"):
            g.cluster0 = torch.tensor(np.asarray(st.get(mol, base + "depth_0")), dtype=torch.long)
            g.cluster1 = torch.tensor(np.asarray(st.get(mol, base + "depth_1")), dtype=torch.long)
        return g
