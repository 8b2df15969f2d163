"""Stand-ins for the un-vendored third-party packages the reference imports.

GENERATOR TOOLING ONLY (build container).  The reference (pure Python, under
/root/reference) cannot be imported as-is here because torch_geometric,
torch_scatter, h5py, markov_clustering and python-louvain are not installed
(SURVEY.md 0.3).  ``install()`` registers minimal pure-torch stand-ins for exactly
the symbols the reference's hot-path modules touch, so that the reference's OWN
layer code (ginet.py / sGAT.py / foutnet.py / community_pooling.py / DataSet.py)
runs unmodified and produces the golden vectors in tests/golden/.

The semantics restated here are the upstream ones resolved by the reference's CI
(torch 1.8.0 wheels of torch-scatter 2.0.x / torch-sparse 0.6.x, PyG 1.7-era;
reference .github/workflows/build.yml:36-48), see SURVEY.md Appendix A.

Nothing in this file is imported by the product, by tests, by smoke() or by
bench.py; it is listed in .gpurunignore and never travels to the GPU box.
"""
import copy
import re
import sys
import types

import numpy as np
import torch


# --------------------------------------------------------------------------- #
# torch_scatter
# --------------------------------------------------------------------------- #
def _broadcast(index, src, dim):
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(dim):
            index = index.unsqueeze(0)
    for _ in range(index.dim(), src.dim()):
        index = index.unsqueeze(-1)
    return index.expand_as(src)


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    index = _broadcast(index, src, dim)
    if out is None:
        size = list(src.size())
        if dim_size is not None:
            size[dim] = dim_size
        elif index.numel() == 0:
            size[dim] = 0
        else:
            size[dim] = int(index.max()) + 1
        out = torch.zeros(size, dtype=src.dtype, device=src.device)
        return out.scatter_add_(dim, index, src)
    return out.scatter_add_(dim, index, src)


scatter_add = scatter_sum


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    out = scatter_sum(src, index, dim, out, dim_size)
    dim_size = out.size(dim)
    index_dim = dim
    if index_dim < 0:
        index_dim = index_dim + src.dim()
    if index.dim() <= index_dim:
        index_dim = index.dim() - 1
    ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
    count = scatter_sum(ones, index, index_dim, None, dim_size)
    count[count < 1] = 1
    count = _broadcast(count, out, dim)
    if out.is_floating_point():
        out.true_divide_(count)
    else:
        out.div_(count, rounding_mode='floor')
    return out


class _ScatterMax(torch.autograd.Function):
    """CPU torch_scatter semantics: sequential scan, strict '>' => first occurrence
    wins ties; empty segment => value 0, arg = src.size(dim)."""

    @staticmethod
    def forward(ctx, src, index, dim_size):
        n, h = src.shape
        out = torch.full((dim_size, h), float('-inf'), dtype=src.dtype)
        arg = torch.full((dim_size, h), n, dtype=torch.long)
        s = src.detach()
        idx = index.tolist()
        for i in range(n):
            c = idx[i]
            upd = s[i] > out[c]
            out[c] = torch.where(upd, s[i], out[c])
            arg[c] = torch.where(upd, torch.full_like(arg[c], i), arg[c])
        out[arg == n] = 0
        ctx.save_for_backward(arg)
        ctx.n = n
        ctx.mark_non_differentiable(arg)
        return out, arg

    @staticmethod
    def backward(ctx, gout, _garg):
        (arg,) = ctx.saved_tensors
        n = ctx.n
        h = gout.size(1)
        gsrc = torch.zeros(n + 1, h, dtype=gout.dtype)
        gsrc.scatter_add_(0, arg, gout)
        return gsrc[:n], None, None


def scatter_max(src, index, dim=-1, out=None, dim_size=None):
    assert out is None and src.dim() == 2 and dim in (0, -2)
    if dim_size is None:
        dim_size = int(index.max()) + 1 if index.numel() > 0 else 0
    return _ScatterMax.apply(src, index, dim_size)


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce='sum'):
    if reduce in ('sum', 'add'):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == 'mean':
        return scatter_mean(src, index, dim, out, dim_size)
    if reduce == 'max':
        return scatter_max(src, index, dim, out, dim_size)[0]
    raise ValueError(reduce)


# --------------------------------------------------------------------------- #
# torch_geometric (subset)
# --------------------------------------------------------------------------- #
def uniform(size, tensor):
    if tensor is not None:
        bound = 1.0 / np.sqrt(size)
        tensor.data.uniform_(-bound, bound)


def remove_self_loops(edge_index, edge_attr=None):
    mask = edge_index[0] != edge_index[1]
    edge_index = edge_index[:, mask]
    if edge_attr is None:
        return edge_index, None
    return edge_index, edge_attr[mask]


def add_self_loops(*a, **k):  # imported by the reference, never called
    raise NotImplementedError


def softmax(*a, **k):  # imported by the reference, never called
    raise NotImplementedError


def coalesce(index, value, m, n, op='add'):
    """torch_sparse.coalesce: sort by row*n+col, merge duplicates (value summed)."""
    row, col = index
    key = row * n + col
    perm = torch.argsort(key, stable=True)
    key = key[perm]
    uniq, inv = torch.unique_consecutive(key, return_inverse=True)
    row_o = torch.div(uniq, n, rounding_mode='floor')
    col_o = uniq - row_o * n
    index_o = torch.stack([row_o, col_o], dim=0)
    if value is None:
        return index_o, None
    value = value[perm]
    value_o = scatter_sum(value, inv, 0, None, uniq.numel())
    return index_o, value_o


def consecutive_cluster(src):
    unique, inv = torch.unique(src, sorted=True, return_inverse=True)
    perm = torch.arange(inv.size(0), dtype=inv.dtype, device=inv.device)
    perm = inv.new_empty(unique.size(0)).scatter_(0, inv, perm)
    return inv, perm


def pool_edge(cluster, edge_index, edge_attr=None):
    num_nodes = cluster.size(0)
    edge_index = cluster[edge_index.view(-1)].view(2, -1)
    edge_index, edge_attr = remove_self_loops(edge_index, edge_attr)
    if edge_index.numel() > 0:
        edge_index, edge_attr = coalesce(edge_index, edge_attr, num_nodes, num_nodes)
    return edge_index, edge_attr


def pool_batch(perm, batch):
    return batch[perm]


def pool_pos(cluster, pos):
    return scatter_mean(pos, cluster, dim=0)


def max_pool_x(cluster, x, batch, size=None):
    assert size is None
    cluster, perm = consecutive_cluster(cluster)
    x = scatter(x, cluster, dim=0, dim_size=None, reduce='max')
    batch = pool_batch(perm, batch)
    return x, batch


_INDEX_RE = re.compile('(index|face)')


class Data(object):
    def __init__(self, x=None, edge_index=None, edge_attr=None, y=None, pos=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        self.edge_attr = edge_attr
        self.y = y
        self.pos = pos
        for k, v in kwargs.items():
            setattr(self, k, v)

    @property
    def keys(self):
        return [k for k, v in self.__dict__.items() if v is not None and not k.startswith('__')]

    def __getitem__(self, key):
        return getattr(self, key, None)

    def __setitem__(self, key, value):
        setattr(self, key, value)

    def __contains__(self, key):
        return key in self.keys

    @property
    def num_nodes(self):
        if self.x is not None:
            return self.x.size(0)
        if self.pos is not None:
            return self.pos.size(0)
        return int(self.edge_index.max()) + 1

    @property
    def num_features(self):
        if self.x is None:
            return 0
        return 1 if self.x.dim() == 1 else self.x.size(1)

    @property
    def num_graphs(self):
        return int(self.batch.max()) + 1

    def __cat_dim__(self, key, value):
        return -1 if bool(_INDEX_RE.search(key)) else 0

    def __inc__(self, key, value):
        return self.num_nodes if bool(_INDEX_RE.search(key)) else 0

    def apply(self, func):
        for k in self.keys:
            v = self.__dict__[k]
            if torch.is_tensor(v):
                self.__dict__[k] = func(v)
        return self

    def to(self, device, *a, **k):
        return self.apply(lambda t: t.to(device, *a, **k))

    def clone(self):
        out = self.__class__.__new__(self.__class__)
        for k, v in self.__dict__.items():
            out.__dict__[k] = v.clone() if torch.is_tensor(v) else copy.deepcopy(v)
        return out


class Batch(Data):
    def __init__(self, batch=None, **kwargs):
        super().__init__(**kwargs)
        self.batch = batch

    @staticmethod
    def from_data_list(data_list, follow_batch=[]):
        keys = data_list[0].keys
        batch = Batch()
        cols = {k: [] for k in keys}
        bvec = []
        cum = 0
        for i, d in enumerate(data_list):
            n = d.num_nodes
            for k in keys:
                v = d[k]
                if torch.is_tensor(v):
                    inc = d.__inc__(k, v)
                    if inc:
                        v = v + cum
                cols[k].append(v)
            bvec.append(torch.full((n,), i, dtype=torch.long))
            cum += n
        for k in keys:
            v0 = cols[k][0]
            if torch.is_tensor(v0):
                batch[k] = torch.cat(cols[k], dim=data_list[0].__cat_dim__(k, v0))
            elif isinstance(v0, (int, float)):
                batch[k] = torch.tensor(cols[k])
            else:
                batch[k] = cols[k]
        batch.batch = torch.cat(bvec, dim=0)
        return batch


class Dataset(torch.utils.data.Dataset):
    def __init__(self, root=None, transform=None, pre_transform=None, pre_filter=None):
        self.root = root
        self.transform = transform
        self.pre_transform = pre_transform

    def __len__(self):
        return self.len()

    def __getitem__(self, idx):
        data = self.get(idx)
        return data if self.transform is None else self.transform(data)


class DataLoader(torch.utils.data.DataLoader):
    def __init__(self, dataset, batch_size=1, shuffle=False, **kwargs):
        super().__init__(dataset, batch_size, shuffle,
                         collate_fn=lambda dl: Batch.from_data_list(dl), **kwargs)


# --------------------------------------------------------------------------- #
# h5py stand-in backed by the exported fixture .npz (read-only)
# --------------------------------------------------------------------------- #
class _H5Dataset(object):
    def __init__(self, arr):
        self._a = arr

    def __getitem__(self, key):
        a = self._a
        if key == ():
            return a[()] if a.ndim == 0 else a
        return a[key]

    @property
    def shape(self):
        return self._a.shape


class _H5Group(object):
    def __init__(self, tree):
        self._t = tree

    def keys(self):
        return self._t.keys()

    def __contains__(self, k):
        return k in self._t

    def __getitem__(self, path):
        node = self._t
        for p in [p for p in path.split('/') if p]:
            node = node[p]
        return _H5Group(node) if isinstance(node, dict) else _H5Dataset(node)

    def close(self):
        pass


_NPZ_FOR = {}


def register_npz(h5_path, npz_path):
    _NPZ_FOR[h5_path] = npz_path


def _h5_File(fname, mode='r'):
    assert mode == 'r', 'h5py stand-in is read-only'
    z = np.load(_NPZ_FOR[fname])
    tree = {}
    for mol in [str(m) for m in z['__mols__']]:
        tree[mol] = {}
    for k in z.files:
        if k == '__mols__':
            continue
        parts = k.split('/')
        node = tree
        for p in parts[:-1]:
            node = node.setdefault(p, {})
        node[parts[-1]] = z[k]
    # keep h5py's alphabetical key order
    def _sort(d):
        return {k: (_sort(v) if isinstance(v, dict) else v) for k, v in sorted(d.items())}
    return _H5Group(_sort(tree))


INSTALLED = []


def install(force=False):
    """Register the stand-ins for the third-party packages that are NOT importable here (all of them in the build
    image).  Where the real torch_scatter / torch_sparse / torch_geometric / h5py exist they are left alone, so the
    generators record goldens from the real packages there; INSTALLED lists what was replaced."""
    import importlib.util
    real = set()
    if not force:
        for top in ("torch_scatter", "torch_sparse", "torch_geometric", "h5py", "community"):
            try:
                if top not in sys.modules and importlib.util.find_spec(top) is not None:
                    real.add(top)
            except (ImportError, ValueError):
                pass

    def mod(name, **attrs):
        if name.split(".")[0] in real:
            return types.ModuleType(name)          # throw-away: the real package stays in charge
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        INSTALLED.append(name)
        return m

    mod('torch_scatter', scatter_sum=scatter_sum, scatter_add=scatter_add,
        scatter_mean=scatter_mean, scatter_max=scatter_max, scatter=scatter)
    mod('torch_sparse', coalesce=coalesce)
    tg = mod('torch_geometric')
    tg.utils = mod('torch_geometric.utils', remove_self_loops=remove_self_loops,
                   add_self_loops=add_self_loops, softmax=softmax)
    tg.nn = mod('torch_geometric.nn', max_pool_x=max_pool_x)
    tg.nn.inits = mod('torch_geometric.nn.inits', uniform=uniform)
    tg.nn.pool = mod('torch_geometric.nn.pool', max_pool_x=max_pool_x)
    tg.nn.pool.pool = mod('torch_geometric.nn.pool.pool', pool_edge=pool_edge,
                          pool_batch=pool_batch, pool_pos=pool_pos)
    tg.nn.pool.consecutive = mod('torch_geometric.nn.pool.consecutive',
                                 consecutive_cluster=consecutive_cluster)
    tg.data = mod('torch_geometric.data', Data=Data, Batch=Batch,
                  DataLoader=DataLoader, Dataset=Dataset)
    tg.data.data = mod('torch_geometric.data.data', Data=Data)
    tg.data.dataset = mod('torch_geometric.data.dataset', Dataset=Dataset)
    mod('h5py', File=_h5_File)
    mod('community')
    mod('markov_clustering')
    if 'tqdm' not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except Exception:
            mod('tqdm', tqdm=lambda x, **k: x)
