"""Trainer counterpart of the reference's ``deeprank_gnn.NeuralNet`` (NeuralNet.py) for images
without torch_geometric / h5py: same constructor vocabulary, ``train`` / ``eval`` / ``test`` /
``save_model`` / ``pretrained_model=`` flow and the same checkpoint dictionary
(NeuralNet.py:775-790), with the per-batch body executed by ``FusedTrainer`` (native step).

``PreCluster`` (MCL) runs on the device when the graphs do not carry
``clustering/<method>/depth_{0,1}`` yet.  ``database`` / ``database_eval`` / ``database_test`` take one path or a
list of paths (.drgs native container, .npz, or .hdf5 where h5py exists).  The per-epoch export keeps the reference's
group / dataset names (``epoch_%04d/{train,eval,test}/{mol,outputs,targets,raw_outputs}`` + the attributes task /
target / batch_size, NeuralNet.py:827-872) in the native container; ``tools/native_to_hdf5.py`` turns it into the HDF5
file the reference writes.  What is NOT reproduced: Louvain clustering, ``Metrics`` beyond the accuracy, plots.
"""
import os
import time

import numpy as np
import torch

from . import _lib, hostcpu
from .dataset import GraphDataSet
from .resident import ResidentGraphSet
from .topology import Topology
from .trainer import FusedTrainer

__all__ = ["NeuralNet"]


def _dist_world_rank():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size(), dist.get_rank()
    return 1, 0


def _index_on(device, idx):
    """int64 index tensor on ``device`` without making the host wait for the stream (pinned + non-blocking on the GPU)."""
    t = torch.as_tensor(idx, dtype=torch.long)
    if torch.device(device).type == "cuda":
        return t.pin_memory().to(device, non_blocking=True)
    return t.to(device)


def _divide(n, percent, shuffle):
    """reference DivideDataSet (DataSet.py:14-42): shuffled index split."""
    index = np.arange(n)
    if shuffle:
        np.random.shuffle(index)
    n1 = int(percent[0] * n)
    return index[:n1], index[n1:]


class _Staged(object):
    """Device tensors whose host copies are STARTED now (pinned buffers, non-blocking, one event behind them) and read
    later: the epoch loop enqueues the next epoch before it looks at the previous one's numbers, so the device never
    idles while the host turns outputs into the reference's Python lists (NeuralNet.py:446-460,508-523 syncs per batch)."""

    # Pinned buffers are reused: allocating page-locked memory costs ~50 us a piece, three pieces per epoch -- as much host time
    # as enqueuing eight mini-batches.  A ring of eight per (shape, dtype): at most four passes of a shape are in flight (train()
    # reads epoch e - 1 while epoch e runs), and a buffer is handed out again only after its reader has let go of it (get()).
    _ring = {}

    @classmethod
    def _pinned(cls, shape, dtype):
        key = (tuple(shape), dtype)
        slots = cls._ring.setdefault(key, {"bufs": [], "next": 0})
        if len(slots["bufs"]) < 8:
            buf = torch.empty(shape, dtype=dtype, pin_memory=True)
            slots["bufs"].append(buf)
            return buf
        buf = slots["bufs"][slots["next"]]
        slots["next"] = (slots["next"] + 1) % 8
        return buf

    def __init__(self, **tensors):
        self.host, self.event = {}, None
        for k, t in tensors.items():
            if t is None or not torch.is_tensor(t):
                self.host[k] = t
            elif t.is_cuda:
                h = self._pinned(t.shape, t.dtype)
                h.copy_(t.detach(), non_blocking=True)
                self.host[k] = h
                self.event = self.event or torch.cuda.Event()
            else:
                self.host[k] = t.detach()
        if self.event is not None:
            self.event.record()

    def get(self):
        if self.event is not None:
            self.event.synchronize()
            self.event = None
            # (the pinned buffers go back to the ring: the caller gets copies it may keep)
            self.host = {k: (v.clone() if torch.is_tensor(v) and v.is_pinned() else v) for k, v in self.host.items()}
        return self.host


class _PassStore(dict):
    """The reference's per-pass record ``{'outputs', 'raw_outputs', 'targets', 'mol'}`` (NeuralNet.py:440-460,508-523: Python
    lists, one entry per graph) whose lists are FORMED WHEN READ: a training run reads the lists of the few epochs it exports
    and of the last one, while building them costs more host time per epoch than enqueuing the epoch's launches (2048 graphs:
    ~0.4 ms of list building against 0.25 ms of launches).  ``defer(key, fn)``: ``fn()`` gives the entries to append to
    ``self[key]`` the first time somebody looks.  ``arrays``: (outputs, targets) of the pass as numpy arrays in the space the
    accuracy is judged in (what _accuracy reads instead of the lists)."""
    LISTS = ('outputs', 'raw_outputs', 'targets', 'mol')

    def __init__(self):
        super().__init__({'outputs': [], 'raw_outputs': [], 'targets': [], 'mol': [], '_pred': [], '_y': []})
        self._deferred = {}
        self.arrays = None

    def defer(self, key, fn):
        self._deferred.setdefault(key, []).append(fn)

    def _settle(self, key):
        fns = self._deferred.pop(key, None)
        if fns:
            cur = dict.__getitem__(self, key)
            for fn in fns:
                cur += fn()

    def __getitem__(self, key):
        if key in self._deferred:
            self._settle(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        if key in self._deferred:
            self._settle(key)
        return dict.get(self, key, default)

    def items(self):
        for key in list(self._deferred):
            self._settle(key)
        return dict.items(self)

    def values(self):
        for key in list(self._deferred):
            self._settle(key)
        return dict.values(self)

    def has_targets(self):
        return bool(self._deferred.get('targets')) or bool(dict.__getitem__(self, 'targets'))


class NeuralNet(object):
    def __init__(self, database, Net, node_feature=['type', 'polarity', 'bsa'], edge_feature=['dist'],
                 target='irmsd', lr=0.01, batch_size=32, percent=[1.0, 0.0], database_eval=None, index=None,
                 class_weights=None, task=None, classes=[0, 1], threshold=None, pretrained_model=None,
                 shuffle=True, outdir='./', cluster_nodes='mcl', transform_sigmoid=False, device=None,
                 _api=None):
        self._api = _api
        hostcpu.fit_torch_threads()
        self.device = torch.device(device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu'))
        if self.device.type != 'cuda' and _api is None:
            raise _lib.DrgnnError("deeprank_gnn_amd.NeuralNet needs an MI355X: there is no CPU path")
        self.outdir = outdir
        if pretrained_model is None:
            self.node_feature, self.edge_feature, self.target = node_feature, edge_feature, target
            self.lr, self.batch_size, self.percent, self.index = lr, batch_size, percent, index
            self.class_weights, self.task, self.classes, self.threshold = class_weights, task, classes, threshold
            self.shuffle, self.cluster_nodes, self.transform_sigmoid = shuffle, cluster_nodes, transform_sigmoid
            if self.task is None:                      # NeuralNet.py:64-78
                if self.target in ['irmsd', 'lrmsd', 'fnat', 'dockQ']:
                    self.task = 'reg'
                elif self.target in ['bin_class', 'capri_classes']:
                    self.task = 'class'
                else:
                    raise ValueError("User target detected -> The task argument is required ('class' or 'reg').")
            if self.threshold is None:
                self.threshold = self.classes[1] if self.task == 'class' else 0.3
            opt_state = model_state = None
        else:
            state = torch.load(pretrained_model, map_location='cpu', weights_only=False)
            for dst, src in (('node_feature', 'node'), ('edge_feature', 'edge'), ('target', 'target'),
                             ('batch_size', 'batch_size'), ('percent', 'percent'), ('lr', 'lr'), ('index', 'index'),
                             ('class_weights', 'class_weight'), ('task', 'task'), ('classes', 'classes'),
                             ('threshold', 'threshold'), ('shuffle', 'shuffle'), ('cluster_nodes', 'cluster_nodes'),
                             ('transform_sigmoid', 'transform_sigmoid')):
                setattr(self, dst, state[src])
            opt_state, model_state = state['optimizer'], state['model']
        self.pretrained = pretrained_model is not None
        self.dataset = GraphDataSet(database, node_feature=self.node_feature, edge_feature=self.edge_feature,
                                    target=self.target, clustering_method=self.cluster_nodes or 'mcl',
                                    index=None if self.pretrained else self.index)     # load_pretrained_model: no index
        first = self.dataset[0]
        if getattr(first, "cluster0", None) is None:
            # the reference runs PreCluster at every construction (NeuralNet.py:139-143); here only
            # when the graphs do not carry the labels yet (same result: MCL is deterministic)
            from .clustering import PreCluster
            print("Loading clusters")
            PreCluster(self.dataset, method=self.cluster_nodes or 'mcl', api=_api, device=self.device)
            first = self.dataset[0]
        if pretrained_model is None:
            i_train, i_valid = _divide(len(self.dataset), self.percent, True)
        else:
            i_train, i_valid = np.arange(len(self.dataset)), np.arange(0)
        self.world, self.rank = _dist_world_rank()
        if self.world > 1:
            # data parallel: ONE train / validation split for the whole job (rank 0's; np.random differs per rank)
            import torch.distributed as dist
            box = [(i_train.tolist(), i_valid.tolist())]
            dist.broadcast_object_list(box, src=0)
            i_train, i_valid = np.asarray(box[0][0], dtype=np.int64), np.asarray(box[0][1], dtype=np.int64)
        self.train_index, self.valid_index = list(i_train), list(i_valid)
        self.eval_dataset = None
        if database_eval is not None:
            self.eval_dataset = GraphDataSet(database_eval, node_feature=self.node_feature,
                                             edge_feature=self.edge_feature, target=self.target,
                                             clustering_method=self.cluster_nodes or 'mcl', index=self.index)

        n_out = 1 if self.task == 'reg' else len(self.classes)
        self.classes_to_idx = {c: i for i, c in enumerate(self.classes)}
        self.idx_to_classes = {i: c for i, c in enumerate(self.classes)}
        self.model = Net(first.num_features, n_out, len(self.edge_feature)).to(self.device)
        if model_state is not None:
            self.model.load_state_dict(model_state)
        weights = None
        if self.task == 'class' and self.class_weights is True:      # NeuralNet.py:247-258
            ys = [int(self.dataset[i].y) for i in self.train_index]
            w = torch.tensor([ys.count(c) for c in self.classes], dtype=torch.float32)
            w = 1.0 / w
            weights = w / w.sum()
        elif self.task == 'class' and isinstance(self.class_weights, (list, tuple)):
            weights = torch.tensor(self.class_weights, dtype=torch.float32)
        self.trainer = FusedTrainer(self.model, lr=self.lr, task=self.task, class_weights=weights, api=_api,
                                    transform_sigmoid=bool(self.transform_sigmoid))
        if opt_state is not None:
            self.trainer.load_optimizer_state_dict(opt_state)
        # data parallel: the replicas start as ONE model -- rank 0's parameters, Adam moments and step counter -- whatever
        # each rank's RNG produced at construction
        self.trainer.broadcast_state(src=0)
        self.train_loss, self.valid_loss, self.train_acc, self.valid_acc = [], [], [], []
        self.data = {}
        self._resident_sets = {}
        self.native_epoch = True      # False: step mini-batch by mini-batch from Python (same results)
        # Per-graph topology (CSR, consecutive clusters, pooled graph, aggregation tiles) depends on the graph alone -- the
        # reference itself precomputes the clustering once per dataset (DataSet.py:45-88) and redoes only the index work in every
        # forward (community_pooling.py:25-30,197-216).  "auto" (default): build it ONCE when a set is uploaded and let every
        # step read it in place (no builder work per mini-batch: same results bit for bit, tests/test_gpu_epoch.py) whenever
        # the cache fits ``topology_cache_budget`` bytes of HBM, else rebuild per mini-batch; True / False force either.
        self.cached_topology = "auto"
        self.topology_cache_budget = 32 << 30       # bytes per resident set (an MI355X holds 288 GB)
        self._cache_choice = {}
        self._order_cache, self._eval_targets, self._batch_counts = {}, {}, {}
        self.exported = []            # (epoch, file) of the epoch data written so far

    # ------------------------------------------------------------------------------
    def _resident(self, dataset):
        """The graph set of ``dataset`` in HBM (uploaded on first use, every graph read once): mini-batches are
        then assembled on the device from graph numbers (resident.py) instead of re-reading and collating on
        the host per batch and per epoch as the reference does (DataSet.py:231-366, NeuralNet.py:153-154)."""
        rs = self._resident_sets.get(id(dataset), (None, None))[1]
        if rs is None:
            rs = ResidentGraphSet(dataset, self.device, api=self._api)
            if self.task == 'class' and rs.y is not None:
                # format_output's target half (NeuralNet.py:616-631): class labels -> class indices
                rs.set_targets(torch.tensor([self.classes_to_idx[int(v)] for v in rs.y.cpu().tolist()]))
            self._resident_sets[id(dataset)] = (dataset, rs)       # keeps the dataset alive: ids stay unique
            keep = {id(self.dataset), id(self.eval_dataset), id(dataset)}
            for key in [k for k in self._resident_sets if k not in keep][:-2]:
                del self._resident_sets[key]                       # test sets of earlier test() calls
        return rs

    def _use_cache(self, rs):
        """Whether the steps over the resident set ``rs`` read its cached topology (``cached_topology``; logged once per set)."""
        mode = self.cached_topology
        if mode is True or mode is False:
            return mode
        key = id(rs)
        choice = self._cache_choice.get(key)
        if choice is None:
            need_w = self.trainer.kind == _lib.SGAT
            ok = bool(rs.has_c0 and rs.has_c1) and not (need_w and rs.edge_attr is None)
            nbytes = rs.topology_cache_bytes(need_weights=need_w) if ok else 0
            choice = ok and nbytes <= int(self.topology_cache_budget)
            self._cache_choice[key] = choice
            import logging
            logging.getLogger("deeprank_gnn_amd").info(
                "resident set of %d graphs: topology %s (cache %.1f MiB, budget %.1f MiB)", len(rs),
                "cached per graph, built once" if choice else "rebuilt per mini-batch", nbytes / 2 ** 20,
                int(self.topology_cache_budget) / 2 ** 20)
        return choice

    def _batches(self, dataset, indices, shuffle):
        order = [int(i) for i in indices]
        if shuffle:
            order = [order[i] for i in torch.randperm(len(order)).tolist()]
        if not order:
            return
        rs = self._resident(dataset)
        ids_dev = rs.upload_ids(order)                 # one small upload per epoch
        for lo in range(0, len(order), self.batch_size):
            yield rs.batch(order[lo:lo + self.batch_size], ids_dev[lo:lo + self.batch_size])

    def _collect(self, pred, batch, store):
        """Keeps the batch's outputs ON THE DEVICE; _finish turns them into the reference's lists once per
        pass (the reference syncs per batch: .item() / .cpu(), NeuralNet.py:446-460,508-523)."""
        store['_pred'].append(pred.detach().clone())   # the trainer reuses its output buffer
        if batch.y is not None:
            store['_y'].append(batch.y)
        store['mol'] += list(batch['mol'])

    def _stage(self, store, loss=None):
        """Device side of a finished pass: one prediction / target tensor each, their host copies and the pass's loss
        (a device scalar) and the trainer's fault word on the way to pinned memory.  No synchronisation."""
        preds, ys = store.pop('_pred'), store.pop('_y')
        store['_staged'] = _Staged(pred=torch.cat(preds) if preds else None, y=torch.cat(ys) if ys else None,
                                   loss=loss, faults=self.trainer.step2[2:3])
        return store

    def _finish(self, store):
        """Host side: waits for the staged copies of THIS pass only and fills the reference's lists; returns the loss."""
        if '_staged' not in store:
            self._stage(store)
        host = store.pop('_staged').get()
        pred, y, loss = host['pred'], host['y'], host['loss']
        self.trainer.raise_on_faults(int(host['faults'][0]))
        loss = None if loss is None else float(loss)
        if pred is None:
            return loss
        # (the lists of the reference's record are formed when somebody reads them: _PassStore)
        if self.task == 'class':
            prob = torch.softmax(pred, dim=1)
            top = prob.argmax(dim=1)
            store.defer('raw_outputs', lambda: prob.tolist())
            store.defer('outputs', lambda: [self.idx_to_classes[i] for i in top.tolist()])
            if y is not None:
                store.defer('targets', lambda: [self.idx_to_classes[int(i)] for i in y.tolist()])
            store.arrays = (top.numpy(), None if y is None else y.numpy().astype(np.int64))
        else:
            flat = pred.reshape(-1)
            store.defer('raw_outputs', lambda: flat.tolist())
            store.defer('outputs', lambda: flat.tolist())
            if y is not None:
                store.defer('targets', lambda: y.tolist())
            store.arrays = (flat.numpy(), None if y is None else y.numpy())
        return loss

    def _accuracy(self, store, threshold=None):
        """Metrics(...).accuracy of the reference (Metrics.py:10-31,113-120,170): predictions and targets are made
        binary at ``threshold`` -- in class-INDEX space for classification (NeuralNet.get_metrics, NeuralNet.py:548-549)
        -- '>' for fnat / bin_class, '<' for the others, and the accuracy is the fraction on which the two agree."""
        arrays = getattr(store, "arrays", None)
        if arrays is not None and arrays[1] is None:
            return None
        if arrays is None and not store['targets']:
            return None
        threshold = self.threshold if threshold is None else threshold
        if self.task == 'class':
            # (test()'s default threshold 4 is a capri class; other class sets fall back to the trainer's threshold)
            thr = self.classes_to_idx[threshold if threshold in self.classes_to_idx else self.threshold]
            if arrays is not None:          # (class INDICES already: what the lists' labels map back to)
                o, t = arrays
            else:
                o = np.asarray([self.classes_to_idx[v] for v in store['outputs']])
                t = np.asarray([self.classes_to_idx[v] for v in store['targets']])
        elif arrays is not None:
            thr, (o, t) = threshold, arrays
        else:
            thr, o, t = threshold, np.asarray(store['outputs']), np.asarray(store['targets'])
        if self.target in ('fnat', 'bin_class'):
            return float(np.mean((o > thr) == (t > thr)))
        return float(np.mean((o < thr) == (t < thr)))

    @staticmethod
    def _new_store():
        return _PassStore()

    def _order_of(self, indices):
        """``indices`` as an int64 tensor, built once per index list (the train / validation split does not change)."""
        key = (id(indices), len(indices))
        held = self._order_cache.get(key)
        if held is None or held[0] is not indices:
            held = self._order_cache[key] = (indices, torch.as_tensor([int(i) for i in indices], dtype=torch.int64))
        return held[1]

    def _epoch(self, epoch):
        """One pass over the training set (NeuralNet.py:477-537) on the native step, ENQUEUED: no host synchronisation
        here -- batches come from the resident set, losses / outputs stay on the device, their host copies are started
        behind the epoch (``_stage``); ``_finish(store)`` reads them (train() does that after it has enqueued the next
        epoch).  Returns the store."""
        store = self._new_store()
        import torch.distributed as dist
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        if self.train_index and world > 1:
            # data parallel: batch_size is the GLOBAL mini-batch on every path (native loop or per mini-batch)
            return self._epoch_data_parallel(store, world, dist.get_rank())
        if self.native_epoch and self.train_index and world == 1:
            # the whole epoch enqueued by the native loop (drgnn_train_epoch): collate, step (+ next topology) and
            # update launches for every mini-batch
            rs = self._resident(self.dataset)
            order = self._order_of(self.train_index)
            if self.shuffle:
                order = order[torch.randperm(order.numel())]
            done = self.trainer.train_epoch(rs, order, self.batch_size, cached=self._use_cache(rs))
            if done is not None:
                losses, pred = done
                store['_pred'].append(pred)
                # the pass's targets and molecule names in visiting order: a host gather of the set's host copy and a list
                # formed when somebody reads it -- no device work, no per-graph Python per epoch
                store['_y'].append(rs.y_host[order])
                store.defer('mol', lambda: [rs.mols[i] for i in order.tolist()])
                return self._stage(store, losses.sum())
        running = torch.zeros((), dtype=torch.float32, device=self.device)
        need_w = self.trainer.kind == _lib.SGAT
        it = self._batches(self.dataset, self.train_index, self.shuffle)
        batch = next(it, None)
        topo = None if batch is None else Topology.from_batch(batch, api=self.trainer.api, need_weights=need_w)
        while batch is not None:
            # one-batch look-ahead: the next mini-batch's topology is built inside this step's
            # backward launch (index tensors only), so only the first batch of an epoch pays a
            # builder launch of its own
            nxt = next(it, None)
            nxt_topo = None if nxt is None else Topology.from_batch(nxt, api=self.trainer.api,
                                                                    need_weights=need_w, build=False)
            loss = self.trainer.train_step(batch, topo=topo, next_topo=nxt_topo)
            running += loss.reshape(())
            self._collect(self.trainer.last_pred, batch, store)
            batch, topo = nxt, nxt_topo
        return self._stage(store, running)

    def _epoch_data_parallel(self, store, world, rank):
        """One epoch with ``batch_size`` as the GLOBAL mini-batch, sharded over the ranks (contiguous shards, sizes differ
        by <= 1) in rank 0's visiting order: per mini-batch gradient launches -> one all-reduce of the flat gradient
        (weighted n_local / n_global) -> Adam.  Every rank applies the same updates, so this equals the single-process
        epoch on the same order (unweighted mean losses; with ``class_weights`` the per-shard normalisation by the shard's
        weight sum makes it an approximation -- refused below).  The store / loss returned describe THIS rank's shard.

        Which loop runs is decided COLLECTIVELY: the native loop (drgnn_train_epoch) only if every
        mini-batch has at least one graph per rank AND every rank's shard fits it (all_reduce MIN of the ranks' probes);
        otherwise all ranks step the same global mini-batches one by one.  Either way every rank issues exactly one
        all-reduce per global mini-batch."""
        import torch.distributed as dist
        from .parallel import shard_range
        if self.task == 'class' and self.trainer.class_w is not None:
            raise _lib.DrgnnError("data-parallel training with class_weights is not supported: the weighted cross-entropy "
                                  "normalises by the weight sum of the mini-batch, which per-shard means recombined by "
                                  "graph count do not reproduce")
        order = torch.tensor([int(i) for i in self.train_index], dtype=torch.int64)
        if self.shuffle:
            order = order[torch.randperm(order.numel())]
        on_dev = dist.get_backend() == "nccl"
        if on_dev:
            od = order.to(self.device)
            dist.broadcast(od, src=0)
            order = od.cpu()
        else:
            dist.broadcast(order, src=0)                       # every rank walks rank 0's order
        order = order.tolist()
        chunks = [order[lo:lo + self.batch_size] for lo in range(0, len(order), self.batch_size)]
        parts = [c[slice(*shard_range(len(c), rank, world))] for c in chunks]
        sizes = [len(c) for c in chunks]
        rs = self._resident(self.dataset)
        # a rank's batch loss is the mean over its shard: weight it back to the global mean of the mini-batch
        w = torch.tensor([len(p) / float(n) for p, n in zip(parts, sizes)], dtype=torch.float32, device=self.device)
        native = bool(self.native_epoch) and all(len(c) >= world for c in chunks)     # (same answer on every rank)
        if native:
            local_bs = len(parts[0])
            mine = [g for p in parts for g in p]
            ok = self.trainer.train_epoch(rs, mine, local_bs, cached=self._use_cache(rs), dp_global_sizes=sizes, probe=True)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.device if on_dev else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            native = bool(int(flag.item()))
        if native:
            done = self.trainer.train_epoch(rs, mine, local_bs, cached=self._use_cache(rs), dp_global_sizes=sizes)
            if done is None:       # cannot happen after a successful probe; never continue with mismatched collectives
                raise _lib.DrgnnError("the native epoch loop refused a configuration its probe had accepted")
            losses, pred = done
            store['_pred'].append(pred)
            store['_y'].append(rs.y_host[torch.as_tensor(mine, dtype=torch.int64)])      # (host gather: no device work)
            store.defer('mol', lambda: [rs.mols[i] for i in mine])
            total = (losses * w).sum()
        else:
            # per mini-batch: the same global mini-batches, each rank steps its shard.  A rank without a graph of a
            # mini-batch smaller than the world steps a stand-in (the mini-batch's first graph) with weight 0, so that
            # the gradient all-reduce stays matched and the sum is the single-process gradient.
            need_w = self.trainer.kind == _lib.SGAT
            total = torch.zeros((), dtype=torch.float32, device=self.device)
            for k, (chunk, part) in enumerate(zip(chunks, parts)):
                ids = part if part else chunk[:1]
                batch = rs.batch(ids, rs.upload_ids(ids))
                topo = Topology.from_batch(batch, api=self.trainer.api, need_weights=need_w)
                loss = self.trainer.train_step(batch, topo=topo, n_global=len(chunk), n_local=len(part))
                total += loss.reshape(()) * w[k]
                if part:
                    self._collect(self.trainer.last_pred, batch, store)
        dist.all_reduce(total)
        return self._stage(store, total)

    def _sum_of_batch_losses(self, pred, y):
        """Sum over the mini-batches of each batch's mean loss (what the reference accumulates, NeuralNet.py:441-447)."""
        # (one segmented mean over all mini-batches instead of a loss call per mini-batch: a validation pass of 8 mini-batches
        # cost the host 0.2 ms that way -- as much as enqueuing the pass itself)
        bs, n = int(self.batch_size), int(pred.size(0))
        nfull = n // bs

        def per_batch(v):
            """[n] -> [n_batches]: sums over the mini-batches (full ones through a reshape: fixed summation order)"""
            parts = []
            if nfull:
                parts.append(v[:nfull * bs].reshape(nfull, bs).sum(dim=1))
            if n > nfull * bs:
                parts.append(v[nfull * bs:].sum().reshape(1))
            return torch.cat(parts) if len(parts) > 1 else parts[0]
        def counts():
            key = (n, bs, pred.device)
            c = self._batch_counts.get(key)
            if c is None:
                c = self._batch_counts[key] = torch.tensor([float(bs)] * nfull + ([float(n - nfull * bs)] if n > nfull * bs else []),
                                                           dtype=torch.float32, device=pred.device)
            return c
        if self.task == 'reg':
            per = ((pred.reshape(n, -1) - y.reshape(n, 1).to(pred.dtype)) ** 2).mean(dim=1)
            denom = counts()
        else:
            w = self.trainer.class_w
            per = torch.nn.functional.cross_entropy(pred, y, weight=w, reduction='none')
            denom = counts() if w is None else per_batch(w[y].to(per.dtype))
        return (per_batch(per) / denom).sum()

    def eval(self, dataset=None, indices=None):
        """Forward only (NeuralNet.py:414-475); returns (loss_sum, store)."""
        store = self._eval_enqueue(dataset, indices)
        loss = self._finish(store)
        return (0.0 if loss is None else loss), store

    def _eval_enqueue(self, dataset=None, indices=None):
        """The forward pass of ``eval`` enqueued, outputs staged (no synchronisation); ``_finish(store)`` completes it."""
        dataset = self.dataset if dataset is None else dataset
        indices = self.valid_index if indices is None else indices
        store = self._new_store()
        order = [int(i) for i in indices]
        if self.native_epoch and order:
            rs = self._resident(dataset)
            pred = self.trainer.predict_epoch(rs, order, self.batch_size, cached=self._use_cache(rs))
            if pred is not None:
                store['_pred'].append(pred)
                store.defer('mol', lambda: [rs.mols[i] for i in order])
                total = None
                if rs.y is not None:
                    held = self._eval_targets.get((id(rs), id(indices), len(order)))
                    if held is None or held[0] is not indices:      # (a validation / test pass visits the same graphs every time)
                        idx = torch.as_tensor(order, dtype=torch.int64)
                        held = self._eval_targets[(id(rs), id(indices), len(order))] = (indices, rs.y[_index_on(rs.y.device, order)], rs.y_host[idx])
                    store['_y'].append(held[2])
                    total = self._sum_of_batch_losses(pred, held[1])
                return self._stage(store, total)
        total = torch.zeros((), dtype=torch.float32, device=self.device)
        need_w = self.trainer.kind == _lib.SGAT
        it = self._batches(dataset, indices, False)
        batch = next(it, None)
        topo = None if batch is None else Topology.from_batch(batch, api=self.trainer.api, need_weights=need_w)
        while batch is not None:
            nxt = next(it, None)
            nxt_topo = None if nxt is None else Topology.from_batch(nxt, api=self.trainer.api,
                                                                    need_weights=need_w, build=False)
            pred = self.trainer.predict(batch, topo=topo, next_topo=nxt_topo)
            if batch.y is not None:
                if self.task == 'reg':
                    total += torch.nn.functional.mse_loss(pred.reshape(-1), batch.y)
                else:
                    total += torch.nn.functional.cross_entropy(pred, batch.y, weight=self.trainer.class_w)
            self._collect(pred, batch, store)
            batch, topo = nxt, nxt_topo
        return self._stage(store, total)

    def train(self, nepoch=1, validate=False, save_model='last', hdf5='train_data.drgs', save_epoch='intermediate',
              save_every=5):
        """NeuralNet.train (NeuralNet.py:265-355): same arguments; epoch data is exported for the last epoch and, with
        ``save_epoch='all'`` / ``'intermediate'``, for every / every ``save_every``-th epoch; the other epochs' outputs are
        dropped as soon as the epoch is over.  'best' checkpoints carry the reference's file name."""
        self.nepoch = nepoch
        fname = self.update_name(self._rank_name(hdf5), self.outdir) if hdf5 else None
        self.data, pending = {}, {}
        validating = validate and (bool(self.valid_index) or self.eval_dataset is not None)

        def close(epoch, t0, store, vstore):
            """host side of an epoch whose work was enqueued earlier: numbers, lists, the progress line"""
            loss = self._finish(store)
            self.train_loss.append(loss)
            self.train_acc.append(self._accuracy(store))
            self.data = {'train': store}
            line = "Epoch [%04d] : train loss %e" % (epoch, loss)
            best_of = self.train_loss
            if vstore is not None:
                vloss = self._finish(vstore)
                vloss = 0.0 if vloss is None else vloss
                self.valid_loss.append(vloss)
                self.valid_acc.append(self._accuracy(vstore))
                self.data['eval'] = vstore
                line += " | valid loss %e" % vloss
                best_of = self.valid_loss
            print(line + " | time %.3f s" % (time.time() - t0))
            if (save_epoch == 'all') or (epoch == nepoch) or (save_epoch == 'intermediate' and epoch % save_every == 0):
                pending['epoch_%04d' % epoch] = self.data
            return min(best_of) == best_of[-1]

        previous = None
        for epoch in range(1, nepoch + 1):
            # Enqueue this epoch (training pass + validation pass) BEFORE reading the previous epoch's numbers: the device
            # runs epoch e while the host turns epoch e-1's outputs into lists.  ('best' checkpoints need this epoch's
            # loss while its parameters are still the current ones: that mode closes every epoch at once.)
            t0 = time.time()
            store = self._epoch(epoch)
            vstore = None
            if validating:
                ds = self.dataset if self.eval_dataset is None else self.eval_dataset
                idx = range(len(ds)) if self.eval_dataset is not None else self.valid_index
                vstore = self._eval_enqueue(ds, list(idx))
            if previous is not None:
                close(*previous)
                previous = None
            if save_model == 'best':
                if close(epoch, t0, store, vstore):
                    self.save_model(os.path.join(self.outdir, 't{}_y{}_b{}_e{}_lr{}_{}.pth.tar'.format(
                        self.task, self.target, str(self.batch_size), str(nepoch), str(self.lr), str(epoch))))
            else:
                previous = (epoch, t0, store, vstore)
        if previous is not None:
            close(*previous)
        if save_model == 'last':
            self.save_model(os.path.join(self.outdir, 't{}_y{}_b{}_e{}_lr{}.pth.tar'.format(
                self.task, self.target, str(self.batch_size), str(nepoch), str(self.lr))))
        if fname:
            self.export(fname, pending)
        return self

    def test(self, database_test=None, threshold=4, hdf5='test_data.drgs'):
        # (without database_test the loaded dataset is tested, pretrained or not)
        ds = self.dataset if database_test is None else GraphDataSet(
            database_test, node_feature=self.node_feature, edge_feature=self.edge_feature, target=self.target,
            clustering_method=self.cluster_nodes or 'mcl')
        if database_test is not None and getattr(ds[0], "cluster0", None) is None:
            from .clustering import PreCluster
            PreCluster(ds, method=self.cluster_nodes or 'mcl', api=self._api, device=self.device)
        loss, store = self.eval(ds, list(range(len(ds))))
        self.data = {'test': store}
        self.test_loss = loss
        self.test_acc = self._accuracy(store, threshold) if store.has_targets() else None
        if hdf5:
            self.export(self.update_name(self._rank_name(hdf5), self.outdir), {'epoch_0000': self.data})
        return store

    def _rank_name(self, hdf5):
        """data parallel: every rank exports ITS shard's outputs; ranks > 0 under their own file name"""
        if getattr(self, "world", 1) > 1 and self.rank > 0:
            stem, ext = os.path.splitext(str(hdf5))
            return "%s.rank%d%s" % (stem, self.rank, ext)
        return hdf5

    @staticmethod
    def update_name(hdf5, outdir):
        """NeuralNet.update_name (NeuralNet.py:633-656): never overwrite an existing export, number the new one."""
        fname = os.path.join(outdir, hdf5)
        stem, ext = os.path.splitext(hdf5)
        count = 0
        while os.path.exists(fname):
            count += 1
            fname = os.path.join(outdir, '{}_{:03d}{}'.format(stem, count, ext))
        return fname

    def export(self, fname, epochs=None):
        """_export_epoch_hdf5 (NeuralNet.py:827-872): groups ``epoch_%04d/<pass>/{mol,outputs,targets,raw_outputs}`` and
        the group attributes task / target / batch_size -- as a native container whose tree mirror
        ``tools/native_to_hdf5.py`` turns into that HDF5 file (``.npz``: flat keys, no attributes)."""
        epochs = {'epoch_0000': self.data} if epochs is None else epochs
        flat = {}
        for grp, passes in epochs.items():
            for pass_type, store in passes.items():
                for k, v in store.items():
                    if k.startswith('_'):
                        continue
                    flat["%s/%s/%s" % (grp, pass_type, k)] = np.asarray(v, dtype='S') if k == 'mol' else np.asarray(v)
        if str(fname).endswith('.npz'):
            np.savez_compressed(fname, **flat)
        else:
            from .container import write_container
            attrs = {grp: {'task': self.task, 'target': self.target, 'batch_size': int(self.batch_size)} for grp in epochs}
            write_container(fname, {"tree/" + k: v for k, v in flat.items()},
                            meta={"kind": "tree", "mols": list(epochs), "attrs": attrs,
                                  "schema": "deeprank_gnn NeuralNet._export_epoch_hdf5 (reference NeuralNet.py:827-872)"})
        self.exported.append(fname)
        return fname

    def save_model(self, filename='model.pth.tar'):
        if getattr(self, "world", 1) > 1 and self.rank > 0:
            return                        # data parallel: the replicas are identical, rank 0 writes the checkpoint
        state = {'model': {k: v.detach().cpu().clone() for k, v in self.model.state_dict().items()},
                 'optimizer': self.trainer.optimizer_state_dict(),
                 'node': self.node_feature, 'edge': self.edge_feature, 'target': self.target, 'task': self.task,
                 'classes': self.classes, 'class_weight': self.class_weights, 'batch_size': self.batch_size,
                 'percent': self.percent, 'lr': self.lr, 'index': self.index, 'shuffle': self.shuffle,
                 'threshold': self.threshold, 'cluster_nodes': self.cluster_nodes,
                 'transform_sigmoid': self.transform_sigmoid}
        torch.save(state, filename)
