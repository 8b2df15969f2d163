"""Native training step: the whole per-mini-batch work of the reference trainer on the device
path, five kernel launches and no autograd graph.

Counterpart of the inner loop body of ``NeuralNet._epoch`` (reference NeuralNet.py:489-506)

    optimizer.zero_grad(); pred = model(batch); loss = loss_fn(pred, y); loss.backward();
    optimizer.step()

with the reference's choices: ``MSELoss`` for regression / ``CrossEntropyLoss(weight)`` for
classification (NeuralNet.py:239-263) and ``torch.optim.Adam(lr)`` (NeuralNet.py:183-184).

    topology  ->  body forward  ->  head (fc1/relu/dropout/fc2 + loss + its backward)
              ->  body backward ->  gradient reduction  [-> all-reduce]  ->  Adam

The model's parameters become views into one flat fp32 buffer (``state_dict`` / checkpoints
keep working), their ``.grad`` views into a second one, so the data-parallel exchange is one
all-reduce of one buffer (deeprank_gnn_amd.parallel) and Adam is one launch.
Everything is enqueued on torch's current stream and is hipGraph-capturable.
"""
import torch
import torch.distributed as dist

from . import _lib, hostcpu
from .functional import H1, H2, _describe, _fill_grads, _split
from .topology import Topology

__all__ = ["FusedTrainer"]


from .fused_autograd import net_layout as _net_layout      # noqa: E402  (kind, n_branch, convs) of a reference net


class _BatchView(object):
    """the two members of a batch _fused_prepare reads"""
    def __init__(self, x, y):
        self.x, self.y = x, y


class FusedTrainer(object):
    EPOCH_CHUNK = 128            # mini-batches per call of the native epoch loop (see _run_epoch)
    _dp_first_batch = 0

    def __init__(self, net, lr=0.01, task="reg", class_weights=None, betas=(0.9, 0.999), eps=1e-8,
                 weight_decay=0.0, seed=None, api=None, transform_sigmoid=False):
        hostcpu.fit_torch_threads()                           # the launching thread must not lose its CPU quota to idle pool threads
        self.net = net
        self.transform_sigmoid = bool(transform_sigmoid)      # regression: sigmoid on the output before the loss
        self.api = api or _lib.get()
        self.kind, self.n_branch, self.convs = _net_layout(net)
        self.task = _lib.TASK_REG if task == "reg" else _lib.TASK_CLASS
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        if seed is None:
            # dropout stream: follows torch.manual_seed (like the reference's F.dropout) and differs per rank, so that
            # data-parallel ranks do not draw the same masks for their different graphs
            rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
            seed = (torch.initial_seed() ^ (rank * 0x9E3779B1)) & 0xFFFFFFFF
        self.seed = int(seed) & 0xFFFFFFFF
        params = list(net.parameters())
        dev = params[0].device
        if self.api is _lib._API:
            _lib.require_device(*params)
        total = sum(p.numel() for p in params)
        self.flat_p = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        # [0] optimiser steps completed, [1] index of the step in flight (fused step launch writes it,
        # the update launch reads it for Adam and commits [0]); `step` is the public 1-element view
        # ([2]: sticky fault bits the fused step raises, see check_faults; [3] reserved)
        self.step2 = torch.zeros(4, dtype=torch.int32, device=dev)
        self.step = self.step2[:1]
        self.fused_step = True       # one launch for fwd + head + bwd whenever a graph fits LDS
        # overrides of the fused step's launch plan (drgnn_step_plan: force_wgs / no_class / no_aggregate / no_split /
        # no_paired; tests and A/B runs) -- per trainer, handed to every plan query and every launch
        self.plan_overrides = {}
        self._xchg = {}              # exchange words of the fused step, one grow-only buffer per batch size
        # what a co-built topology must hold: the hierarchical node order, read by the aggregation-first step kernels
        # (sGAT / FoutNet: every training launch; GINet: the two-workgroup layout only, see _flags_for)
        self.topo_flags = _lib.TOPO_HIER
        self._desc_cache, self._slab_cache = {}, {}
        self._epoch_bytes = {}        # scratch need of the native epoch loop by epoch shape (_run_epoch)
        self._ids_memo = {}           # (set, inference) -> (host order, its ids on the device) of the last such pass
        self._epoch_scratch = None
        self._loss_buf = torch.zeros(1, dtype=torch.float32, device=dev)
        self.offset = {}
        off = 0
        with torch.no_grad():
            for name, p in net.named_parameters():
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                p.grad = self.flat_g[off:off + n].view(p.shape)
                self.offset[name] = off
                off += n
        head = ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
        o0 = self.offset[head[0]]
        expect = o0
        for name in head:      # the head reducer writes one contiguous block
            assert self.offset[name] == expect, "unexpected parameter order for the FC head"
            expect += dict(net.named_parameters())[name].numel()
        self.head_grad_offset = o0
        self.R, self.H, self.O = net.fc1.in_features, net.fc1.out_features, net.fc2.out_features
        self.class_w = None
        if class_weights is not None:
            self.class_w = torch.as_tensor(class_weights, dtype=torch.float32, device=dev).contiguous()
        self.live = tuple(p for c in self.convs for p in c.live_parameters())
        self.live_grads = tuple(p.grad for p in self.live)

    # ---------------------------------------------------------------------------
    @property
    def loss(self):
        """[1] device tensor: the loss of the last training step, ONE fixed buffer (recorded hipGraphs keep its address; a
        reference to it stays current).  Every update launch writes it -- ``train_step`` directly, ``train_epoch`` through its
        last mini-batch's update launch (drgnn_epoch_plan.last_loss) -- on the stream of that launch: no lazy state."""
        return self._loss_buf

    def _head_desc(self, train):
        hd = _lib.HeadDesc()
        hd.R, hd.H, hd.O, hd.task, hd.train = self.R, self.H, self.O, self.task, int(train)
        hd.p_drop = float(getattr(self.net, "dropout", 0.0)) if train else 0.0
        hd.seed = self.seed
        hd.transform_sigmoid = int(self.transform_sigmoid and self.task == _lib.TASK_REG)
        n = self.net
        hd.w1, hd.b1 = n.fc1.weight.data_ptr(), n.fc1.bias.data_ptr()
        hd.w2, hd.b2 = n.fc2.weight.data_ptr(), n.fc2.bias.data_ptr()
        hd.class_w = None if self.class_w is None else self.class_w.data_ptr()
        # test hook (drgnn_head_desc.drop_mask): an explicit [B, H] 0 / 1 mask instead of the hash stream
        mask = getattr(self, "drop_mask", None) if train else None
        if mask is not None:
            assert mask.dtype == torch.float32 and mask.is_contiguous() and mask.shape[-1] == self.H
        hd.drop_mask = None if mask is None else mask.data_ptr()
        return hd

    def _body_forward(self, batch, topo, stream, step_inc=None):
        api = self.api
        x = batch.x.contiguous()
        n_nodes, n_feat = x.shape
        dev = x.device
        B = topo.n_graphs
        nb = self.n_branch
        xp = torch.empty((nb, n_nodes, H1), dtype=torch.float32, device=dev)
        arg0 = torch.empty((nb, n_nodes, H1), dtype=torch.int32, device=dev)
        arg1 = torch.empty((nb, n_nodes, H2), dtype=torch.int32, device=dev)
        readout = torch.empty((B, H2 * nb), dtype=torch.float32, device=dev)
        scratch = None
        need = max(api.net_lds_bytes(self.kind, n_feat, topo.max_nodes, topo.max_edges, topo.max_c0, b)
                   for b in (False, True))
        if topo.max_nodes == 0 or need > 160 * 1024:
            scratch = torch.empty(api.net_scratch_elems(self.kind, n_feat, n_nodes, topo.n_edges, B),
                                  dtype=torch.float32, device=dev)
        desc = _describe(self.kind, n_feat, self.live, nb)
        api.net_forward(desc, x, topo.ws_i32, topo.ws_f32, n_nodes, topo.n_edges, B, topo.max_nodes,
                        topo.max_edges, topo.max_c0, xp, arg0, arg1, readout, scratch, stream, step_inc=step_inc)
        return x, desc, xp, arg0, arg1, readout, scratch

    # -- launch plan ------------------------------------------------------------------------------------------------
    def _tiles_match(self, topo):
        """The aggregation tiles of a workspace built WITH edge weights are weighted sums (what sGAT starts from); GINet /
        FoutNet start from plain sums: a workspace of the other flavour is stepped without its tiles."""
        return (getattr(topo, "ws_f32", None) is not None) == (self.kind == _lib.SGAT)

    def _usable_flags(self, topo, x=None):
        """The TOPO_* flags of ``topo`` as a launch of this net may rely on them: TILES only with tiles of this kind's
        flavour that were formed from the ``x`` the launch steps (a Topology bakes the neighbour sums of its ``x`` in at
        build time: another tensor, or the same one modified in place since, makes them stale) in 16-byte aligned memory."""
        flags = int(getattr(topo, "flags", 0))
        tiles = getattr(topo, "tiles", None)
        ok = tiles is not None and (flags & _lib.TOPO_TILES) and self._tiles_match(topo)
        if ok and x is not None:
            tx = getattr(topo, "x", None)
            ok = (tx is not None and tx.data_ptr() == x.data_ptr() and tuple(tx.shape) == tuple(x.shape) and
                  (x.shape[1] % 4 != 0 or x.data_ptr() % 16 == 0) and getattr(topo, "_tiles_x_version", None) == x._version)
        if not ok:
            flags &= ~_lib.TOPO_TILES
        return flags

    def _plan(self, n_feat, max_nodes, max_edges, max_c0, B, co=0, train=True, topo_flags=0):
        return self.api.step_plan(self.kind, n_feat, max_nodes, max_edges, max_c0, self.R, self.H, self.O, B, co, train,
                                  topo_flags, self.plan_overrides)

    def _plan_for(self, topo, n_feat, next_topo=None, train=True, x=None, flags=None):
        return self._plan(n_feat, topo.max_nodes, topo.max_edges, topo.max_c0, topo.n_graphs,
                          0 if next_topo is None else next_topo.n_graphs, train,
                          self._usable_flags(topo, x) if flags is None else flags)

    def _can_fuse(self, topo, n_feat, next_topo=None, train=True, x=None):
        """True when the launch on ``topo`` (what it holds NOW, for the ``x`` given) is one of the fused step kernels --
        judged by the plan of exactly that launch (kernel family, layout and LDS need: drgnn_net_step_plan)."""
        if not self.fused_step or topo.max_nodes <= 0:
            return False
        p = self._plan_for(topo, n_feat, next_topo, train, x)
        return p.family != _lib.STEP_FAMILY_NONE and 0 < p.lds_bytes <= 160 * 1024

    def _xchg_for(self, plan, B, dev):
        """Exchange words of a launch with this plan: ONE buffer per batch size, grown to the largest need seen (the words
        carry the step index as a tag, so stale ones are harmless; the launch is told the stride through its bounds)."""
        words = int(plan.xchg_words)
        if words <= 0 and self.n_branch == 1:
            return None
        words = max(words, self.n_branch * max(self.H, 32))
        buf = self._xchg.get(B)
        if buf is None or buf.shape[1] < words:
            buf = self._xchg[B] = torch.zeros((max(B, 1), words), dtype=torch.int64, device=dev)
        return buf

    def _step_buffers(self, plan, n_feat, B, dev):
        """descriptors (per feature width) and slabs (per batch size and layout) of a fused step"""
        nb = self.n_branch
        ck = self._desc_cache.get(n_feat)
        if ck is None:
            g1 = (_lib.ConvGrads * _lib.MAX_BRANCH)()
            g2 = (_lib.ConvGrads * _lib.MAX_BRANCH)()
            for b, (l1, l2) in enumerate(_split(self.kind, self.live_grads, nb)):
                _fill_grads(g1[b], self.kind, l1, n_feat, H1)
                _fill_grads(g2[b], self.kind, l2, H1, H2)
            ck = self._desc_cache[n_feat] = (g1, g2, _describe(self.kind, n_feat, self.live, nb))
        slabs = max(int(plan.slabs_per_graph), nb)
        bk = self._slab_cache.get((B, n_feat, slabs))
        if bk is None:
            bk = self._slab_cache[(B, n_feat, slabs)] = (
                torch.empty((B, H2 * nb), dtype=torch.float32, device=dev),
                torch.empty((max(B * slabs, 1), self.api.net_partial_elems(self.kind, n_feat)), dtype=torch.float32, device=dev),
                torch.empty((max(B, 1), self.api.head_compact_elems(self.R, self.H, self.O)), dtype=torch.float32, device=dev))
        return ck, bk, slabs

    def _fused_prepare(self, batch, topo, train=True, next_topo=None):
        """Buffers and descriptors of one fused step (allocation only, no launch)."""
        x = batch.x.contiguous()
        n_nodes, n_feat = x.shape
        dev = x.device
        B = topo.n_graphs
        y = getattr(batch, "y", None)
        if y is not None:
            y = y.to(torch.float32).contiguous() if self.task == _lib.TASK_REG else y.to(torch.int64).contiguous()
        topo_flags = self._usable_flags(topo, x)
        if (int(getattr(topo, "flags", 0)) & _lib.TOPO_TILES) and not (topo_flags & _lib.TOPO_TILES) and \
                self._tiles_match(topo) and getattr(topo, "tiles", None) is not None and \
                getattr(topo, "_inputs", None) is not None and \
                (x.shape[1] % 4 != 0 or x.data_ptr() % 16 == 0) and tuple(x.shape) == tuple(topo.x.shape):
            # the tiles were formed from other node features than the ones being stepped (x replaced or modified in place
            # since the build): form them again from this x (own launch, same stream)
            topo.x = x
            topo.rebuild()
            topo_flags = self._usable_flags(topo, x)
        plan = self._plan_for(topo, n_feat, next_topo, train, x, flags=topo_flags)
        (g1, g2, desc), (readout, partials, hp), slabs = self._step_buffers(plan, n_feat, B, dev)
        xchg = self._xchg_for(plan, B, dev)
        # host copies of the mini-batch's offsets (Batch.from_data_list / the resident set record them): they travel in
        # the launch arguments, so a workgroup need not fetch them from the workspace first
        bd = getattr(batch, "__dict__", {})
        hn, he = bd.get("_host_node_ptr"), bd.get("_host_edge_ptr")
        tiles = getattr(topo, "tiles", None) if (topo_flags & _lib.TOPO_TILES) else None
        if hn is not None and he is not None and len(hn) == B + 1 and B <= 64:
            hints = _lib.step_hints(node_ptr=hn, edge_ptr=he, topo_flags=topo_flags, tiles=tiles, plan=plan)
        else:
            hints = _lib.step_hints(topo_flags=topo_flags, tiles=tiles, plan=plan)
        return dict(
            hints=hints, slabs=slabs, plan=plan,
            x=x, y=y, topo=topo, B=B, n_nodes=n_nodes, xchg=xchg, g1=g1, g2=g2, desc=desc,
            stream=_lib.current_stream(x), pred=torch.empty((B, self.O), dtype=torch.float32, device=dev),
            readout=readout, partials=partials, hp=hp)

    def _flags_for(self, topo, n_feat, train=True):
        """Request flags of a topology the next launch co-builds.  When the launch that will step it is one of the
        aggregation-first kernels (judged by its plan, for a following mini-batch of the same size): the hierarchical node
        order, the aggregation tiles, and nothing those kernels do not read (TOPO_LEAN: the builder's short chains);
        _fused_launch_step rebuilds in full should that turn out wrong.  Otherwise the plain build."""
        if getattr(topo, "tiles", None) is not None and self._tiles_match(topo) and not getattr(topo, "_tiles_separately", False):
            af = _lib.TOPO_HIER | _lib.TOPO_LEAN | _lib.TOPO_TILES
            if self._plan(n_feat, topo.max_nodes, topo.max_edges, topo.max_c0, topo.n_graphs, topo.n_graphs, train, af).lean_ok:
                return af
        return 0 if self.kind == _lib.GINET else _lib.TOPO_HIER

    def _fused_launch_step(self, c, next_topo=None):
        """ONE launch: body fwd + head/loss + body bwd (+ the next mini-batch's topology)."""
        t = c["topo"]
        if (int(getattr(t, "flags", 0)) & _lib.TOPO_LEAN) and not c["plan"].lean_ok:
            # a lean workspace under a launch that reads more: build the rest (own launch, same stream), plan again
            t.rebuild()
            c.update(self._fused_prepare(_BatchView(c["x"], c["y"]), t, True, next_topo))
        self.api.net_train_step(c["desc"], self._head_desc(True), c["x"], c["y"], self.step2, t.ws_i32, t.ws_f32,
                                c["n_nodes"], t.n_edges, c["B"], t.max_nodes, t.max_edges, t.max_c0, c["pred"],
                                c["readout"], c["hp"], c["partials"], c["xchg"], c["stream"],
                                next_topology=None if next_topo is None else next_topo.request(self._flags_for(next_topo, c["x"].shape[1])),
                                hints=None if c.get("hints") is None else c["hints"][0])

    def _fused_launch_update(self, c, apply_adam=True, lr=None):
        """Second launch: fixed-order reduction of the slabs (+ dW_fc1 = dhid^T readout) and Adam."""
        self.api.step_update(c["desc"], c["partials"], c["B"], c["g1"], c["g2"], c["hp"], c["readout"], self.R,
                             self.H, self.O, self.head_grad_offset, self.flat_p, self.flat_g, self.exp_avg,
                             self.exp_avg_sq, self.step2, self.loss, self.lr if lr is None else lr,
                             self.betas[0], self.betas[1], self.eps, c["stream"], apply_adam=apply_adam,
                             slabs_per_graph=c.get("slabs", 0))

    def _fused(self, batch, topo, apply_adam, next_topo):
        c = self._fused_prepare(batch, topo, True, next_topo)
        self._fused_launch_step(c, next_topo)
        self._fused_launch_update(c, apply_adam)
        self.last_pred = c["pred"]
        self.last_batch_size = c["B"]
        return self.loss

    # -- cached topology (declared mode): a mini-batch is a list of graph numbers of a resident set ---------------
    def _cached_prepare(self, cache, ids, ids_dev=None, train=True, next_ids_dev=None):
        """Buffers of one step over the graphs ``ids`` (host numbers; ``ids_dev``: the same as int32 on the device) of
        ``cache`` (resident.TopologyCache).  ``next_ids_dev``: the NEXT mini-batch's graph numbers (device int32): spare
        workgroups of this launch prefetch them (drgnn_step_hints.next_ids)."""
        import numpy as np
        api = self.api
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        B = int(ids.size)
        gset = cache.set
        if ids_dev is None:
            ids_dev = gset.upload_ids(ids)
        n_feat, dev, nb = gset.n_feat, gset.device, self.n_branch
        if self.kind == _lib.SGAT and not cache.with_weights:
            raise ValueError("sGAT needs a topology cache built with edge weights (topology_cache(need_weights=True))")
        max_nodes, max_edges, max_c0 = cache.bounds(ids)
        topo_flags = int(getattr(cache.topo, "flags", 0))
        tiles = cache.tiles_for(self.kind == _lib.SGAT) if (topo_flags & _lib.TOPO_TILES) else None
        if tiles is None or (n_feat % 4 == 0 and gset.x.data_ptr() % 16 != 0):
            topo_flags &= ~_lib.TOPO_TILES
            tiles = None
        plan = self._plan(n_feat, max_nodes, max_edges, max_c0, B, 0, train, topo_flags)
        if not (plan.family != _lib.STEP_FAMILY_NONE and 0 < plan.lds_bytes <= 160 * 1024):
            raise _lib.DrgnnError("a graph of this mini-batch does not fit the fused step kernel's LDS budget")
        (g1, g2, desc), (readout, partials, hp), slabs = self._step_buffers(plan, n_feat, B, dev)
        xchg = self._xchg_for(plan, B, dev)
        # (beyond 64 graphs the offsets no longer travel in the kernel arguments, but the library still range-checks the ids)
        hints = _lib.step_hints(set_node_ptr=gset.node_ptr, set_edge_ptr=gset.edge_ptr, ids=ids,
                                topo_flags=topo_flags, tiles=tiles, plan=plan, next_ids=next_ids_dev)
        return dict(hints=hints, slabs=slabs, plan=plan,
                    cache=cache, ids_dev=ids_dev, B=B, bounds=(max_nodes, max_edges, max_c0), xchg=xchg, g1=g1, g2=g2,
                    desc=desc, stream=_lib.current_stream(gset.x), readout=readout, partials=partials, hp=hp,
                    pred=torch.empty((B, self.O), dtype=torch.float32, device=dev))

    def _cached_launch_step(self, c, train=True):
        mn, me, mc = c["bounds"]
        self.api.net_train_step_cached(c["desc"], self._head_desc(train), c["cache"].desc_for(self.kind == _lib.SGAT), c["ids_dev"], c["B"], mn, me, mc,
                                       self.step2, c["pred"], c["readout"], c["hp"] if train else None,
                                       c["partials"] if train else None, c["xchg"], c["stream"],
                                       hints=None if c.get("hints") is None else c["hints"][0])

    def train_step_cached(self, cache, ids, ids_dev=None, apply_adam=True, next_ids_dev=None):
        """One optimisation step on the graphs ``ids`` of a cached set: the fused step launch reading the cached
        topology in place + the update launch.  Same arithmetic as ``train_step`` on the collated mini-batch."""
        c = self._cached_prepare(cache, ids, ids_dev, True, next_ids_dev)
        want = torch.float32 if self.task == _lib.TASK_REG else torch.int64
        if cache.set.y is None or cache.set.y.dtype != want:
            raise ValueError("the set's targets must be %s for this task" % want)
        self._cached_launch_step(c, True)
        self._fused_launch_update(c, apply_adam)
        self.last_pred = c["pred"]
        self.last_batch_size = c["B"]
        return self.loss

    @torch.no_grad()
    def predict_cached(self, cache, ids, ids_dev=None):
        c = self._cached_prepare(cache, ids, ids_dev, train=False)
        self._cached_launch_step(c, False)
        return c["pred"]

    def _topology_of(self, batch, train):
        """The workspace of a mini-batch nobody built one for.  Graphs the builder stages no x tile for
        (``Topology._tiles_separately``) get their aggregation tiles from the stand-alone launch when a fused kernel will read
        them (the from-memory instances: drgnn_step_plan.from_memory)."""
        topo = Topology.from_batch(batch, api=self.api, need_weights=(self.kind == _lib.SGAT), build=False)
        flags = topo.full_flags()
        if getattr(topo, "_tiles_separately", False) and topo.tiles is not None and self.fused_step and topo.max_nodes > 0:
            with_tiles = flags | _lib.TOPO_TILES
            p = self._plan(int(batch.x.shape[1]), topo.max_nodes, topo.max_edges, topo.max_c0, topo.n_graphs, 0, train, with_tiles)
            if p.family == _lib.STEP_FAMILY_AGGREGATE:
                flags = with_tiles
        return topo.rebuild(flags)

    def _backward(self, batch, topo, apply_adam, next_topo=None):
        """Fused step when every graph fits LDS (see _fused); else fwd (++step) -> bwd with the per-graph
        head + loss inside (and, when given, the NEXT mini-batch's topology build sharing that launch)
        -> fixed-order reduction of the partials, with Adam applied in the same launch when
        ``apply_adam``."""
        api = self.api
        if topo is None:
            topo = self._topology_of(batch, True)
        if self._can_fuse(topo, batch.x.shape[1], next_topo, True, batch.x):
            return self._fused(batch, topo, apply_adam, next_topo)
        if int(getattr(topo, "flags", 0)) & _lib.TOPO_LEAN:
            topo.rebuild()       # the launch pair reads CSC0 and the member lists a lean build leaves out
        stream = _lib.current_stream(batch.x)
        x, desc, xp, arg0, arg1, readout, scratch = self._body_forward(batch, topo, stream, step_inc=self.step)
        B = topo.n_graphs
        dev = x.device
        n_nodes, n_feat = x.shape
        pred = torch.empty((B, self.O), dtype=torch.float32, device=dev)
        y = batch.y
        y = y.to(torch.float32).contiguous() if self.task == _lib.TASK_REG else y.to(torch.int64).contiguous()
        partials = torch.empty((max(B * self.n_branch, 1), api.net_partial_elems(self.kind, n_feat)),
                               dtype=torch.float32, device=dev)
        g1 = (_lib.ConvGrads * _lib.MAX_BRANCH)()
        g2 = (_lib.ConvGrads * _lib.MAX_BRANCH)()
        for b, (l1, l2) in enumerate(_split(self.kind, self.live_grads, self.n_branch)):
            _fill_grads(g1[b], self.kind, l1, n_feat, H1)
            _fill_grads(g2[b], self.kind, l2, H1, H2)
        hp = torch.empty((max(B, 1), api.head_partial_elems(self.R, self.H, self.O)),
                         dtype=torch.float32, device=dev)
        api.net_backward_fused_head(desc, self._head_desc(True), x, readout, y, self.step, topo.ws_i32,
                                    topo.ws_f32, n_nodes, topo.n_edges, B, topo.max_nodes, topo.max_edges,
                                    topo.max_c0, xp, arg0, arg1, pred, hp, None, partials, scratch, stream,
                                    next_topology=None if next_topo is None else next_topo.request(self.topo_flags))
        api.train_update(desc, partials, B, g1, g2, hp, self.R, self.H, self.O, self.head_grad_offset,
                         self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.step, self.loss,
                         self.lr, self.betas[0], self.betas[1], self.eps, stream, apply_adam=apply_adam)
        self.last_pred = pred
        self.last_batch_size = B
        return self.loss

    def compute_gradients(self, batch, topo=None, next_topo=None):
        """Everything of a step except the parameter update: leaves d(mean loss over THIS
        batch)/d(params) in ``flat_g`` (= every ``p.grad``), the loss in ``self.loss``, predictions in
        ``self.last_pred``; the step counter already counts this step."""
        return self._backward(batch, topo, False, next_topo)

    def all_reduce_gradients(self, n_local=None, n_global=None, group=None):
        """Data parallel exchange: ONE all-reduce of the flat gradient buffer (RCCL over xGMI with
        the nccl backend).  Each rank holds the gradient of the mean loss over its own shard, so
        it is weighted by n_local / n_global (equal shards: 1 / world)."""
        if not (dist.is_available() and dist.is_initialized()):
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        oneshot = getattr(self, "_oneshot", None)
        if oneshot is not None:
            # opt-in: one launch, one xGMI round trip (parallel.OneShotAllReduce) instead of the RCCL ring
            w = (float(n_local if n_local is not None else self.last_batch_size) / float(n_global)) if n_global else None
            oneshot(self.flat_g, weight=w)
            return
        if n_global:
            self.flat_g.mul_(float(n_local if n_local is not None else self.last_batch_size) / float(n_global))
        elif dist.get_backend(group) == "nccl" and getattr(self, "_avg_ok", True):
            # equal shards: RCCL averages inside the collective (one launch less per step)
            try:
                dist.all_reduce(self.flat_g, op=dist.ReduceOp.AVG, group=group)
                return
            except (RuntimeError, ValueError):      # a build without ncclAvg: scale + sum from now on
                self._avg_ok = False
                self.flat_g.mul_(1.0 / world)
        else:
            self.flat_g.mul_(1.0 / world)
        dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=group)

    def use_oneshot_allreduce(self, group=None, verify=True):
        """Route ``all_reduce_gradients`` through the one-shot peer-to-peer exchange (parallel.OneShotAllReduce).
        ``verify``: run a few exchanges of known vectors first and agree over the process group that every rank got the
        right sums without an expired wait; on any failure (no peer access, hipIpc refused, a wait expired, a wrong
        sum) the trainer stays on the collective library's all-reduce.  Returns the exchanger or None."""
        from .parallel import OneShotAllReduce
        self._oneshot = None
        ok, ar = 1, None
        try:
            ar = OneShotAllReduce(self.flat_g.numel(), self.flat_g.device, api=self.api, group=group)
            if verify:
                world, rank = ar.world, ar.rank
                for it in range(3):
                    v = torch.full((ar.n,), float(rank + 1 + it), dtype=torch.float32, device=self.flat_g.device)
                    v[::7] += 0.25 * rank
                    ar(v, weight=1.0)
                    if v.is_cuda:
                        torch.cuda.synchronize(v.device)
                    want = sum(float(r + 1 + it) for r in range(world))
                    want7 = want + 0.25 * sum(range(world))
                    chk = v.cpu()
                    good = bool(torch.all(chk[1::7] == want)) and bool(torch.all(chk[::7] == want7))
                    if not good or int(ar.status.item()) != 0:
                        ok = 0
                        break
        except Exception:                      # pragma: no cover - depends on the node
            ok = 0
        if verify and dist.is_available() and dist.is_initialized():
            flag = torch.tensor([ok], dtype=torch.int32, device=self.flat_g.device if dist.get_backend(group) == "nccl" else "cpu")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
            ok = int(flag.item())
        if ok:
            self._oneshot = ar
        elif ar is not None:
            ar.close()               # the trial failed somewhere: release the exchange buffer and the peers' mappings
        return self._oneshot

    def apply_update(self):
        """Adam on the flat buffers (one launch)."""
        self.api.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.step, self.lr,
                           self.betas[0], self.betas[1], self.eps, self.weight_decay,
                           _lib.current_stream(self.flat_p))

    def train_step(self, batch, topo=None, n_global=None, group=None, next_topo=None, n_local=None):
        """One optimisation step on ``batch``; returns the (device) loss of this rank's shard.
        ``next_topo``: an allocated-but-unbuilt ``Topology`` of the NEXT mini-batch
        (``Topology.from_batch(next_batch, build=False)``): it is built inside this step's backward
        launch (independent work sharing the launch), so the next step starts without a builder
        launch.
        Single process: 3 launches (+1 for the topology unless it was co-built by the previous step):
        body fwd, body bwd incl. head + loss, reduce+Adam.  With torch.distributed initialised:
        reduce, ONE all-reduce of the flat gradient, Adam."""
        distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
        if not distributed and self.weight_decay == 0.0:
            return self._backward(batch, topo, True, next_topo)
        loss = self.compute_gradients(batch, topo, next_topo)
        # n_local: this rank's weight in the global mini-batch when it differs from the batch it stepped (0 for a rank
        # that has no graph of a mini-batch smaller than the world and steps a stand-in to keep the collectives matched)
        self.all_reduce_gradients(n_local=n_local, n_global=n_global, group=group)
        self.apply_update()
        return loss

    def broadcast_state(self, src=0, group=None):
        """Data parallel start: every replica takes rank ``src``'s parameters, Adam moments and step counter, so that the
        replicas are one model whatever the ranks' RNG seeds were."""
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
            return
        on_dev = dist.get_backend(group) == "nccl"
        for t in (self.flat_p, self.exp_avg, self.exp_avg_sq, self.step2):
            buf = t if on_dev else t.detach().cpu()
            dist.broadcast(buf, src=src, group=group)
            if not on_dev:
                t.copy_(buf)

    def faults(self):
        """Sticky fault bits raised by the fused-step kernels since the last reset (synchronises): bit
        ``_lib.FAULT_EXCHANGE`` = a GINet branch workgroup waited in vain for its partner's half of fc1 (its outputs and
        that step's loss are NaN)."""
        return int(self.step2[2])

    def check_faults(self):
        """Raise if a kernel has reported a fault (synchronises)."""
        self.raise_on_faults(self.faults())

    def raise_on_faults(self, bits):
        """``bits``: the fault word as read by the caller (NeuralNet copies it to the host behind every epoch)."""
        if bits:
            self.step2[2] = 0
            raise _lib.DrgnnError("fused training step reported fault bits 0x%x%s" % (
                bits, ": the branch workgroups of a graph did not meet (exchange wait expired); the losses of this epoch "
                      "are NaN -- reduce the batch size or set trainer.fused_step = False" if bits & _lib.FAULT_EXCHANGE else ""))

    def predict_epoch(self, gset, order, batch_size, cached=False):
        """Inference over the graphs ``order`` of the resident set, native loop (one launch per mini-batch, no
        synchronisation): pred [len(order), O] on the device, or None when a graph needs the per-batch path."""
        done = self._run_epoch(gset, order, batch_size, inference=True, cached=cached)
        return None if done is None else done[1]

    def train_epoch(self, gset, order, batch_size, cached=False, dp_global_sizes=None, group=None, probe=False):
        """A whole epoch over the resident set ``gset`` (resident.ResidentGraphSet) in visiting order ``order``
        (graph numbers), driven by the native loop ``drgnn_train_epoch``: per mini-batch the fused step launch
        (whose extra workgroups build the next mini-batch's topology and gather its node rows straight from the
        resident set) and the update launch -- no Python and no host synchronisation between mini-batches.  Returns (losses [n_batches], pred [len(order), O]) as
        device tensors, or None when this configuration needs the per-batch path (data parallel, weight decay,
        a graph too large for the fused kernels)."""
        if self.weight_decay != 0.0:
            return None
        want = torch.float32 if self.task == _lib.TASK_REG else torch.int64
        if gset.y is None or gset.y.dtype != want:
            return None
        self._dp = None
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            # data parallel: ``order`` is THIS rank's shard of the epoch; per mini-batch the loop enqueues the gradient
            # launches, calls back here for the ONE all-reduce of the flat gradient (enqueued on the same stream, no
            # synchronisation), then enqueues Adam.  ``dp_global_sizes[k]``: graphs of mini-batch k over ALL ranks
            # (None: equal shards); every rank must run the same number of mini-batches.
            self._dp = (dp_global_sizes, group)
        # probe: only answer whether the native loop takes this configuration (True / None), nothing is launched -- data
        # parallel callers agree on the answer over ALL ranks before any of them enters the loop's collectives
        return self._run_epoch(gset, order, batch_size, inference=False, cached=cached, probe=probe)

    def _run_epoch(self, gset, order, batch_size, inference, cached=False, probe=False):
        """``cached``: step the mini-batches out of the set's topology cache (built once, ``gset.topology_cache``):
        no builder / offset / gather work in the loop (declared mode; the default rebuilds every mini-batch's topology
        like the reference does in every forward pass)."""
        import ctypes
        import numpy as np
        if not self.fused_step or not (gset.has_c0 and gset.has_c1):
            return None
        need_w = self.kind == _lib.SGAT
        if need_w and gset.edge_attr is None:
            return None
        if torch.is_tensor(order):      # (a Python list of 10^4 numbers costs ~1 ms to convert: as long as 40 mini-batches)
            order = order.detach().to(device="cpu", dtype=torch.int32).numpy()
        ids_host = np.ascontiguousarray(np.asarray(order, dtype=np.int32).reshape(-1))
        n = int(ids_host.size)
        dev = self.flat_p.device
        nb = (n + batch_size - 1) // batch_size
        # (training: every entry is written by its mini-batch's update launch; a fill launch would cost the epoch 4.7 us)
        losses = (torch.zeros if inference or n == 0 else torch.empty)((nb,), dtype=torch.float32, device=dev)
        pred = torch.empty((n, self.O), dtype=torch.float32, device=dev)
        if n == 0:
            return losses, pred
        # (a pass that visits the graphs in the order of the previous such pass -- every validation / test pass does -- reuses
        # that pass's ids on the device: an upload costs the host ~75 us and the device a cross-stream event)
        memo = self._ids_memo.get((id(gset), bool(inference)))
        if memo is not None and memo[0].size == ids_host.size and np.array_equal(memo[0], ids_host):
            ids_dev = memo[1]
        else:
            ids_dev = gset.upload_ids(ids_host)
            self._ids_memo[(id(gset), bool(inference))] = (ids_host.copy(), ids_dev)
        n_feat = gset.n_feat
        ck = self._desc_cache.get(n_feat)
        if ck is None:
            g1 = (_lib.ConvGrads * _lib.MAX_BRANCH)()
            g2 = (_lib.ConvGrads * _lib.MAX_BRANCH)()
            for b, (l1, l2) in enumerate(_split(self.kind, self.live_grads, self.n_branch)):
                _fill_grads(g1[b], self.kind, l1, n_feat, H1)
                _fill_grads(g2[b], self.kind, l2, H1, H2)
            ck = self._desc_cache[n_feat] = (g1, g2, _describe(self.kind, n_feat, self.live, self.n_branch))
        g1, g2, desc = ck
        head = self._head_desc(not inference)
        vp = ctypes.c_void_p
        plan = _lib.EpochPlan()
        plan.inference = int(inference)
        plan.set = ctypes.cast(ctypes.pointer(gset._desc), vp)
        plan.host_node_ptr, plan.host_edge_ptr = gset.node_ptr.ctypes.data, gset.edge_ptr.ctypes.data
        plan.host_c1_ptr = gset.c1_ptr.ctypes.data
        plan.ids, plan.host_ids, plan.n_ids = ids_dev.data_ptr(), ids_host.ctypes.data, n
        plan.batch_size, plan.need_weights = int(batch_size), int(need_w)
        plan.net = ctypes.cast(ctypes.pointer(desc), vp)
        plan.head = ctypes.cast(ctypes.pointer(head), vp)
        plan.g_conv1, plan.g_conv2 = ctypes.cast(g1, vp), ctypes.cast(g2, vp)
        plan.head_offset = self.head_grad_offset
        plan.flat_param, plan.flat_grad = self.flat_p.data_ptr(), self.flat_g.data_ptr()
        plan.exp_avg, plan.exp_avg_sq, plan.n_param = self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.flat_p.numel()
        plan.step2 = self.step2.data_ptr()
        plan.lr, plan.beta1, plan.beta2, plan.eps = self.lr, self.betas[0], self.betas[1], self.eps
        if cached:
            cache = gset.topology_cache(need_weights=need_w)
            plan.cache = ctypes.cast(ctypes.pointer(cache.desc_for(self.kind == _lib.SGAT)), vp)
        ov = None
        if self.plan_overrides:
            ov = _lib.StepPlan()
            for key, val in self.plan_overrides.items():
                if key not in _lib.StepPlan.OVERRIDES:
                    raise KeyError("unknown step-plan override %r" % (key,))
                setattr(ov, key, int(val))
            plan.step_overrides = ctypes.addressof(ov)
        callback = None
        if not inference and getattr(self, "_dp", None) is not None:
            sizes, group = self._dp

            def exchange(user, k, n_local, stream):
                try:
                    k = int(k) + self._dp_first_batch       # (k counts inside the piece of the epoch being enqueued)
                    self.all_reduce_gradients(n_local=int(n_local), n_global=(None if sizes is None else int(sizes[k])),
                                              group=group)
                    return 0
                except Exception as exc:          # never let an exception cross the C frame
                    self._dp_error = exc
                    return -1
            callback = _lib.EXCHANGE_FN(exchange)
            plan.exchange = ctypes.cast(callback, vp)
        # The plan's scratch need is a walk over every mini-batch's graphs on the host (~0.2 ms per 64 mini-batches: a fifth of the
        # time the device takes for them) and the loop itself repeats that walk: an epoch of the same shape as an earlier one
        # (same set, sizes, mode) reuses that epoch's answer -- drgnn_train_epoch checks the scratch against what THIS order needs
        # and refuses (DRGNN_E_CAPACITY) if a differently composed mini-batch needs more: the need is then computed afresh.
        shape_key = (id(gset), n, int(batch_size), bool(cached), bool(inference), n_feat, callback is not None,
                     tuple(sorted(self.plan_overrides.items())))
        known = self._epoch_bytes.get(shape_key)
        nbytes = known if (known is not None and not probe) else self.api.train_epoch_scratch_bytes(plan)
        if nbytes is None:
            return None
        if probe:
            return True
        self._epoch_bytes[shape_key] = max(nbytes, known or 0)
        scratch = self._epoch_scratch
        if scratch is None or scratch.numel() < nbytes:
            scratch = self._epoch_scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._dp_error = None
        stream = _lib.current_stream(self.flat_p)
        # Long epochs go to the native loop in pieces of EPOCH_CHUNK mini-batches, at most two pieces ahead of the device:
        # a host that enqueues thousands of launches ahead of the GPU saturates the runtime's queue and every launch
        # gets SLOWER (measured: 1024 mini-batches enqueued at once ran at 31 - 41 us per mini-batch against 22 - 25 us
        # for 64 at a time).  The pieces are stream-ordered like the whole would be; only the host waits (on an event
        # two pieces back), never the device.
        chunk = self.EPOCH_CHUNK
        in_flight = []
        try:
            for c0 in range(0, nb, chunk):
                c1 = min(nb, c0 + chunk)
                lo, hi = c0 * batch_size, min(n, c1 * batch_size)
                if c0 > 0 or c1 < nb:
                    plan.ids = ids_dev.data_ptr() + 4 * lo
                    plan.host_ids = ids_host.ctypes.data + 4 * lo
                    plan.n_ids = hi - lo
                    self._dp_first_batch = c0
                # the last mini-batch of the LAST piece writes the trainer's loss word as well
                plan.last_loss = self._loss_buf.data_ptr() if (c1 == nb and not inference) else None
                if dev.type == "cuda" and len(in_flight) >= 2:
                    in_flight.pop(0).synchronize()
                try:
                    self.api.train_epoch(plan, scratch, pred[lo:], losses[c0:], stream)
                except _lib.DrgnnError:
                    if known is None or self._dp_error is not None:
                        raise
                    # a remembered scratch size that this order outgrows (nothing of this piece was launched; earlier pieces
                    # keep their stream order): size it for THIS epoch's order and go again
                    known = None
                    piece = (plan.ids, plan.host_ids, plan.n_ids)
                    plan.ids, plan.host_ids, plan.n_ids = ids_dev.data_ptr(), ids_host.ctypes.data, n
                    nbytes = self.api.train_epoch_scratch_bytes(plan)
                    plan.ids, plan.host_ids, plan.n_ids = piece
                    if nbytes is None:
                        self._epoch_bytes.pop(shape_key, None)
                        if c0 == 0:
                            return None      # (nothing was launched: the caller steps this epoch mini-batch by mini-batch)
                        raise
                    self._epoch_bytes[shape_key] = nbytes
                    if scratch.numel() < nbytes:
                        scratch = self._epoch_scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                    self.api.train_epoch(plan, scratch, pred[lo:], losses[c0:], stream)
                if dev.type == "cuda" and c1 < nb:
                    ev = torch.cuda.Event()
                    ev.record()
                    in_flight.append(ev)
        except _lib.DrgnnError:
            if self._dp_error is not None:
                raise self._dp_error
            raise
        finally:
            self._dp_first_batch = 0
        del callback, ov
        if not inference:
            self.last_pred = pred[(nb - 1) * batch_size:]
            self.last_batch_size = n - (nb - 1) * batch_size
        return losses, pred

    # -- torch.optim.Adam compatible optimiser state --------------------------------------
    def optimizer_state_dict(self):
        """Same structure as ``torch.optim.Adam(...).state_dict()`` over ``net.parameters()`` (what the
        reference stores under 'optimizer', NeuralNet.py:776), so either side can resume the other."""
        state = {}
        step = int(self.step)
        for i, (name, p) in enumerate(self.net.named_parameters()):
            off, n = self.offset[name], p.numel()
            if step > 0:
                state[i] = {'step': torch.tensor(float(step)),
                            'exp_avg': self.exp_avg[off:off + n].view(p.shape).detach().cpu().clone(),
                            'exp_avg_sq': self.exp_avg_sq[off:off + n].view(p.shape).detach().cpu().clone()}
        group = {'lr': self.lr, 'betas': tuple(self.betas), 'eps': self.eps, 'weight_decay': self.weight_decay,
                 'amsgrad': False, 'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False,
                 'fused': None, 'params': list(range(len(self.offset)))}
        return {'state': state, 'param_groups': [group]}

    def load_optimizer_state_dict(self, sd):
        group = sd['param_groups'][0]
        self.lr, self.betas, self.eps = float(group['lr']), tuple(group['betas']), float(group['eps'])
        self.weight_decay = float(group.get('weight_decay', 0.0))
        steps = set()
        with torch.no_grad():
            for i, (name, p) in enumerate(self.net.named_parameters()):
                st = sd['state'].get(i)
                if st is None:
                    continue
                off, n = self.offset[name], p.numel()
                self.exp_avg[off:off + n].copy_(st['exp_avg'].reshape(-1).to(self.exp_avg.device))
                self.exp_avg_sq[off:off + n].copy_(st['exp_avg_sq'].reshape(-1).to(self.exp_avg.device))
                steps.add(int(float(st['step'])))
            if steps:
                self.step.fill_(max(steps))
            for buf in self._xchg.values():      # tags of an earlier run must not match again
                buf.zero_()

    @torch.no_grad()
    def predict(self, batch, topo=None, next_topo=None):
        """Inference (dropout off), fully on the native path; returns pred [B, O].  One launch (the fused step
        kernel in inference mode) when the graphs fit its LDS budget, else body forward + head.  ``next_topo``
        (``Topology.from_batch(next_batch, build=False)``) is built by extra workgroups of the same launch, as in
        training: a stream of batches needs one launch per batch."""
        api = self.api
        if topo is None:
            topo = self._topology_of(batch, False)
        if self._can_fuse(topo, batch.x.shape[1], next_topo, False, batch.x):
            c = self._fused_prepare(batch, topo, False, next_topo)
            if (int(getattr(topo, "flags", 0)) & _lib.TOPO_LEAN) and not c["plan"].lean_ok:
                topo.rebuild()       # this inference launch reads the depth-0 member lists a lean build leaves out
                c = self._fused_prepare(batch, topo, False, next_topo)
            api.net_train_step(c["desc"], self._head_desc(False), c["x"], None, self.step2, topo.ws_i32, topo.ws_f32,
                               c["n_nodes"], topo.n_edges, c["B"], topo.max_nodes, topo.max_edges, topo.max_c0,
                               c["pred"], c["readout"], None, None, c["xchg"], c["stream"],
                               next_topology=None if next_topo is None else next_topo.request(
                                   self._flags_for(next_topo, c["x"].shape[1], train=False)),
                               hints=None if c.get("hints") is None else c["hints"][0])
            return c["pred"]
        if int(getattr(topo, "flags", 0)) & _lib.TOPO_LEAN:
            topo.rebuild()       # the forward launch reads the depth-0 member lists a lean build leaves out
        if next_topo is not None:
            next_topo.rebuild()
        stream = _lib.current_stream(batch.x)
        x, desc, xp, arg0, arg1, readout, scratch = self._body_forward(batch, topo, stream)
        pred = torch.empty((topo.n_graphs, self.O), dtype=torch.float32, device=x.device)
        api.head_step(self._head_desc(False), readout, None, topo.n_graphs, self.step, pred, None, None, stream)
        return pred
