"""``model(batch)`` / ``loss.backward()`` of GINet / sGAT / FoutNet on the fused step kernels -- the drop-in boundary.

The reference trainer runs, per mini-batch (reference NeuralNet.py:489-506)

    optimizer.zero_grad(); pred = model(batch); loss = loss_fn(pred, y); loss.backward(); optimizer.step()

with the loss and the optimiser OUTSIDE the model.  Everything between ``batch`` and ``pred`` is row-wise per graph
(conv -> pool -> conv -> pool -> graph mean -> fc1 -> relu -> dropout -> fc2, ginet.py:99-141), so

    d loss / d theta  =  sum_g  (d loss / d pred_g) . (d pred_g / d theta)

and the aggregation-first step kernels (csrc/drgnn_step2.h / drgnn_step3.h), which keep a graph's forward AND backward in one
workgroup, serve this boundary in two ways (include/drgnn.h: DRGNN_TASK_GRAD, drgnn_step_gradients):

* one output (every regression net, the reference's default): ``model(batch)`` is ONE launch of the training-step kernel with
  an upstream gradient of one -- it writes ``pred`` and leaves d pred_g / d theta in the per-graph slabs -- and
  ``loss.backward()`` is ONE launch that contracts the slabs with whatever d loss / d pred autograd hands over.  Any loss.
* several outputs (classification): ``model(batch)`` is the forward-only instance (dropout mask of the step), ``backward`` the
  training-step kernel fed with d loss / d pred [B, O] instead of a target, then the slab sum.

The topology workspace of the batch (CSR, consecutive clusters, pooled graph, level-0 aggregation tiles) is a lean + tiles
build (DRGNN_TOPO_LEAN | DRGNN_TOPO_TILES), kept with the batch object while its index tensors are unchanged.  Buffers are
grow-only per batch size; per call only ``pred`` and one flat gradient buffer are allocated (cached-allocator blocks, no
launch); the parameters' ``.grad`` become views of that buffer without a copy.  No CPU path, and no silent other path: ``run`` returns None (the caller then takes the launch pair
of functional.net_body) only for shapes outside the fused kernels -- ``last_path`` says which one ran.
"""
import ctypes
import weakref

import torch

from . import _lib
from .functional import H1, H2, _describe, _fill_grads, _split
from .topology import Topology

__all__ = ["StepEngine", "engine_for", "net_layout"]


def net_layout(net):
    """(kind, n_branch, [conv modules in kernel order]) of one of the three reference nets."""
    name = type(net).__name__
    if name == "GINet":
        return _lib.GINET, 2, [net.conv1, net.conv2, net.conv1_ext, net.conv2_ext]
    if name == "sGAT":
        return _lib.SGAT, 1, [net.conv1, net.conv2]
    if name == "FoutNet":
        return _lib.FOUT, 1, [net.conv1, net.conv2]
    raise TypeError("the fused step drives GINet / sGAT / FoutNet, not %s" % name)


class _Call(object):
    """One ``model(batch)`` whose backward may still come: what the backward launch needs."""
    __slots__ = ("mode", "x", "topo", "plan", "hints", "bufs", "B", "n_feat", "p_drop", "done", "stream", "pred", "__weakref__")


class _StepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, engine, call, *params):
        ctx.engine, ctx.call = engine, call
        ctx.set_materialize_grads(False)
        return engine._forward_launch(call)

    @staticmethod
    def backward(ctx, gpred):
        engine, call = ctx.engine, ctx.call
        n = len(engine.params)
        if gpred is None:
            return (None, None) + (None,) * n
        grads = engine._backward_launch(call, gpred)
        return (None, None) + tuple(g if need else None for g, need in zip(grads, ctx.needs_input_grad[2:]))


class StepEngine(object):
    """Per-model state of the fused autograd path (created on first use, ``engine_for``)."""

    def __init__(self, net, api=None):
        self.api = api or _lib.get()
        self.net = weakref.proxy(net)
        self.kind, self.n_branch, convs = net_layout(net)
        self.convs = convs
        named = list(net.named_parameters())
        self.names = [n for n, _ in named]
        owner, _, self.first_name = self.names[0].rpartition(".")
        self._first_owner = weakref.proxy(net.get_submodule(owner) if owner else net)
        self.params = [p for _, p in named]
        dev = self.params[0].device
        self.device = dev
        self.offset, off = {}, 0
        for n, p in named:
            self.offset[n] = off
            off += p.numel()
        self.total = off
        head = ["fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
        expect = self.offset[head[0]]
        self.head_offset = expect
        lookup = dict(named)
        for n in head:      # the slab sum writes the head's gradient as one contiguous block
            if self.offset[n] != expect:
                raise _lib.DrgnnError("unexpected parameter order for the FC head")
            expect += lookup[n].numel()
        self.live = tuple(p for c in convs for p in c.live_parameters())
        live_ids = {id(p) for p in self.live}
        head_ids = {id(lookup[n]) for n in head}
        # parameters no kernel writes a gradient for (GINetConvLayer's attention: identically zero, ginet.py:63-66)
        self.dead = [(self.offset[n], p.numel()) for n, p in named if id(p) not in live_ids and id(p) not in head_ids]
        if len(self.dead) > _lib.ZERO_RANGES:
            raise _lib.DrgnnError("more untouched parameters than drgnn_step_gradients clears")
        self.R, self.H, self.O = net.fc1.in_features, net.fc1.out_features, net.fc2.out_features
        # Every backward writes ONE fresh flat buffer (a cached-allocator block, no launch) and hands autograd fresh views of it:
        # nobody else holds them, so AccumulateGrad adopts them as the parameters' .grad without a copy (16 copy launches per
        # step otherwise), and a gradient still referenced somewhere -- a .grad kept by the caller, a sum over several forwards
        # in flight inside autograd's input buffers -- is never written again.
        self._shapes = [(self.offset[n], p.numel(), tuple(p.shape)) for n, p in named]
        self._sizes = [n for _, n, _ in self._shapes]
        self._view_shapes = [None if len(shape) == 1 else shape for _, _, shape in self._shapes]
        self._live_slots = None
        self._grad_tpl = {}
        self.step2 = torch.zeros(4, dtype=torch.int32, device=dev)
        self.seed = int(torch.initial_seed()) & 0xFFFFFFFF
        self._desc_key, self._desc, self._heads = None, None, {}
        self._bufs = {}            # (B, n_feat, slabs) -> [readout, partials, head slabs, owner weakref]
        self._xchg = {}
        self._ones = {}
        self._fwd_readout = {}     # per batch size: the readout of forward-only launches (an output nobody reads)
        self._topos = weakref.WeakKeyDictionary()
        self._pending = weakref.WeakSet()
        self._uncommitted = False  # a training launch has written step2[1] and no slab sum has committed it yet
        self.plan_overrides = {}
        self.cache_topology = True
        self.last_path = None      # 'jacobian' / 'two-launch' / 'inference' / None (the caller's launch pair)
        self.last_plan = None
        self.last_reason = None    # why the last call was outside the fused kernels (last_path None)

    # (an engine is launch state, not model state: copies and pickles of the net start without one)
    def __deepcopy__(self, memo):
        return None

    def __reduce__(self):
        return (type(None), ())

    # -- descriptors -----------------------------------------------------------------------------------------------------
    def _descs(self, n_feat):
        key = (n_feat,) + tuple(p.data_ptr() for p in self.live) + (self.net.fc1.weight.data_ptr(), self.net.fc1.bias.data_ptr(),
                                                                    self.net.fc2.weight.data_ptr(), self.net.fc2.bias.data_ptr())
        if key != self._desc_key:
            for p in self.params:
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise TypeError("the hot path computes in fp32 on contiguous parameters, like the reference")
            if self.api is _lib._API:
                _lib.require_device(*self.params)
            self._desc = _describe(self.kind, n_feat, tuple(p.detach() for p in self.live), self.n_branch)
            self._desc_key = key
            self._heads = {}          # (they hold the head's parameter addresses)
        return self._desc

    def _grads_for(self, flat, n_feat):
        """(g_conv1, g_conv2, zero ranges, per-parameter views, head block address) of the flat gradient buffer ``flat``.  The
        descriptor structures are built once per feature count; per call only their pointers move with the buffer."""
        # (one split call + a view per matrix: fresh tensor objects every time, nobody else holds them)
        views = tuple(part if shape is None else part.view(shape)
                      for part, shape in zip(flat.split(self._sizes), self._view_shapes))
        base = flat.data_ptr()
        tpl = self._grad_tpl.get(n_feat)
        if tpl is None:
            if self._live_slots is None:
                index = {id(p): i for i, p in enumerate(self.params)}
                self._live_slots = tuple(index[id(p)] for p in self.live)
            live_grads = tuple(views[i] for i in self._live_slots)
            g1 = (_lib.ConvGrads * _lib.MAX_BRANCH)()
            g2 = (_lib.ConvGrads * _lib.MAX_BRANCH)()
            for b, (l1, l2) in enumerate(_split(self.kind, live_grads, self.n_branch)):
                _fill_grads(g1[b], self.kind, l1, n_feat, H1)
                _fill_grads(g2[b], self.kind, l2, H1, H2)
            patch = [(g[b], f, getattr(g[b], f) - base) for g in (g1, g2) for b in range(self.n_branch)
                     for f in ("w_nbr", "w_self", "bias") if getattr(g[b], f)]
            zp = (ctypes.c_void_p * _lib.ZERO_RANGES)()
            zl = (ctypes.c_int64 * _lib.ZERO_RANGES)()
            for i, (off, n) in enumerate(self.dead):
                zl[i] = n
            tpl = self._grad_tpl[n_feat] = (g1, g2, zp, zl, patch)
        g1, g2, zp, zl, patch = tpl
        for struct, field, off in patch:
            setattr(struct, field, base + off)
        for i, (off, n) in enumerate(self.dead):
            zp[i] = base + 4 * off
        return g1, g2, zp, zl, views, base + 4 * self.head_offset

    def _head_desc(self, train, task, p_drop):
        mask = getattr(self, "drop_mask", None) if train else None        # test hook, as FusedTrainer's
        key = (int(train), task, float(p_drop), None if mask is None else mask.data_ptr())
        hd = self._heads.get(key)
        if hd is None:
            hd = self._heads[key] = _lib.HeadDesc()
            n = self.net
            hd.R, hd.H, hd.O, hd.task, hd.train = self.R, self.H, self.O, task, int(train)
            hd.p_drop = float(p_drop)
            hd.seed = self.seed
            hd.transform_sigmoid = 0        # (NeuralNet.format_output transforms outside the model, NeuralNet.py:616-631)
            hd.w1, hd.b1 = n.fc1.weight.data_ptr(), n.fc1.bias.data_ptr()
            hd.w2, hd.b2 = n.fc2.weight.data_ptr(), n.fc2.bias.data_ptr()
            hd.class_w = None
            hd.drop_mask = None if mask is None else mask.data_ptr()
        return hd

    # -- topology of a batch ---------------------------------------------------------------------------------------------
    @staticmethod
    def _stamp(data, need_w):
        vals = []
        for key in ("x", "edge_index", "batch", "cluster0", "cluster1") + (("edge_attr",) if need_w else ()):
            t = getattr(data, key, None)
            vals.append(None if t is None else (t.data_ptr(), t._version, tuple(t.shape)))
        return tuple(vals)

    def topology_for(self, data):
        """The batch's workspace: lean + tiles build, kept with the batch object while its tensors are unchanged."""
        need_w = self.kind == _lib.SGAT
        stamp = self._stamp(data, need_w)
        held = None
        if self.cache_topology:
            try:
                held = self._topos.get(data)
            except TypeError:       # (not weak-referenceable / not hashable: no caching for this batch type)
                held = None
        if held is not None and held[0] == stamp:
            return held[1]
        topo = Topology.from_batch(data, api=self.api, need_weights=need_w, build=False)
        flags = topo.full_flags()
        if topo.tiles is not None:      # (also tiles the builder cannot stage: asked for only when a fused kernel will read them)
            x = data.x
            af = _lib.TOPO_HIER | _lib.TOPO_LEAN | _lib.TOPO_TILES
            p = self._plan(int(x.shape[1]), topo, True, af)
            if p.lean_ok and p.family == _lib.STEP_FAMILY_AGGREGATE:
                flags = af
        topo.rebuild(flags)
        if self.cache_topology:
            try:
                self._topos[data] = (stamp, topo, {})
            except TypeError:
                pass
        return topo

    def _usable_flags(self, topo, x):
        flags = int(getattr(topo, "flags", 0))
        tiles = getattr(topo, "tiles", None)
        ok = tiles is not None and (flags & _lib.TOPO_TILES) and ((topo.ws_f32 is not None) == (self.kind == _lib.SGAT))
        if ok:
            tx = getattr(topo, "x", None)
            ok = (tx is not None and tx.data_ptr() == x.data_ptr() and tuple(tx.shape) == tuple(x.shape) and
                  (x.shape[1] % 4 != 0 or x.data_ptr() % 16 == 0) and getattr(topo, "_tiles_x_version", None) == x._version)
        if not ok:
            flags &= ~_lib.TOPO_TILES
        return flags

    def _plan(self, n_feat, topo, train, topo_flags):
        return self.api.step_plan(self.kind, n_feat, topo.max_nodes, topo.max_edges, topo.max_c0, self.R, self.H, self.O,
                                  topo.n_graphs, 0, train, topo_flags, self.plan_overrides)

    # -- buffers ---------------------------------------------------------------------------------------------------------
    def _buffers(self, plan, n_feat, B, call):
        slabs = max(int(plan.slabs_per_graph), self.n_branch)
        key = (B, n_feat, slabs)
        held = self._bufs.get(key)
        if held is not None:
            owner = held[3]() if held[3] is not None else None
            if owner is None or owner.done:
                held[3] = weakref.ref(call)
                return held[:3], slabs
        dev = self.device
        fresh = [torch.empty((max(B, 1), H2 * self.n_branch), dtype=torch.float32, device=dev),
                 torch.empty((max(B * slabs, 1), self.api.net_partial_elems(self.kind, n_feat)), dtype=torch.float32, device=dev),
                 torch.empty((max(B, 1), self.api.head_compact_elems(self.R, self.H, self.O)), dtype=torch.float32, device=dev),
                 weakref.ref(call)]
        if held is None:
            self._bufs[key] = fresh
        # (else: the cached set belongs to a forward whose backward is still to come -- this call keeps a private one)
        return fresh[:3], slabs

    def _xchg_for(self, plan, B):
        words = int(plan.xchg_words)
        if words <= 0 and self.n_branch == 1:
            return None
        words = max(words, self.n_branch * max(self.H, 32))
        buf = self._xchg.get(B)
        if buf is None or buf.shape[1] < words:
            buf = self._xchg[B] = torch.zeros((max(B, 1), words), dtype=torch.int64, device=self.device)
        return buf

    # -- one call --------------------------------------------------------------------------------------------------------
    def _outside(self, why):
        """This call is the caller's launch pair's; ``last_reason`` says why."""
        self.last_reason = why
        return None

    def run(self, data, topo=None):
        """pred [B, O] through the fused kernels, or None when this call is outside them (see the module docstring)."""
        net = self.net
        x = data.x
        self.last_path = self.last_reason = None
        if not (torch.is_tensor(x) and x.dim() == 2 and x.dtype == torch.float32 and x.is_contiguous()) or x.requires_grad:
            return self._outside("node features are not a contiguous float32 [n, F] tensor without gradient")
        if topo is not None and topo.api is not self.api:
            return self._outside("workspace of another library build")          # (a workspace of another library build -- the CPU suite's emulation: its launch pair)
        if self.api is _lib._API and not x.is_cuda:
            _lib.require_device(x)
        if self.params[0].device != x.device or self.params[0].device != self.device:
            return self._outside("parameters and node features on different devices")
        held = None
        if topo is None:
            topo = self.topology_for(data)
            if self.cache_topology:
                try:
                    held = self._topos.get(data)        # (stamp, topo, launch contexts of this batch by mode)
                except TypeError:
                    held = None
        B, n_feat = topo.n_graphs, int(x.shape[1])
        if B <= 0 or topo.max_nodes <= 0 or x.shape[0] != topo.n_nodes:
            return self._outside("empty batch or a workspace of another batch")
        want_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.params)
        p_drop = float(getattr(net, "dropout", 0.0)) if net.training else 0.0
        mode = "inference"
        if want_grad:
            mode = "jacobian" if self.O == 1 else "two-launch"
        ctx_key = (mode, tuple(sorted(self.plan_overrides.items())))
        ctx = held[2].get(ctx_key) if held is not None and held[1] is topo else None
        if ctx is not None:
            # the same batch object, tensors unchanged (topology_for compared the stamp): plan and launch hints as last time
            flags, plan, hints, bplan, bhints = ctx
            if mode == "two-launch" and p_drop > 0.0 and any(not c.done for c in self._pending):
                return self._outside("dropout with an earlier forward still awaiting its backward")
            return self._issue(mode, x, topo, plan, hints, bplan, bhints, B, n_feat, p_drop, want_grad)
        flags = self._usable_flags(topo, x)
        if not (flags & _lib.TOPO_TILES) and getattr(topo, "tiles", None) is not None and \
                ((topo.ws_f32 is not None) == (self.kind == _lib.SGAT)) and tuple(topo.x.shape) == tuple(x.shape) and \
                (n_feat % 4 != 0 or x.data_ptr() % 16 == 0) and getattr(topo, "_inputs", None) is not None:
            # tiles formed from other node features than the ones stepped (x replaced / modified in place): form them again
            topo.x = x
            topo.rebuild(int(topo.flags) | _lib.TOPO_TILES)
            flags = self._usable_flags(topo, x)
        plan = self._plan(n_feat, topo, mode == "jacobian", flags)
        if plan.family != _lib.STEP_FAMILY_AGGREGATE or not (0 < plan.lds_bytes <= 160 * 1024):
            why = ""
            if getattr(topo, "tiles", None) is not None and (topo.ws_f32 is not None) != (self.kind == _lib.SGAT):
                why = "; the workspace's tiles are %s sums, this net starts from %s ones (Topology.from_batch(need_weights=...))" % (
                    ("edge-weighted", "plain") if topo.ws_f32 is not None else ("plain", "edge-weighted"))
            return self._outside("no fused kernel for this launch (family %d, topology flags 0x%x, %d features, %d / %d / %d "
                                 "nodes / edges / clusters per graph at most%s)" % (plan.family, flags, n_feat, topo.max_nodes,
                                                                                   topo.max_edges, topo.max_c0, why))
        if mode == "two-launch":
            chk = self._plan(n_feat, topo, True, flags)
            if chk.family != _lib.STEP_FAMILY_AGGREGATE or not (0 < chk.lds_bytes <= 160 * 1024):
                return self._outside("no fused training kernel for this launch (family %d)" % chk.family)
            if p_drop > 0.0 and any(not c.done for c in self._pending):
                return self._outside("dropout with an earlier forward still awaiting its backward")      # (an earlier forward's backward would move the dropout stream between this forward and its own)
        bplan = bhints = None
        bd = getattr(data, "__dict__", {})
        hn, he = bd.get("_host_node_ptr"), bd.get("_host_edge_ptr")
        if hn is None:           # (a foreign batch object: the tables Topology.from_batch derived)
            hn, he = getattr(topo, "host_node_ptr", None), getattr(topo, "host_edge_ptr", None)
        tiles = topo.tiles if (flags & _lib.TOPO_TILES) else None

        def hints_for(pl):
            if hn is not None and he is not None and len(hn) == B + 1 and B <= 64:
                return _lib.step_hints(node_ptr=hn, edge_ptr=he, topo_flags=flags, tiles=tiles, plan=pl)
            return _lib.step_hints(topo_flags=flags, tiles=tiles, plan=pl)
        hints = hints_for(plan)
        if mode == "two-launch":
            bplan = self._plan(n_feat, topo, True, flags)
            bhints = hints_for(bplan)
        if held is not None and held[1] is topo:
            held[2][ctx_key] = (flags, plan, hints, bplan, bhints)
        return self._issue(mode, x, topo, plan, hints, bplan, bhints, B, n_feat, p_drop, want_grad)

    def _issue(self, mode, x, topo, plan, hints, bplan, bhints, B, n_feat, p_drop, want_grad):
        call = _Call()
        call.mode, call.x, call.topo, call.plan, call.B, call.n_feat, call.p_drop = mode, x, topo, plan, B, n_feat, p_drop
        call.done = not want_grad
        call.stream = _lib.current_stream(x)
        call.hints = hints
        if mode == "two-launch":
            call.bufs = (bplan, bhints)
        self.last_path, self.last_plan = mode, plan
        if want_grad:
            self._pending.add(call)
            return _StepFn.apply(self, call, *self.params)
        return self._forward_launch(call)

    def _forward_launch(self, call):
        api, topo, x, B = self.api, call.topo, call.x, call.B
        desc = self._descs(call.n_feat)
        pred = torch.empty((B, self.O), dtype=torch.float32, device=x.device)
        xchg = self._xchg_for(call.plan, B)
        if call.mode == "jacobian":
            if self._uncommitted:
                # a training launch whose backward has not run (yet): its step index was never committed, and a second launch
                # with the same index would find the first one's exchange words carrying its own tag
                self.step2[0:1].copy_(self.step2[1:2])
            self._uncommitted = True
            (readout, partials, hp), slabs = self._buffers(call.plan, call.n_feat, B, call)
            ones = self._ones.get(B)
            if ones is None:
                ones = self._ones[B] = torch.ones((B, 1), dtype=torch.float32, device=x.device)
            call.bufs = (readout, partials, hp, slabs)
            head = self._head_desc(1, _lib.TASK_GRAD, call.p_drop)
            api.net_train_step(desc, head, x, ones, self.step2, topo.ws_i32, topo.ws_f32, topo.n_nodes, topo.n_edges, B,
                               topo.max_nodes, topo.max_edges, topo.max_c0, pred, readout, hp, partials, xchg, call.stream,
                               hints=call.hints[0])
        else:
            # forward only: the inference instance; in training mode with the dropout mask of the step in flight (train = 2)
            readout = self._fwd_readout.get(B)
            if readout is None:
                readout = self._fwd_readout[B] = torch.empty((B, H2 * self.n_branch), dtype=torch.float32, device=x.device)
            head = self._head_desc(2 if call.p_drop > 0.0 else 0, _lib.TASK_REG, call.p_drop)
            api.net_train_step(desc, head, x, None, self.step2, topo.ws_i32, topo.ws_f32, topo.n_nodes, topo.n_edges, B,
                               topo.max_nodes, topo.max_edges, topo.max_c0, pred, readout, None, None, xchg, call.stream,
                               hints=call.hints[0])
        return pred

    def _backward_launch(self, call, gpred):
        api, B = self.api, call.B
        if call.done:
            raise RuntimeError("this forward's buffers were released (backward through it a second time)")
        gpred = gpred.to(torch.float32).contiguous()
        stream = _lib.current_stream(call.x)
        desc = self._descs(call.n_feat)
        flat = torch.empty(self.total, dtype=torch.float32, device=self.device)
        g1, g2, zp, zl, views, head_grad = self._grads_for(flat, call.n_feat)
        if call.mode == "jacobian":
            readout, partials, hp, slabs = call.bufs
            weight = gpred.view(-1)
        else:
            bplan, bhints = call.bufs
            topo, x = call.topo, call.x
            (readout, partials, hp), slabs = self._buffers(bplan, call.n_feat, B, call)
            pred = torch.empty((B, self.O), dtype=torch.float32, device=x.device)
            head = self._head_desc(1, _lib.TASK_GRAD, call.p_drop)
            api.net_train_step(desc, head, x, gpred, self.step2, topo.ws_i32, topo.ws_f32, topo.n_nodes, topo.n_edges, B,
                               topo.max_nodes, topo.max_edges, topo.max_c0, pred, readout, hp, partials,
                               self._xchg_for(bplan, B), stream, hints=bhints[0])
            weight = None
        api.step_gradients(desc, partials, B, g1, g2, hp, readout, self.R, self.H, self.O, head_grad, weight, zp, zl,
                           len(self.dead), self.step2, slabs, stream)
        self._uncommitted = False
        call.done = True
        return views


def engine_for(net):
    """The net's engine (created on first use; rebuilt when parameters were replaced or moved to another device).  The check
    is two identity tests, not a walk over the module tree: this runs in front of every ``model(batch)``."""
    eng = net.__dict__.get("_drgnn_engine")
    if eng is not None:
        first, last = eng.params[0], eng.params[-1]
        if net.fc2.bias is last and eng._first_owner._parameters.get(eng.first_name) is first and first.device == eng.device:
            return eng
    eng = net.__dict__["_drgnn_engine"] = StepEngine(net)
    return eng
