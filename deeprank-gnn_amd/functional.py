"""Autograd glue between torch modules and the fused HIP network body.

``net_body(kind, x, topo, params)`` runs, for every (graph, branch) of the mini-batch,
    conv1 -> relu -> community max-pool -> conv2 -> relu -> max_pool_x -> graph mean
i.e. reference ginet.py:103-114,133 / sGAT.py:119-133 / foutnet.py:108-120, and returns
the per-graph readout ``[B, 32 * n_branch]``; the FC head stays in torch.
Forward and backward are one kernel launch each (plus a small deterministic reduction of
the per-graph weight-gradient partials).  No CPU path: tensors must live on the GPU.
"""
import ctypes

import torch

from . import _lib
from ._lib import GINET, SGAT, FOUT, ConvGrads, ConvParams, NetDesc

__all__ = ["net_body", "zero_grad_passthrough", "KIND_PARAMS"]

H1, H2 = 16, 32
_F32 = 4

# tensors per branch, in the order the autograd function receives them
KIND_PARAMS = {
    GINET: 2,   # conv1.fc.weight [16,F], conv2.fc.weight [32,16]
    SGAT: 4,    # conv1.weight [2F,16], conv1.bias, conv2.weight [32,32], conv2.bias
    FOUT: 6,    # conv1.Wc, conv1.Wn [F,16], conv1.bias, conv2.Wc, conv2.Wn [16,32], conv2.bias
}


def _api():
    return _lib.get()


def _fill_conv(cp, kind, tensors, k_in, h):
    """Describe one layer's weights as strided [K,H] operands living inside the model's
    own parameter tensors (include/drgnn.h: drgnn_conv_params)."""
    if kind == GINET:
        (w,) = tensors                      # nn.Linear weight [H, K]: (k,h) -> w[h*K + k]
        cp.w_nbr, cp.nbr_sk, cp.nbr_sh = w.data_ptr(), 1, k_in
        cp.w_self, cp.self_sk, cp.self_sh = None, 0, 0
        cp.bias = None
    elif kind == SGAT:
        w, b = tensors                      # [2K, H]: rows 0..K-1 act on x_i, K..2K-1 on x_j
        cp.w_self, cp.self_sk, cp.self_sh = w.data_ptr(), h, 1
        cp.w_nbr, cp.nbr_sk, cp.nbr_sh = w.data_ptr() + k_in * h * _F32, h, 1
        cp.bias = b.data_ptr()
    else:
        wc, wn, b = tensors                 # [K, H] each
        cp.w_self, cp.self_sk, cp.self_sh = wc.data_ptr(), h, 1
        cp.w_nbr, cp.nbr_sk, cp.nbr_sh = wn.data_ptr(), h, 1
        cp.bias = b.data_ptr()


def _fill_grads(cg, kind, tensors, k_in, h):
    if kind == GINET:
        (w,) = tensors
        cg.w_nbr, cg.w_self, cg.bias = w.data_ptr(), None, None
    elif kind == SGAT:
        w, b = tensors
        cg.w_self, cg.w_nbr, cg.bias = w.data_ptr(), w.data_ptr() + k_in * h * _F32, b.data_ptr()
    else:
        wc, wn, b = tensors
        cg.w_self, cg.w_nbr, cg.bias = wc.data_ptr(), wn.data_ptr(), b.data_ptr()


def _split(kind, params, n_branch):
    per = KIND_PARAMS[kind]
    half = per // 2
    out = []
    for b in range(n_branch):
        chunk = params[b * per:(b + 1) * per]
        out.append((chunk[:half], chunk[half:]))
    return out


def _describe(kind, n_feat, params, n_branch):
    desc = NetDesc()
    desc.kind, desc.n_branch, desc.n_feat = kind, n_branch, n_feat
    for b, (l1, l2) in enumerate(_split(kind, params, n_branch)):
        _fill_conv(desc.conv1[b], kind, l1, n_feat, H1)
        _fill_conv(desc.conv2[b], kind, l2, H1, H2)
    return desc


class _NetBody(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, topo, kind, n_branch, *params):
        api = topo.api
        if api is _lib._API:
            _lib.require_device(x, *params)
        x = x.contiguous()
        params = tuple(p.detach().contiguous() for p in params)
        if x.dtype != torch.float32 or any(p.dtype != torch.float32 for p in params):
            raise TypeError("the hot path computes in fp32, like the reference")
        n_nodes, n_feat = x.shape
        if n_nodes != topo.n_nodes:
            raise ValueError("x has %d rows, the batch has %d nodes" % (n_nodes, topo.n_nodes))
        if kind == SGAT and topo.ws_f32 is None:
            raise ValueError("sGAT needs edge_attr (sGAT.py:76)")
        dev = x.device
        B = topo.n_graphs
        xp = torch.empty((n_branch, n_nodes, H1), dtype=torch.float32, device=dev)
        arg0 = torch.empty((n_branch, n_nodes, H1), dtype=torch.int32, device=dev)
        arg1 = torch.empty((n_branch, n_nodes, H2), dtype=torch.int32, device=dev)
        readout = torch.empty((B, H2 * n_branch), dtype=torch.float32, device=dev)
        scratch = None
        lds = api.net_lds_bytes(kind, n_feat, topo.max_nodes, topo.max_edges, topo.max_c0, False)
        if lds == 0 or lds > 160 * 1024:
            scratch = torch.empty(api.net_scratch_elems(kind, n_feat, n_nodes, topo.n_edges, B),
                                  dtype=torch.float32, device=dev)
        desc = _describe(kind, n_feat, params, n_branch)
        api.net_forward(desc, x, topo.ws_i32, topo.ws_f32, n_nodes, topo.n_edges, B, topo.max_nodes,
                        topo.max_edges, topo.max_c0, xp, arg0, arg1, readout, scratch, _lib.current_stream(x))
        ctx.topo, ctx.kind, ctx.n_branch = topo, kind, n_branch
        ctx.save_for_backward(x, xp, arg0, arg1, *params)
        ctx.mark_non_differentiable(arg0, arg1)
        return readout

    @staticmethod
    def backward(ctx, grad_readout):
        topo, kind, n_branch = ctx.topo, ctx.kind, ctx.n_branch
        api = topo.api
        x, xp, arg0, arg1 = ctx.saved_tensors[:4]
        params = ctx.saved_tensors[4:]
        n_nodes, n_feat = x.shape
        dev = x.device
        B = topo.n_graphs
        grad_readout = grad_readout.contiguous()
        grads = tuple(torch.empty_like(p) for p in params)
        n_part = api.net_partial_elems(kind, n_feat)
        partials = torch.empty((max(B * n_branch, 1), n_part), dtype=torch.float32, device=dev)
        grad_x = None
        if ctx.needs_input_grad[0]:
            grad_x = torch.empty((n_branch, n_nodes, n_feat), dtype=torch.float32, device=dev)
        scratch = None
        lds = api.net_lds_bytes(kind, n_feat, topo.max_nodes, topo.max_edges, topo.max_c0, True)
        if lds == 0 or lds > 160 * 1024:
            scratch = torch.empty(api.net_scratch_elems(kind, n_feat, n_nodes, topo.n_edges, B),
                                  dtype=torch.float32, device=dev)
        desc = _describe(kind, n_feat, params, n_branch)
        g1 = (ConvGrads * _lib.MAX_BRANCH)()
        g2 = (ConvGrads * _lib.MAX_BRANCH)()
        for b, (l1, l2) in enumerate(_split(kind, grads, n_branch)):
            _fill_grads(g1[b], kind, l1, n_feat, H1)
            _fill_grads(g2[b], kind, l2, H1, H2)
        stream = _lib.current_stream(x)
        api.net_backward(desc, x, grad_readout, topo.ws_i32, topo.ws_f32, n_nodes, topo.n_edges, B,
                         topo.max_nodes, topo.max_edges, topo.max_c0, xp, arg0, arg1, grad_x, partials,
                         scratch, stream)
        api.net_reduce_grads(desc, partials, n_nodes, B, g1, g2, grad_x, stream)
        if B == 0:
            grads = tuple(torch.zeros_like(p) for p in params)
        gx = None if grad_x is None else grad_x[0]
        return (gx, None, None, None) + grads


def net_body(kind, x, topo, params, n_branch=1):
    return _NetBody.apply(x, topo, kind, n_branch, *params)


class _ZeroGradPassthrough(torch.autograd.Function):
    """Identity on ``y`` that hands exactly-zero gradients to ``dead`` parameters.

    GINetConvLayer's attention is a softmax over a size-1 axis (ginet.py:63-66), so its
    ``fc_attention`` / ``fc_edge_attr`` weights receive zero (not ``None``) gradients in
    the reference; optimisers see the same here without running the dead arithmetic."""

    @staticmethod
    def forward(ctx, y, *dead):
        ctx.shapes = [(d.shape, d.dtype, d.device) for d in dead]
        return y.view_as(y)

    @staticmethod
    def backward(ctx, gy):
        return (gy,) + tuple(torch.zeros(s, dtype=dt, device=dv) for s, dt, dv in ctx.shapes)


def zero_grad_passthrough(y, dead):
    return _ZeroGradPassthrough.apply(y, *dead)
