"""Stand-alone convolution layers (the modules of ginet.py / sGAT.py / foutnet.py called
directly, as in the reference README's "custom GNN" example): arbitrary output width, whole
input treated as one block-diagonal graph, forward and backward on the device through
``drgnn_conv_layer_forward / _backward``.  The three shipped nets do NOT go through here:
they use the fused per-graph kernels (functional.net_body)."""
import torch

from . import _lib
from ._lib import ConvGrads, ConvParams, GINET, SGAT, FOUT
from .functional import _fill_conv, _fill_grads
from .topology import Topology

__all__ = ["conv_layer_forward"]


def _width(kind, params):
    if kind == GINET:
        return params[0].shape[0]
    return params[0].shape[1]


class _ConvLayer(torch.autograd.Function):
    @staticmethod
    def forward(ctx, kind, api, x, edge_index, edge_attr, *params):
        if api is _lib._API:
            _lib.require_device(x, edge_index, *params)
        x = x.contiguous()
        params = tuple(p.detach().contiguous() for p in params)
        n_nodes, n_feat = x.shape
        H = _width(kind, params)
        ea = None
        if kind == SGAT:
            if edge_attr is None:
                raise ValueError("sGraphAttentionLayer needs edge_attr")
            if edge_attr.dim() == 2 and edge_attr.size(1) != 1:
                raise ValueError("only one edge feature is supported (edge_attr broadcasts over channels)")
            ea = edge_attr
        topo = Topology.single_graph(edge_index, ea, n_nodes, api=api)
        HC = H if kind == GINET else 2 * H
        u = torch.empty((n_nodes, HC), dtype=torch.float32, device=x.device)
        out = torch.empty((n_nodes, H), dtype=torch.float32, device=x.device)
        cp = ConvParams()
        _fill_conv(cp, kind, params, n_feat, H)
        api.conv_layer_forward(kind, x, n_feat, H, cp, topo.ws_i32, topo.ws_f32, topo.n_edges, u, out,
                               _lib.current_stream(x))
        ctx.kind, ctx.api, ctx.topo, ctx.H = kind, api, topo, H
        ctx.save_for_backward(x, *params)
        _ConvLayer.last_topology = topo
        return out

    @staticmethod
    def backward(ctx, grad_out):
        kind, api, topo, H = ctx.kind, ctx.api, ctx.topo, ctx.H
        x = ctx.saved_tensors[0]
        params = ctx.saved_tensors[1:]
        n_nodes, n_feat = x.shape
        grad_out = grad_out.contiguous()
        HC = H if kind == GINET else 2 * H
        du = torch.empty((n_nodes, HC), dtype=torch.float32, device=x.device)
        slabs = max(api.conv_layer_slabs(n_nodes), 1)
        partials = torch.empty((slabs, api.conv_layer_partial_elems(kind, n_feat, H)), dtype=torch.float32,
                               device=x.device)
        grads = tuple(torch.empty_like(p) for p in params)
        if n_nodes == 0:
            grads = tuple(torch.zeros_like(p) for p in params)
        cp, cg = ConvParams(), ConvGrads()
        _fill_conv(cp, kind, params, n_feat, H)
        _fill_grads(cg, kind, grads, n_feat, H)
        gx = torch.empty_like(x) if ctx.needs_input_grad[2] else None
        api.conv_layer_backward(kind, x, n_feat, H, cp, topo.ws_i32, topo.ws_f32, topo.n_edges, grad_out, du,
                                partials, cg, gx, _lib.current_stream(x))
        return (None, None, gx, None, None) + grads


def conv_layer_forward(kind, x, edge_index, edge_attr, params, api=None, return_topology=False):
    out = _ConvLayer.apply(kind, api or _lib.get(), x, edge_index, edge_attr, *params)
    if return_topology:
        return out, _ConvLayer.last_topology      # the single-graph workspace the call built (CSR / CSC of the input)
    return out
