"""sGAT on the fused MI355X path -- API and parameters of reference deeprank_gnn/sGAT.py.

``sGraphAttentionLayer`` (sGAT.py:19-98):
    z_i = 1/N_i * sum_j a_ij * [x_i || x_j] W + b     (a_ij = edge attribute, N_i = degree,
    scatter_mean with count clamped to 1 -> an isolated node yields b).
``sGAT.forward`` (sGAT.py:114-138): conv1 -> relu -> community_pooling -> conv2 -> relu ->
max_pool_x -> graph mean -> fc1 -> relu -> fc2.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn import Parameter

from . import _lib
from .functional import net_body
from .topology import Topology
from .fused_autograd import engine_for
from .composed import composed_forward, default_layers

__all__ = ["sGAT", "sGraphAttentionLayer"]


class sGraphAttentionLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, undirected=True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.undirected = undirected
        self.weight = Parameter(torch.Tensor(2 * in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)               # sGAT.py:50-53
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(2 * self.in_channels)
        self.weight.data.uniform_(-bound, bound)
        if self.bias is not None:
            self.bias.data.uniform_(-bound, bound)

    def live_parameters(self):
        if self.bias is None or not self.undirected:
            raise NotImplementedError("the fused nets use the configuration the reference nets build "
                                      "(bias=True, undirected=True); other options run through forward()")
        return (self.weight, self.bias)

    def forward(self, x, edge_index, edge_attr):
        """Device kernels (drgnn_conv_layer_forward / _backward) for every constructor option:
        * bias=False: the kernel adds a constant zero row;
        * undirected=False (sGAT.py:86-87): ``scatter_mean(alpha, col, out=out)`` adds the column sums into the row means
          and divides the WHOLE buffer by the clamped column counts, i.e.
              out_i = rowmean_i / max(indeg_i, 1) + colmean_i
          and colmean is this same layer on the reversed edges with the two halves of W swapped."""
        from .layers import conv_layer_forward
        zero = torch.zeros(self.out_channels, dtype=x.dtype, device=x.device)
        if self.undirected:
            return conv_layer_forward(_lib.SGAT, x, edge_index, edge_attr, (self.weight, zero if self.bias is None else self.bias))
        F_in = self.in_channels
        rowmean = conv_layer_forward(_lib.SGAT, x, edge_index, edge_attr, (self.weight, zero))
        swapped = torch.cat([self.weight[F_in:], self.weight[:F_in]], dim=0)
        rev = torch.stack([edge_index[1], edge_index[0]])
        colmean, topo_rev = conv_layer_forward(_lib.SGAT, x, rev, edge_attr, (swapped, zero), return_topology=True)
        rp = topo_rev.array("ROWPTR0")[:x.size(0) + 1]           # CSR of the reversed graph: row i = edges with col == i
        indeg = (rp[1:] - rp[:-1]).clamp(min=1).to(x.dtype).view(-1, 1)
        out = rowmean / indeg + colmean
        return out if self.bias is None else out + self.bias

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)


class sGAT(nn.Module):
    def __init__(self, input_shape, output_shape=1, input_shape_edge=None):
        super().__init__()
        self.conv1 = sGraphAttentionLayer(input_shape, 16)
        self.conv2 = sGraphAttentionLayer(16, 32)
        self.fc1 = nn.Linear(32, 64)
        self.fc2 = nn.Linear(64, output_shape)
        self.clustering = 'mcl'

    def body(self, data, topo=None):
        if topo is None:
            topo = Topology.from_batch(data)
        live = self.conv1.live_parameters() + self.conv2.live_parameters()
        return net_body(_lib.SGAT, data.x, topo, live, n_branch=1)

    def forward(self, data, topo=None):
        """pred [B, output_shape].  On the fused step kernels whenever the batch fits them (fused_autograd: one launch for
        ``model(batch)``, one for ``loss.backward()``); otherwise the launch pair of ``body`` + the head in torch."""
        if not default_layers(self):
            # a layer built with an option the reference nets do not use: the general path, stage by stage (composed.py)
            return composed_forward(self, data)
        pred = engine_for(self).run(data, topo)
        if pred is not None:
            return pred
        x = self.body(data, topo)
        x = F.relu(self.fc1(x))
        return self.fc2(x)
