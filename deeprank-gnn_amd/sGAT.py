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

__all__ = ["sGAT", "sGraphAttentionLayer"]


class sGraphAttentionLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True, undirected=True):
        super().__init__()
        if not bias or not undirected:
            raise NotImplementedError("only the configuration the reference nets build "
                                      "(bias=True, undirected=True) is on the device path")
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.undirected = undirected
        self.weight = Parameter(torch.Tensor(2 * in_channels, out_channels))
        self.bias = Parameter(torch.Tensor(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(2 * self.in_channels)
        self.weight.data.uniform_(-bound, bound)
        self.bias.data.uniform_(-bound, bound)

    def live_parameters(self):
        return (self.weight, self.bias)

    def forward(self, x, edge_index, edge_attr):
        from .layers import conv_layer_forward
        return conv_layer_forward(_lib.SGAT, x, edge_index, edge_attr, self.live_parameters())

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
        x = self.body(data, topo)
        x = F.relu(self.fc1(x))
        return self.fc2(x)
