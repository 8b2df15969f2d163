"""GINet on the fused MI355X path -- same constructor, attributes, parameter names, shapes
and initialisation as reference deeprank_gnn/ginet.py, so checkpoints load with
``strict=True`` and ``deeprank_gnn.NeuralNet(database, GINet, ...)`` can use it unchanged.

What the reference computes (and this class reproduces):
  * ``GINetConvLayer`` (ginet.py:22-78): ``z = scatter_sum(alpha * fc(x[col]), row)`` where
    ``alpha = softmax(leaky_relu(fc_attention([fc(x_row) | fc(x_col) | fc_edge_attr(a)])), dim=1)``
    is a softmax over ONE element, i.e. identically 1 -- so ``z_i = sum_{e: row=i} W x_col``
    and the attention / edge parameters get zero gradients;
  * ``GINet.forward`` (ginet.py:99-141): two branches (conv1/conv2 and conv1_ext/conv2_ext)
    over the SAME edge_index, each conv -> relu -> community_pooling(cluster0) -> conv ->
    relu -> max_pool_x(cluster1); graph mean of both; fc1 -> relu -> dropout(0.4) -> fc2.
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib
from .functional import net_body, zero_grad_passthrough
from .topology import Topology
from .fused_autograd import engine_for
from .composed import composed_forward, default_layers

__all__ = ["GINet", "GINetConvLayer"]


def _uniform(size, tensor):
    if tensor is not None:
        bound = 1.0 / math.sqrt(size)
        tensor.data.uniform_(-bound, bound)


class GINetConvLayer(nn.Module):
    def __init__(self, in_channels, out_channels, number_edge_features=1, bias=False):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.fc = nn.Linear(in_channels, out_channels, bias=bias)
        self.fc_edge_attr = nn.Linear(number_edge_features, number_edge_features, bias=bias)
        self.fc_attention = nn.Linear(2 * out_channels + number_edge_features, 1, bias=bias)
        self.reset_parameters()

    def reset_parameters(self):
        size = self.in_channels
        _uniform(size, self.fc.weight)
        _uniform(size, self.fc_attention.weight)
        _uniform(size, self.fc_edge_attr.weight)

    def live_parameters(self):
        if self.fc.bias is not None:
            raise NotImplementedError("the fused GINet uses bias=False (what the reference net builds); "
                                      "bias=True runs through forward()")
        return (self.fc.weight,)

    def dead_parameters(self):
        return (self.fc_edge_attr.weight, self.fc_attention.weight)

    def forward(self, x, edge_index, edge_attr=None):
        from .layers import conv_layer_forward
        dead = self.dead_parameters()
        if self.fc.bias is None:
            z = conv_layer_forward(_lib.GINET, x, edge_index, edge_attr, self.live_parameters())
        else:
            # GINetConvLayer(bias=True) (ginet.py:26-37): fc's bias is added PER EDGE before the sum over a row's edges
            # (z_i = sum_e (W x_col + b)); with a constant-one input column it is one more column of the weight, so
            # the same device kernel computes it, and autograd splits the gradient back into fc.weight / fc.bias.
            # The biases of the two dead Linear layers get zero gradients like their weights.
            ones = torch.ones((x.size(0), 1), dtype=x.dtype, device=x.device)
            z = conv_layer_forward(_lib.GINET, torch.cat([x, ones], dim=1), edge_index, edge_attr,
                                   (torch.cat([self.fc.weight, self.fc.bias.view(-1, 1)], dim=1),))
            dead = dead + (self.fc_edge_attr.bias, self.fc_attention.bias)
        return zero_grad_passthrough(z, dead)

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)


class GINet(nn.Module):
    # input_shape -> number of node input features
    # output_shape -> number of output value per graph
    # input_shape_edge -> number of edge input features
    def __init__(self, input_shape, output_shape=1, input_shape_edge=1):
        super().__init__()
        self.conv1 = GINetConvLayer(input_shape, 16, input_shape_edge)
        self.conv2 = GINetConvLayer(16, 32, input_shape_edge)
        self.conv1_ext = GINetConvLayer(input_shape, 16, input_shape_edge)
        self.conv2_ext = GINetConvLayer(16, 32, input_shape_edge)
        self.fc1 = nn.Linear(2 * 32, 128)
        self.fc2 = nn.Linear(128, output_shape)
        self.clustering = 'mcl'
        self.dropout = 0.4

    def body(self, data, topo=None):
        """Per-graph readout [B, 64] = [branch(conv1, conv2) | branch(conv1_ext, conv2_ext)]."""
        if topo is None:
            topo = Topology.from_batch(data, need_weights=False)
        convs = (self.conv1, self.conv2, self.conv1_ext, self.conv2_ext)
        live = tuple(p for c in convs for p in c.live_parameters())
        dead = tuple(p for c in convs for p in c.dead_parameters())
        readout = net_body(_lib.GINET, data.x, topo, live, n_branch=2)
        return zero_grad_passthrough(readout, dead)

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
        x = F.dropout(x, self.dropout, training=self.training)
        return self.fc2(x)
