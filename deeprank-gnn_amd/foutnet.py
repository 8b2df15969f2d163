"""FoutNet on the fused MI355X path -- API and parameters of reference deeprank_gnn/foutnet.py.

``FoutLayer`` (foutnet.py:15-87, eq. 1 of Fout et al. NIPS 2017):
    z_i = x_i Wc + 1/N_i * sum_j x_j Wn + b      (mean over the out-edges of i; the reference's
    per-node Python loop takes the mean of an empty slice for an isolated node -> NaN row;
    reproduced, and -- as in the reference -- dropped by the following max-pool).
``FoutNet.forward`` (foutnet.py:103-125): same skeleton as sGAT, edge_attr unused.
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

__all__ = ["FoutNet", "FoutLayer"]


class FoutLayer(nn.Module):
    def __init__(self, in_channels, out_channels, bias=True):
        super().__init__()
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.Wc = Parameter(torch.Tensor(in_channels, out_channels))
        self.Wn = Parameter(torch.Tensor(in_channels, out_channels))
        if bias:
            self.bias = Parameter(torch.Tensor(out_channels))
        else:
            self.register_parameter('bias', None)               # foutnet.py:43-46
        self.reset_parameters()

    def reset_parameters(self):
        bound = 1.0 / math.sqrt(self.in_channels)
        self.Wc.data.uniform_(-bound, bound)
        self.Wn.data.uniform_(-bound, bound)
        if self.bias is not None:
            self.bias.data.uniform_(-bound, bound)

    def live_parameters(self):
        if self.bias is None:
            raise NotImplementedError("the fused FoutNet uses bias=True (what the reference net builds); "
                                      "bias=False runs through forward()")
        return (self.Wc, self.Wn, self.bias)

    def forward(self, x, edge_index):
        from .layers import conv_layer_forward
        bias = self.bias if self.bias is not None else torch.zeros(self.out_channels, dtype=x.dtype, device=x.device)
        return conv_layer_forward(_lib.FOUT, x, edge_index, None, (self.Wc, self.Wn, bias))

    def __repr__(self):
        return '{}({}, {})'.format(self.__class__.__name__, self.in_channels, self.out_channels)


class FoutNet(nn.Module):
    def __init__(self, input_shape, output_shape=1, input_shape_edge=None):
        super().__init__()
        self.conv1 = FoutLayer(input_shape, 16)
        self.conv2 = FoutLayer(16, 32)
        self.fc1 = nn.Linear(32, 64)
        self.fc2 = nn.Linear(64, output_shape)
        self.clustering = 'mcl'

    def body(self, data, topo=None):
        if topo is None:
            topo = Topology.from_batch(data, need_weights=False)
        live = self.conv1.live_parameters() + self.conv2.live_parameters()
        return net_body(_lib.FOUT, data.x, topo, live, n_branch=1)

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
