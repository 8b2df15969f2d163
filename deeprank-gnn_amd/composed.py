"""The nets' forward composed from the stand-alone layers and the pooling functions, the way the reference writes it
(reference ginet.py:99-141, sGAT.py:114-138, foutnet.py:103-125):

    x = relu(conv1(x, edge_index, edge_attr)); data = community_pooling(get_preloaded_cluster(cluster0, batch), data)
    x = relu(conv2(...)); x, batch = max_pool_x(get_preloaded_cluster(cluster1, batch), x, batch); x = scatter_mean(x, batch)

Every stage is its own launch(es) on the device (layers.conv_layer_forward, community_pooling.*), dynamically shaped results
synchronise like the reference's -- this is the GENERAL path: it takes every constructor option of the layers
(``GINetConvLayer(bias=True)``, ``sGraphAttentionLayer(bias=False, undirected=False)``, ``FoutLayer(bias=False)``), which the
fused step kernels (fused_autograd) and the launch pair (functional.net_body) do not.  The nets switch to it when a layer was
built with such an option; with the options the reference nets build themselves it computes what the fused path computes
(tests/test_gpu_dropin.py).
"""
import torch
import torch.nn.functional as F

from .community_pooling import community_pooling, get_preloaded_cluster, max_pool_x, scatter_mean

__all__ = ["composed_forward", "default_layers"]


def default_layers(net):
    """True when every convolution of ``net`` has the options the reference nets build (what the fused kernels implement)."""
    for conv in (getattr(net, n, None) for n in ("conv1", "conv2", "conv1_ext", "conv2_ext")):
        if conv is None:
            continue
        name = type(conv).__name__
        if name == "GINetConvLayer" and conv.fc.bias is not None:
            return False
        if name == "sGraphAttentionLayer" and (conv.bias is None or not conv.undirected):
            return False
        if name == "FoutLayer" and conv.bias is None:
            return False
    return True


def _conv(layer, data):
    if type(layer).__name__ == "FoutLayer":
        return layer(data.x, data.edge_index)                      # foutnet.py:108 (no edge_attr)
    return layer(data.x, data.edge_index, data.edge_attr)


def _branch(conv1, conv2, data):
    act = F.relu
    data.x = act(_conv(conv1, data))
    cluster = get_preloaded_cluster(data.cluster0.clone(), data.batch)       # (the function offsets in place, like the reference's)
    data = community_pooling(cluster, data)
    data.x = act(_conv(conv2, data))
    cluster = get_preloaded_cluster(data.cluster1.clone(), data.batch)
    x, batch = max_pool_x(cluster, data.x, data.batch)
    return scatter_mean(x, batch, dim=0)


def composed_forward(net, data):
    """pred [B, out] of GINet / sGAT / FoutNet through the function-level API; ``data`` is left untouched (the branches work on
    shallow copies: the reference mutates ``data.x`` of the caller's batch, ginet.py:104)."""
    import copy
    pairs = [(net.conv1, net.conv2)]
    if hasattr(net, "conv1_ext"):
        pairs.append((net.conv1_ext, net.conv2_ext))
    outs = [_branch(c1, c2, copy.copy(data)) for c1, c2 in pairs]
    x = outs[0] if len(outs) == 1 else torch.cat(outs, dim=1)
    x = F.relu(net.fc1(x))
    if hasattr(net, "dropout"):
        x = F.dropout(x, net.dropout, training=net.training)      # ginet.py:138
    return net.fc2(x)
