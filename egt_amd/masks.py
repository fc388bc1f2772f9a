"""Mask producers of the EGT attention path — the model code around the
layer creates these in the reference; they define what `mask` / `M` mean to the kernels, so
they are part of the path's contract and are bit-exact (integer / boolean work).

    node_mask_from_features   Neg1MaskedEmbedding.compute_mask   lib/base/xformer_layers/masking.py:35-43
    node_mask_from_masking    keras.layers.Masking(mask_value=-1) lib/models/cifar10/dc.py:69
    constrained_edge_mask     AdjMatModel.get_edge_mask           lib/models/graph_model_base.py:131-142
    (num_virtual_nodes > 0)   VirtualNodeEmbedding.compute_mask   lib/base/graph_layers/virtual_nodes.py:47-50
                              VNModel.get_edge_mask               lib/models/graph_model_base.py:248-268

All three run as HIP kernels through the C-ABI (egt_masks.hip); there is no CPU path.
"""
from __future__ import annotations

import torch

from . import _lib as L
from .functional import _need_gpu


def node_mask_from_features(node_features: torch.Tensor, num_virtual_nodes: int = 0) -> torch.Tensor:
    """[B,N] integer node features (padding value -1) -> [B, nv+N] bool, True = real node:
    `(x + 1) != 0` (Embedding(mask_zero=True) applied to inputs+1), virtual nodes always True."""
    _need_gpu(node_features)
    if node_features.dim() != 2:
        raise ValueError("node_features must be [B,N]")
    if node_features.dtype.is_floating_point or node_features.dtype == torch.bool:
        raise TypeError("node_features must be an integer tensor (use node_mask_from_masking for float rows)")
    x = node_features.to(torch.int32).contiguous()
    B, N = x.shape
    out = torch.empty(B, N + num_virtual_nodes, dtype=torch.uint8, device=x.device)
    L.check(L.load().egt_node_mask_from_features(L.ptr(x), B, N, int(num_virtual_nodes), L.ptr(out),
                                                L.current_stream()))
    return out.view(torch.bool)


def node_mask_from_masking(node_features: torch.Tensor, mask_value: float = -1.0,
                           num_virtual_nodes: int = 0) -> torch.Tensor:
    """[B,N,F] float feature rows -> [B, nv+N] bool: keras.layers.Masking keeps a step when ANY of
    its features differs from mask_value."""
    _need_gpu(node_features)
    if node_features.dim() != 3:
        raise ValueError("node_features must be [B,N,F]")
    x = node_features.to(torch.float32).contiguous()
    B, N, F = x.shape
    out = torch.empty(B, N + num_virtual_nodes, dtype=torch.uint8, device=x.device)
    L.check(L.load().egt_node_mask_from_float_features(L.ptr(x), B, N, F, float(mask_value),
                                                      int(num_virtual_nodes), L.ptr(out), L.current_stream()))
    return out.view(torch.bool)


def constrained_edge_mask(adj: torch.Tensor, num_heads: int = 8, num_virtual_nodes: int = 0) -> torch.Tensor:
    """[B,N,N] adjacency -> M [B, nv+N, nv+N, H] fp32 (the 4th input of the 'constrained' variant):
    adjacency tiled over heads; rows and columns of virtual nodes are all ones."""
    _need_gpu(adj)
    if adj.dim() != 3 or adj.shape[1] != adj.shape[2]:
        raise ValueError("adj must be [B,N,N]")
    a = adj.to(torch.float32).contiguous()
    B, N, _ = a.shape
    NO = N + num_virtual_nodes
    M = torch.empty(B, NO, NO, num_heads, dtype=torch.float32, device=a.device)
    L.check(L.load().egt_constrained_edge_mask(L.ptr(a), B, N, int(num_heads), int(num_virtual_nodes),
                                              L.ptr(M), L.current_stream()))
    return M
