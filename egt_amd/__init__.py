"""egt_amd — MI355X-native EGT edge-augmented attention hot path.

Public surface (mirrors the reference's operator interface for this path):
    EGT, EGTBlock, EGTStack, custom_layers      (egt_amd.layers)
    egt_attention, edge_proj, edge_update        (egt_amd.functional)
    FlatGradAllReduce, shard_batch               (egt_amd.dp)
    FFN, ffn                                     (egt_amd.ffn: the ffn_block step after the attention block)
    node_mask_from_features, node_mask_from_masking, constrained_edge_mask   (egt_amd.masks: mask producers)
    DeviceSeeds, GraphedStep                     (egt_amd.graph: hipGraph capture of a step, device-resident mask seeds)
"""
from .layers import EGT, EGTBlock, EGTStack, EGTLayerStack, custom_layers, KerasDense, KerasLayerNorm  # noqa: F401
from .functional import AttnConfig, egt_attention, edge_proj, edge_update, mask_sample  # noqa: F401
from .ffn import FFN, ffn  # noqa: F401
from .model import (ZincDCTransformer, PatternDCTransformer, Cifar10DCTransformer, sparse_xent_loss, edge_embed, mae_loss, weighted_sparse_xent_loss,  # noqa: F401
                    class_weights_from_sizes)
from .graph import DeviceSeeds, GraphedStep  # noqa: F401
from .masks import node_mask_from_features, node_mask_from_masking, constrained_edge_mask  # noqa: F401

__all__ = ["EGT", "EGTBlock", "EGTStack", "EGTLayerStack", "custom_layers", "AttnConfig", "egt_attention",
           "edge_proj", "edge_update", "mask_sample", "FFN", "ffn", "node_mask_from_features",
           "node_mask_from_masking", "constrained_edge_mask", "ZincDCTransformer", "PatternDCTransformer", "Cifar10DCTransformer", "sparse_xent_loss", "edge_embed", "mae_loss",
           "weighted_sparse_xent_loss", "class_weights_from_sizes", "DeviceSeeds", "GraphedStep"]
