"""Fused pair operator for the large-head geometry (egt_pair_fwd / egt_pair_bwd in include/egt_amd.h):
(V_att, e') = norm_edge -> attention_gates / dense_edge_b -> EGT -> dense_edge_r + res_edge in ONE pair kernel per
direction (graph_xformer_model_base.py:195-218 around egt_layers.py:57-143).  The node-side Dense layers of mha_block
stay torch GEMMs around it (EGTBlock._pair_forward)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .functional import _f32c, _u8c, _need_gpu
from .fused import _desc, _params_struct, grad_sinks

# the eight edge-side parameters, in egt_block_params order (node-side slots stay NULL)
_EDGE = (("norm_edge", "gamma"), ("norm_edge", "beta"), ("attention_gates", "kernel"), ("attention_gates", "bias"),
         ("dense_edge_b", "kernel"), ("dense_edge_b", "bias"), ("dense_edge_r", "kernel"), ("dense_edge_r", "bias"))
_SLOTS = (0, 1, 2, 3, 4, 5, 12, 13)   # their positions in BLOCK_PARAM_FIELDS


def _struct(ts):
    full = [None] * len(L.BLOCK_PARAM_FIELDS)
    for s, t in zip(_SLOTS, ts):
        full[s] = t
    return _params_struct(full)


def pair_supported(blk, h, e, attn_mask, rand_mask) -> bool:
    if blk.edge_channel_type != "residual" or not blk.gated or blk.add_n_norm or blk.edge_activation is not None:
        return False
    if blk.mha.scale_degree or blk.mha.attn_dropout > 0 or blk.mha.num_virtual_nodes > 0:
        return False
    if blk.training and (blk.node_dropout > 0 or blk.edge_dropout > 0):
        return False
    if attn_mask is not None or rand_mask is not None:
        return False
    if not (h.is_cuda and e.is_cuda) or e.dtype != torch.float32 or h.dtype != torch.float32:
        return False
    if blk.model_width % blk.num_heads:
        return False
    lib = L.load()
    if not hasattr(lib, "egt_pair_fwd"):
        return False
    d = _desc(blk, h.shape[0], h.shape[1], False, 0, e.dtype)
    return bool(lib.egt_pair_supported(C.byref(d)))


class _PairOp(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, e, key_mask, desc, *params):
        _need_gpu(qkv, e)
        lib = L.load()
        qkv = _f32c(qkv); e = _f32c(e)
        key_mask = _u8c(key_mask)
        ctx.param_objs = params
        params = tuple(_f32c(p) for p in params)
        B, N = e.shape[0], e.shape[1]
        Dh = qkv.shape[-1] // 3
        v_att = torch.empty(B, N, Dh, device=e.device, dtype=torch.float32)
        e_out = torch.empty_like(e)
        rowstats = torch.empty(B, N, desc.H, 4, device=e.device, dtype=torch.float32)
        desc.reserved = L.ATTN_WS_SHARED if any(ctx.needs_input_grad[:2]) or any(ctx.needs_input_grad[4:]) else 0
        ws = torch.empty(lib.egt_pair_workspace_bytes(C.byref(desc)), dtype=torch.uint8, device=e.device)
        L.check(lib.egt_pair_fwd(C.byref(desc), C.byref(_struct(params)), L.ptr(qkv), L.ptr(e), L.ptr(key_mask),
                                 L.ptr(v_att), L.ptr(e_out), L.ptr(rowstats), L.ptr(ws), L.current_stream()))
        ctx.desc = desc
        ctx.save_for_backward(qkv, e, key_mask, v_att, rowstats, ws if desc.reserved else None, *params)
        return v_att, e_out

    @staticmethod
    def backward(ctx, d_v_att, d_e_out):
        lib = L.load()
        qkv, e, key_mask, v_att, rowstats, ws, *params = ctx.saved_tensors
        desc = ctx.desc
        d_v_att = torch.zeros_like(v_att) if d_v_att is None else _f32c(d_v_att)
        d_e_out = torch.zeros_like(e) if d_e_out is None else _f32c(d_e_out)
        if ws is None:
            desc.reserved = 0
            ws = torch.empty(lib.egt_pair_workspace_bytes(C.byref(desc)), dtype=torch.uint8, device=e.device)
        d_qkv = torch.empty_like(qkv)
        d_e = torch.empty_like(e)
        bufs, rets = grad_sinks(ctx.param_objs)
        L.check(lib.egt_pair_bwd(C.byref(desc), C.byref(_struct(params)), L.ptr(qkv), L.ptr(e), L.ptr(key_mask),
                                 L.ptr(v_att), L.ptr(rowstats), L.ptr(d_v_att), L.ptr(d_e_out), L.ptr(d_qkv), L.ptr(d_e),
                                 C.byref(_struct(bufs)), L.ptr(ws), L.current_stream()))
        return (d_qkv, d_e, None, None) + tuple(rets)


def block_pair(blk, h, e, mask):
    """EGTBlock forward on the fused pair operator: torch node-side Dense layers (library GEMMs) around it
    (mha_block, graph_xformer_model_base.py:106-145, inside edge_update_residual, :192-223)."""
    B, N = h.shape[0], h.shape[1]
    m = blk.mha
    training = bool(blk.training and m.random_mask_prob > 0)
    seed = m.next_seed() if training else 0                 # the counter-hash stream of the fused / composed paths (oracle/rng_ref.py)
    desc = _desc(blk, B, N, training, seed)
    y = h                                                   # :107
    qkv = blk.dense_qkv(blk.norm_mha(h))                    # :109-113
    params = [getattr(getattr(blk, mod), attr) for mod, attr in _EDGE]
    v_att, e2 = _PairOp.apply(qkv, e, mask, desc, *params)  # :117-131, :195-218
    return blk.dense_mha(v_att) + y, e2                     # :136-140
