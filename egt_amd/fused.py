"""Fused attention block: one C-ABI call per layer per direction
(egt_block_fwd / egt_block_bwd in include/egt_amd.h)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .functional import _f32c, _u8c, _need_gpu

_GRAD_ORDER = (("norm_edge", "gamma"), ("norm_edge", "beta"),
               ("attention_gates", "kernel"), ("attention_gates", "bias"),
               ("dense_edge_b", "kernel"), ("dense_edge_b", "bias"),
               ("norm_mha", "gamma"), ("norm_mha", "beta"),
               ("dense_qkv", "kernel"), ("dense_qkv", "bias"),
               ("dense_mha", "kernel"), ("dense_mha", "bias"),
               ("dense_edge_r", "kernel"), ("dense_edge_r", "bias"))


def _desc(blk, B, N, training, seed, edge_dtype=torch.float32, seed_device=None) -> L.BlockDesc:
    flags = 0
    if seed_device is not None:      # egt_amd.graph.DeviceSeeds: the kernels complete the seed from HBM
        flags |= L.BF_SEED_DEVICE
    if blk.gated:
        flags |= L.BF_GATE
    if blk.edge_channel_type == "constrained":
        flags |= L.BF_ATTN_MASK
    if training:
        flags |= L.BF_TRAINING
    if blk.edge_channel_type == "bias":
        flags |= L.BF_NO_EDGE_LN
    lo = hi = 0.0
    if blk.mha.clip_logits_value is not None:
        flags |= L.BF_CLIP
        lo, hi = float(blk.mha.clip_logits_value[0]), float(blk.mha.clip_logits_value[1])
    return L.BlockDesc(B=B, N=N, H=blk.num_heads, d=blk.model_width // blk.num_heads,
                       De=blk.edge_width, dtype=L.EGT_BF16 if edge_dtype == torch.bfloat16 else L.EGT_F32, flags=flags, clip_lo=lo, clip_hi=hi,
                       random_mask_prob=float(blk.mha.random_mask_prob), ln_eps=1e-3, reserved=0,
                       seed=int(seed) & 0xFFFFFFFFFFFFFFFF,
                       seed_device=None if seed_device is None else seed_device[0].ptr(seed_device[1]))


def block_supported(blk, h, e, attn_mask, rand_mask) -> bool:
    """Configurations the fused kernels cover (everything else composes)."""
    if blk.edge_channel_type not in ("residual", "constrained", "bias"):
        return False
    if blk.add_n_norm or blk.edge_activation is not None:
        return False
    if blk.mha.scale_degree or blk.mha.attn_dropout > 0 or blk.mha.num_virtual_nodes > 0:
        return False
    if blk.training and (blk.node_dropout > 0 or blk.edge_dropout > 0):
        return False
    if blk.edge_channel_type == "constrained" and attn_mask is None:
        return False
    if not (h.is_cuda and e.is_cuda) or e.dtype not in (torch.float32, torch.bfloat16):
        return False
    if h.dtype not in (torch.float32, torch.bfloat16):
        return False
    if blk.model_width % blk.num_heads:
        return False
    lib = L.load()
    if not hasattr(lib, "egt_block_fwd"):
        return False
    d = _desc(blk, h.shape[0], h.shape[1], False, 0, e.dtype)
    return bool(lib.egt_block_supported(C.byref(d)))


def _edge_c(t, dtype=None):
    """edge tensor as the kernels take it: contiguous fp32 or bf16 (EGT_BF16: bf16 in HBM, fp32 math)"""
    if t is None:
        return None
    want = dtype if dtype is not None else (torch.bfloat16 if t.dtype == torch.bfloat16 else torch.float32)
    return t.to(want).contiguous()


def _node_io(h):
    """node tensors are fp32 at the C boundary; a bf16 caller gets its dtype back"""
    return (h.float(), h.dtype) if h.dtype != torch.float32 else (h, None)


def _params_struct(tensors) -> L.BlockParams:
    st = L.BlockParams()
    for name, t in zip(L.BLOCK_PARAM_FIELDS, tensors):
        setattr(st, name, None if t is None else t.data_ptr())
    return st


def grad_sinks(params):
    """Where a fused backward writes each parameter gradient, and what it returns to autograd.  A parameter whose .grad is a
    pre-bound view of a flat gradient buffer opened for direct writes (FlatGradAllReduce(direct=True)) receives its gradient
    IN that view and autograd gets None (no `grad += g` launch); everything else gets a fresh tensor as before."""
    bufs, rets = [], []
    for p in params:
        if p is None:
            bufs.append(None); rets.append(None)
            continue
        g = p.grad if getattr(p, "_egt_direct_grad", False) else None
        if g is not None and g.dtype == torch.float32 and g.is_contiguous() and g.shape == p.shape and g.device == p.device:
            bufs.append(g); rets.append(None)
        else:
            t = torch.empty_like(p, dtype=torch.float32)
            bufs.append(t); rets.append(t)
    return bufs, rets


class _FusedBlock(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h, e, key_mask, attn_mask, rand_mask, desc, *params):
        _need_gpu(h, e)
        lib = L.load()
        h = _f32c(h); e = _edge_c(e)
        key_mask = _u8c(key_mask); rand_mask = _u8c(rand_mask)
        attn_mask = None if attn_mask is None else _f32c(attn_mask.to(torch.float32))
        ctx.param_objs = params          # the Parameter objects themselves (grad_sinks looks at their .grad in the backward)
        params = tuple(None if p is None else _f32c(p) for p in params)
        dev = h.device
        h_out = torch.empty_like(h)
        e_out = torch.empty_like(e)
        saved = torch.empty(lib.egt_block_saved_bytes(C.byref(desc)), dtype=torch.uint8, device=dev)
        ws = torch.empty(lib.egt_block_workspace_bytes(C.byref(desc)), dtype=torch.uint8, device=dev)
        pst = _params_struct(params)
        L.check(lib.egt_block_fwd(C.byref(desc), C.byref(pst), L.ptr(h), L.ptr(e), L.ptr(key_mask),
                                  L.ptr(attn_mask), L.ptr(rand_mask), L.ptr(h_out), L.ptr(e_out),
                                  L.ptr(saved), L.ptr(ws), L.current_stream()))
        ctx.desc = desc
        ctx.nparams = len(params)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(h, e, key_mask, attn_mask, rand_mask, saved, *params)
        return h_out, e_out

    @staticmethod
    def backward(ctx, dh_out, de_out):
        lib = L.load()
        h, e, key_mask, attn_mask, rand_mask, saved, *params = ctx.saved_tensors
        desc = ctx.desc
        dev = h.device
        if de_out is None:
            de_out = torch.zeros_like(e)               # 'bias': the caller continues with e itself
        if dh_out is None:
            dh_out = torch.zeros_like(h)
        dh_out = _f32c(dh_out); de_out = _edge_c(de_out, e.dtype)
        dh = torch.empty_like(h)
        de = torch.empty_like(e)
        grads, rets = grad_sinks(ctx.param_objs)
        ws = torch.empty(lib.egt_block_workspace_bytes(C.byref(desc)), dtype=torch.uint8, device=dev)
        pst, gst = _params_struct(params), _params_struct(grads)
        L.check(lib.egt_block_bwd(C.byref(desc), C.byref(pst), L.ptr(h), L.ptr(e), L.ptr(key_mask),
                                  L.ptr(attn_mask), L.ptr(rand_mask), L.ptr(saved), L.ptr(dh_out),
                                  L.ptr(de_out), L.ptr(dh), L.ptr(de), C.byref(gst), L.ptr(ws),
                                  L.current_stream()))
        return (dh, de, None, None, None, None, *rets)


def _block_params(blk, e):
    """the 14 C-ABI parameters of a block.  'bias' edge channels (EGT-simple) have no norm_edge and no
    dense_edge_r: identity LN parameters and a zero update are passed instead (EGT_BF_NO_EDGE_LN)."""
    params = []
    mods = blk._modules                  # direct dict reads: nn.Module.__getattr__ was a quarter of the host time of a stack step
    for mod, attr in _GRAD_ORDER:
        m = mods.get(mod)
        params.append(None if m is None else m._parameters[attr])
    if blk.edge_channel_type == "bias":
        params[0], params[1] = blk._id_gamma, blk._id_beta       # constants owned by the block (no per-call allocation)
        params[12], params[13] = blk._zero_Wr, blk._zero_br
    return params


def block_fused(blk, h, e, mask, attn_mask, rand_mask=None):
    training = blk.training and blk.mha.random_mask_prob > 0.0
    sdev = blk.mha.seed_device if (training and rand_mask is None) else None
    seed = blk.mha.next_seed() if (training and rand_mask is None and sdev is None) else 0
    desc = _desc(blk, h.shape[0], h.shape[1], training, seed, e.dtype, sdev)
    params = _block_params(blk, e)
    if blk.edge_channel_type != "constrained":
        attn_mask = None
    h, hdt = _node_io(h)
    h2, e2 = _FusedBlock.apply(h, e, mask, attn_mask, rand_mask, desc, *params)
    if blk.edge_channel_type == "bias":
        e2 = e                                         # :190 returns e0; the kernel's e' equals it (zero update)
    return (h2 if hdt is None else h2.to(hdt)), e2


# ------------------------------------------------------------------ layer stack ---
class _FusedStack(torch.autograd.Function):
    """All model_height attention blocks in one C-ABI call per direction
    (egt_stack_fwd / egt_stack_bwd)."""

    @staticmethod
    def forward(ctx, h, e, key_mask, attn_mask, desc, layers, holder, *params):
        _need_gpu(h, e)
        lib = L.load()
        h = _f32c(h); e = _edge_c(e)
        key_mask = _u8c(key_mask)
        attn_mask = None if attn_mask is None else _f32c(attn_mask.to(torch.float32))
        params = tuple(None if p is None else _f32c(p) for p in params)
        dev = h.device
        h_out, e_out = torch.empty_like(h), torch.empty_like(e)
        saved = torch.empty(lib.egt_stack_saved_bytes(C.byref(desc), layers), dtype=torch.uint8, device=dev)
        ws = torch.empty(lib.egt_stack_workspace_bytes(C.byref(desc), layers), dtype=torch.uint8, device=dev)
        parr = (L.BlockParams * layers)(*[_params_struct(params[14 * i:14 * i + 14]) for i in range(layers)])
        L.check(lib.egt_stack_fwd(C.byref(desc), layers, parr, L.ptr(h), L.ptr(e), L.ptr(key_mask),
                                  L.ptr(attn_mask), L.ptr(h_out), L.ptr(e_out), L.ptr(saved), L.ptr(ws),
                                  L.current_stream()))
        ctx.desc, ctx.layers, ctx.holder = desc, layers, holder
        ctx.parr = parr
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(h, e, key_mask, attn_mask, saved, *params)
        return h_out, e_out

    @staticmethod
    def backward(ctx, dh_out, de_out):
        lib = L.load()
        h, e, key_mask, attn_mask, saved, *params = ctx.saved_tensors
        desc, layers = ctx.desc, ctx.layers
        dev = h.device
        if de_out is None:
            de_out = torch.zeros_like(e)
        if dh_out is None:
            dh_out = torch.zeros_like(h)
        dh_out = _f32c(dh_out); de_out = _edge_c(de_out, e.dtype)
        dh, de = torch.empty_like(h), torch.empty_like(e)
        sink = getattr(ctx.holder, "sink", None) if ctx.holder is not None else None
        if sink is not None:
            # EGTStack.bind_flat_gradients(): every parameter's .grad already IS its view of the holder's persistent flat
            # buffer.  The C backward writes there (addresses by arithmetic: no per-parameter view, data_ptr() or
            # AccumulateGrad node — two thirds of the eager step's host time) and autograd gets nothing to accumulate.
            sizes = [p.numel() for p in params if p is not None]
            if sink.numel() != sum(sizes) or sink.device != dev:
                raise RuntimeError("bound flat gradient buffer does not match the stack's parameters; call bind_flat_gradients() again")
            base, off, ptrs = sink.data_ptr(), 0, []
            for p in params:
                if p is None:
                    ptrs.append(None)
                else:
                    ptrs.append(base + 4 * off)
                    off += p.numel()
            garr = C.cast((C.c_void_p * len(ptrs))(*ptrs), C.POINTER(L.BlockParams))
            # the parameters' .grad must still BE the views of the buffer: an optimizer.zero_grad() (set_to_none=True is torch's
            # default) between bind and backward would otherwise leave every stack parameter without a gradient, silently.
            # Checked on the first and last parameter only (the per-parameter walk is the host time this path exists to avoid)
            sp = getattr(ctx.holder, "sink_params", None)
            if sp:
                g0, g1 = sp[0].grad, sp[-1].grad
                if g0 is None or g1 is None or g0.data_ptr() != base or g1.data_ptr() != base + 4 * (off - sp[-1].numel()):
                    o2 = 0
                    for q in sp:
                        q.grad = sink[o2:o2 + q.numel()].view_as(q)
                        o2 += q.numel()
            ws = torch.empty(lib.egt_stack_workspace_bytes(C.byref(desc), layers), dtype=torch.uint8, device=dev)
            L.check(lib.egt_stack_bwd(C.byref(desc), layers, ctx.parr, L.ptr(h), L.ptr(e), L.ptr(key_mask),
                                      L.ptr(attn_mask), L.ptr(saved), L.ptr(dh_out), L.ptr(de_out), L.ptr(dh),
                                      L.ptr(de), garr, L.ptr(ws), L.current_stream()))
            ctx.holder.flat = sink
            return (dh, de, None, None, None, None, None, *([None] * len(params)))
        # every parameter gradient is a view of ONE flat buffer: the data-parallel all-reduce
        # (egt_amd.dp) runs on it directly, and autograd adopts the views without copies
        sizes = [p.numel() for p in params if p is not None]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=dev)
        pieces = iter(flat.split(sizes))          # one call for all views (the per-parameter slicing was ~0.4 ms of host time per step)
        grads = []
        for p in params:
            if p is None:
                grads.append(None)
            else:
                v = next(pieces)
                grads.append(v if p.dim() == 1 else v.view(p.shape))
        if ctx.holder is not None:
            ctx.holder.flat = flat
        ws = torch.empty(lib.egt_stack_workspace_bytes(C.byref(desc), layers), dtype=torch.uint8, device=dev)
        parr = ctx.parr                            # the forward's struct array: same parameter tensors (kept alive by saved_tensors)
        garr = (L.BlockParams * layers)(*[_params_struct(grads[14 * i:14 * i + 14]) for i in range(layers)])
        L.check(lib.egt_stack_bwd(C.byref(desc), layers, parr, L.ptr(h), L.ptr(e), L.ptr(key_mask),
                                  L.ptr(attn_mask), L.ptr(saved), L.ptr(dh_out), L.ptr(de_out), L.ptr(dh),
                                  L.ptr(de), garr, L.ptr(ws), L.current_stream()))
        return (dh, de, None, None, None, None, None, *grads)


def stack_supported(stack, h, e, attn_mask) -> bool:
    blocks = list(stack.blocks)
    if not blocks or not hasattr(L.load(), "egt_stack_fwd"):
        return False
    b0 = blocks[0]
    for b in blocks:
        if b.fused is False or b.fused == "off":
            return False
        if not block_supported(b, h, e, attn_mask, None):
            return False
        same = (b.model_width == b0.model_width and b.edge_width == b0.edge_width and
                b.edge_channel_type == b0.edge_channel_type and b.gated == b0.gated and
                b.mha.clip_logits_value == b0.mha.clip_logits_value and
                b.mha.random_mask_prob == b0.mha.random_mask_prob and b.training == b0.training)
        if not same:
            return False
    return True


def stack_fused(stack, h, e, mask, attn_mask):
    blocks = list(stack.blocks)
    b0 = blocks[0]
    training = b0.training and b0.mha.random_mask_prob > 0.0
    sdev = b0.mha.seed_device if training else None
    seed = b0.mha.next_seed() if (training and sdev is None) else 0
    desc = _desc(b0, h.shape[0], h.shape[1], training, seed, e.dtype, sdev)
    params = []
    for blk in blocks:
        params += _block_params(blk, e)
    if b0.edge_channel_type != "constrained":
        attn_mask = None
    h, hdt = _node_io(h)
    h2, e2 = _FusedStack.apply(h, e, mask, attn_mask, desc, len(blocks), stack.grad_holder, *params)
    if b0.edge_channel_type == "bias":
        e2 = e
    return (h2 if hdt is None else h2.to(hdt)), e2


def layer_seed(seed: int, layer: int) -> int:
    """Seed of layer `layer` inside egt_stack_* (mirrors egt_block.hip:layer_seed)."""
    return (seed ^ (0x9E3779B97F4A7C15 * (layer + 1))) & 0xFFFFFFFFFFFFFFFF
