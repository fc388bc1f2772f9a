"""Channel FFN of the reference's ffn_block:
    y = x + Dense_2(act(Dense_1(LayerNorm(x))))
(lib/models/graph_xformer_model_base.py:230-258, applied per channel type by ffn_block :309-324;
pre-norm, no cross-talk, ffn_multiplier 2).  One C-ABI call per direction (egt_ffn_fwd / egt_ffn_bwd
in include/egt_amd.h) for widths 8/16/32/48/64, fp32, elu / relu; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import math

import torch
from torch import nn

from . import _lib as L
from .functional import _f32c, _need_gpu

_ACT = {"relu": L.ACT_RELU, "elu": L.ACT_ELU}
# how the matrix products are evaluated (include/egt_amd.h EGT_MM_*): "f32" exact fp32 MFMA; "bf16x3" 3-term
# bfloat16 split on the bf16 matrix pipe (per-product error 2^-16, same parity tolerances as fp32);
# "bf16" plain bfloat16 products (rtol 2e-2).  Tensors and accumulation are fp32 in every mode.
_MM = {"f32": L.MM_F32, "bf16x3": L.MM_BF16X3, "bf16": L.MM_BF16}


def _desc(rows: int, width: int, activation: str, eps: float, matmul: str = "f32") -> L.FfnDesc:
    if activation not in _ACT:
        raise ValueError(f"fused FFN activation must be one of {sorted(_ACT)} (got {activation!r})")
    if matmul not in _MM:
        raise ValueError(f"matmul must be one of {sorted(_MM)} (got {matmul!r})")
    return L.FfnDesc(rows=rows, width=width, dtype=L.EGT_F32, activation=_ACT[activation], ln_eps=eps,
                     matmul=_MM[matmul], flags=0)


def _pstruct(tensors) -> L.FfnParams:
    st = L.FfnParams()
    for name, t in zip(L.FFN_PARAM_FIELDS, tensors):
        setattr(st, name, t.data_ptr())
    return st


class _FusedFFN(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, desc, *params):
        _need_gpu(x)
        lib = L.load()
        x = _f32c(x)
        ctx.param_objs = params
        params = tuple(_f32c(p) for p in params)
        y = torch.empty_like(x)
        ws = torch.empty(lib.egt_ffn_workspace_bytes(C.byref(desc)), dtype=torch.uint8, device=x.device)
        pst = _pstruct(params)
        L.check(lib.egt_ffn_fwd(C.byref(desc), C.byref(pst), L.ptr(x), L.ptr(y), L.ptr(ws), L.current_stream()))
        ctx.desc = desc
        ctx.ws = ws   # the backward reuses the prepared operands in it (EGT_FFN_WS_PREPARED: one launch less)
        ctx.save_for_backward(x, *params)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.load()
        x, *params = ctx.saved_tensors
        desc = ctx.desc
        dy = _f32c(dy)
        dx = torch.empty_like(x)
        from .fused import grad_sinks
        grads, rets = grad_sinks(ctx.param_objs)
        bdesc = L.FfnDesc.from_buffer_copy(desc)
        bdesc.flags = L.FFN_WS_PREPARED
        pst, gst = _pstruct(params), _pstruct(grads)
        L.check(lib.egt_ffn_bwd(C.byref(bdesc), C.byref(pst), L.ptr(x), L.ptr(dy), L.ptr(dx), C.byref(gst),
                                L.ptr(ctx.ws), L.current_stream()))
        return (dx, None, *rets)


def ffn(x, norm_gamma, norm_beta, lr1_kernel, lr1_bias, lr2_kernel, lr2_bias, activation="elu", eps=1e-3, matmul="f32"):
    """x: [..., W] -> [..., W]; kernels in Keras layout [in, out]."""
    W = x.shape[-1]
    rows = x.numel() // W
    params = (norm_gamma, norm_beta, lr1_kernel, lr1_bias, lr2_kernel, lr2_bias)
    desc = _desc(rows, W, activation, eps, matmul)
    if not L.load().egt_ffn_supported(C.byref(desc)):
        raise ValueError(f"fused FFN covers widths 8/16/32/48/64 in fp32 (width 8: exact fp32 products only); "
                         f"got width {W}, dtype {x.dtype}, matmul {matmul!r}")
    return _FusedFFN.apply(x, desc, *params)


class FFN(nn.Module):
    """ffnlr1 -> ffnact -> ffnlr2 of one channel type; parameters under the Keras layer names
    (norm_fnn_<tag>, fnn_lr1_<tag>, fnn_lr2_<tag>: keras_named_parameters)."""

    def __init__(self, width: int, ffn_multiplier: float = 2.0, activation: str = "elu", matmul: str = "f32"):
        super().__init__()
        hid = round(width * ffn_multiplier)
        if hid != 2 * width:
            raise ValueError("fused FFN is built for ffn_multiplier = 2")
        if activation not in _ACT:
            raise ValueError(f"fused FFN activation must be one of {sorted(_ACT)} (got {activation!r})")
        # fail at construction for a width the kernels do not cover (there is no composed fallback);
        # the C library decides (egt_ffn_supported), so the Python side never drifts from it
        if not L.load().egt_ffn_supported(C.byref(_desc(16, width, activation, 1e-3, matmul if matmul in _MM else "f32"))):
            raise ValueError(f"fused FFN does not cover width {width} with matmul={matmul!r} (fp32; width 8: exact products only)")
        if matmul not in _MM:
            raise ValueError(f"matmul must be one of {sorted(_MM)} (got {matmul!r})")
        self.width, self.activation, self.matmul = width, activation, matmul
        self.norm_gamma = nn.Parameter(torch.ones(width))
        self.norm_beta = nn.Parameter(torch.zeros(width))
        lim1 = math.sqrt(6.0 / (width + hid))            # Keras Dense default: glorot_uniform
        self.lr1_kernel = nn.Parameter(torch.empty(width, hid).uniform_(-lim1, lim1))
        self.lr1_bias = nn.Parameter(torch.zeros(hid))
        self.lr2_kernel = nn.Parameter(torch.empty(hid, width).uniform_(-lim1, lim1))
        self.lr2_bias = nn.Parameter(torch.zeros(width))

    def keras_named_parameters(self, tag: str):
        return {f"norm_fnn_{tag}/gamma": self.norm_gamma, f"norm_fnn_{tag}/beta": self.norm_beta,
                f"fnn_lr1_{tag}/kernel": self.lr1_kernel, f"fnn_lr1_{tag}/bias": self.lr1_bias,
                f"fnn_lr2_{tag}/kernel": self.lr2_kernel, f"fnn_lr2_{tag}/bias": self.lr2_bias}

    def forward(self, x):
        return ffn(x, self.norm_gamma, self.norm_beta, self.lr1_kernel, self.lr1_bias, self.lr2_kernel,
                   self.lr2_bias, activation=self.activation, matmul=self.matmul)
