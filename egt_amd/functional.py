"""torch.autograd bindings of the HIP kernels (through the C-ABI, via ctypes).

PyTorch is plumbing here: device memory, streams and the autograd tape.  All
arithmetic on the [B,N,N,*] tensors runs in egt_amd/csrc/*.hip.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Sequence

import os

import torch

from . import _lib as L


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "egt_amd: tensors must live on a ROCm device; this package has no CPU path "
                "(the CPU restatement under oracle/ is test infrastructure only)")


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise TypeError(f"egt_amd: fp32 tensors only for now (got {t.dtype})")
    return t.contiguous()


def _u8c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if t.dtype == torch.bool:
        return t.contiguous().view(torch.uint8)    # a bool tensor IS one 0 / 1 byte per element: no conversion kernel per call
    elif t.dtype != torch.uint8:
        t = (t != 0).to(torch.uint8)
    return t.contiguous()


@dataclass(frozen=True)
class AttnConfig:
    """Operator attributes of EGT.__init__ (egt_layers.py:5-16) + run state."""
    num_heads: int = 8
    clip_logits_value: Optional[Sequence[float]] = (-5.0, 5.0)
    scale_degree: bool = False
    scaler_type: str = "log"
    num_virtual_nodes: int = 0
    random_mask_prob: float = 0.0
    attn_dropout: float = 0.0
    training: bool = False
    seed: int = 0
    need_a_tild: bool = False
    use_mfma: bool = True   # MFMA-tiled kernels where they cover the configuration


def _attn_desc(cfg: AttnConfig, B, N, d, has_E, has_G, has_M) -> L.AttnDesc:
    flags = 0
    if has_E:
        flags |= L.F_EDGE_INPUT
    if has_G:
        flags |= L.F_GATE_INPUT
    if has_M:
        flags |= L.F_ATTN_MASK
    if cfg.scale_degree:
        flags |= L.F_SCALE_DEGREE
    if cfg.scaler_type == "linear":
        flags |= L.F_SCALER_LINEAR
    if cfg.training:
        flags |= L.F_TRAINING
    lo = hi = 0.0
    if cfg.clip_logits_value is not None:
        flags |= L.F_CLIP
        lo, hi = float(cfg.clip_logits_value[0]), float(cfg.clip_logits_value[1])
    return L.AttnDesc(B=B, N=N, H=cfg.num_heads, d=d, dtype=L.EGT_F32, flags=flags,
                      clip_lo=lo, clip_hi=hi, random_mask_prob=float(cfg.random_mask_prob),
                      attn_dropout=float(cfg.attn_dropout),
                      num_virtual_nodes=int(cfg.num_virtual_nodes), reserved=0,
                      seed=int(cfg.seed) & 0xFFFFFFFFFFFFFFFF)


class _EGTAttention(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, E, G, M, key_mask, rand_mask, drop_keep, cfg: AttnConfig):
        _need_gpu(qkv, E, G, M, key_mask)
        lib = L.load()
        qkv = _f32c(qkv); E = _f32c(E); G = _f32c(G)
        M = None if M is None else _f32c(M.to(torch.float32))
        key_mask = _u8c(key_mask); rand_mask = _u8c(rand_mask); drop_keep = _u8c(drop_keep)
        B, N, C3 = qkv.shape
        H = cfg.num_heads
        assert C3 % (H * 3) == 0                      # egt_layers.py:70
        d = C3 // (H * 3)
        desc = _attn_desc(cfg, B, N, d, E is not None, G is not None, M is not None)
        mfma_ws = None
        v_att = torch.empty(B, N, d * H, device=qkv.device, dtype=torch.float32)
        h_hat = torch.empty(B, N, N, H, device=qkv.device, dtype=torch.float32)
        a_tild = torch.empty(B, N, N, H, device=qkv.device, dtype=torch.float32) if cfg.need_a_tild else None
        rowstats = torch.empty(B, N, H, 4, device=qkv.device, dtype=torch.float32)
        if (cfg.use_mfma and drop_keep is None
                and lib.egt_attn_mfma_supported(C.byref(desc), 1 if cfg.need_a_tild else 0)):
            # large-head geometry: QK^T / A.V on MFMA tiles (egt_attn_mfma.hip)
            # one workspace for both directions when a backward will follow: q/k/v are packed once (EGT_ATTN_WS_SHARED)
            # (grad mode is always off inside autograd.Function.forward: the backward is announced by ctx.needs_input_grad)
            # Memory: the shared workspace (five packed q/k/v arrays + the dO slot + the dA tiles, ~ the size of h_hat at
            # N = 512, d = 64) then lives from forward to backward of EVERY MFMA attention node of a deep model;
            # EGT_ATTN_WS_SHARED=0 opts out (the backward re-packs: one more k_attn_pack section per layer, nothing kept).
            shared = any(ctx.needs_input_grad[:3]) and os.environ.get("EGT_ATTN_WS_SHARED", "1") != "0"
            if shared:
                desc.reserved = L.ATTN_WS_SHARED
                ws = torch.empty(lib.egt_attn_mfma_workspace_bytes(C.byref(desc)), device=qkv.device, dtype=torch.uint8)
                mfma_ws = ws
            else:
                ws = torch.empty(lib.egt_attn_mfma_fwd_workspace_bytes(C.byref(desc)), device=qkv.device,
                                 dtype=torch.uint8)
            L.check(lib.egt_attn_mfma_fwd(C.byref(desc), L.ptr(qkv), L.ptr(E), L.ptr(G), L.ptr(key_mask),
                                          L.ptr(M), L.ptr(rand_mask), L.ptr(v_att), L.ptr(h_hat),
                                          L.ptr(rowstats), L.ptr(ws), L.current_stream()))
        else:
            L.check(lib.egt_attn_fwd(C.byref(desc), L.ptr(qkv), L.ptr(E), L.ptr(G), L.ptr(key_mask),
                                     L.ptr(M), L.ptr(rand_mask), L.ptr(drop_keep), L.ptr(v_att),
                                     L.ptr(h_hat), L.ptr(a_tild), L.ptr(rowstats), L.current_stream()))
        ctx.cfg = cfg
        ctx.desc = desc
        ctx.has = (E is not None, G is not None, M is not None)
        # the workspace travels with the saved tensors: freed with them when the graph is (retain_graph=False), not with ctx
        ctx.save_for_backward(qkv, E, G, M, key_mask, rand_mask, drop_keep, v_att, rowstats, mfma_ws)
        ctx.set_materialize_grads(False)
        if a_tild is None:
            a_tild = torch.empty(0, device=qkv.device)
        ctx.mark_non_differentiable(a_tild)
        return v_att, h_hat, a_tild

    @staticmethod
    def backward(ctx, d_v_att, d_h_hat, _d_a_tild):
        lib = L.load()
        qkv, E, G, M, key_mask, rand_mask, drop_keep, v_att, rowstats, mfma_ws = ctx.saved_tensors
        desc = ctx.desc
        if d_v_att is None:
            d_v_att = torch.zeros_like(v_att)
        d_v_att = _f32c(d_v_att)
        d_h_hat = _f32c(d_h_hat)
        d_qkv = torch.empty_like(qkv)
        d_E = torch.empty_like(E) if E is not None else None
        d_G = torch.empty_like(G) if G is not None else None
        mfma = bool(ctx.cfg.use_mfma and drop_keep is None and lib.egt_attn_mfma_supported(C.byref(desc), 0))
        if mfma and mfma_ws is not None:
            ws = mfma_ws                          # the forward's workspace: q/k/v operand copies already in place
        else:
            desc.reserved = 0
            nbytes = (lib.egt_attn_mfma_workspace_bytes if mfma else lib.egt_attn_bwd_workspace_bytes)(C.byref(desc))
            ws = torch.empty(nbytes, device=qkv.device, dtype=torch.uint8)
        if mfma:
            L.check(lib.egt_attn_mfma_bwd(C.byref(desc), L.ptr(qkv), L.ptr(E), L.ptr(G), L.ptr(key_mask),
                                          L.ptr(M), L.ptr(rand_mask), L.ptr(v_att), L.ptr(rowstats),
                                          L.ptr(d_v_att), L.ptr(d_h_hat), L.ptr(d_qkv), L.ptr(d_E),
                                          L.ptr(d_G), L.ptr(ws), L.current_stream()))
        else:
            L.check(lib.egt_attn_bwd(C.byref(desc), L.ptr(qkv), L.ptr(E), L.ptr(G), L.ptr(key_mask),
                                     L.ptr(M), L.ptr(rand_mask), L.ptr(drop_keep), L.ptr(v_att),
                                     L.ptr(rowstats), L.ptr(d_v_att), L.ptr(d_h_hat), L.ptr(d_qkv),
                                     L.ptr(d_E), L.ptr(d_G), L.ptr(ws), L.current_stream()))
        return d_qkv, d_E, d_G, None, None, None, None, None


def egt_attention(qkv, E=None, G=None, M=None, key_mask=None, *, cfg: AttnConfig,
                  rand_mask=None, drop_keep=None):
    """(V_att, H_hat, A_tild) = EGT([QKV,E,G,M], mask) — egt_layers.py:57-213."""
    return _EGTAttention.apply(qkv, E, G, M, key_mask, rand_mask, drop_keep, cfg)


def mask_sample(which: int, seed: int, prob: float, B: int, N: int, H: int, device="cuda"):
    """Materialise the in-kernel sample stream (uint8 [B,N,N,H])."""
    lib = L.load()
    out = torch.empty(B, N, N, H, dtype=torch.uint8, device=device)
    L.check(lib.egt_mask_sample(int(which), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                                C.c_float(prob), B, N, H, L.ptr(out), L.current_stream()))
    return out


# ------------------------------------------------------------------ edge ops ---
_ACTS = {None: (L.ACT_NONE, 0.0), "relu": (L.ACT_RELU, 0.0), "elu": (L.ACT_ELU, 0.0)}


def _act_code(edge_activation):
    if edge_activation is None:
        return L.ACT_NONE, 0.0
    ea = edge_activation.lower()
    if ea.startswith("lrelu"):                      # graph_xformer_model_base.py:150-156
        return L.ACT_LRELU, float(ea[-1]) / 10
    if ea in _ACTS:
        return _ACTS[ea]
    raise ValueError(f"unsupported edge_activation {edge_activation}")


def _edge_desc(e, use_ln, gates, act, alpha, eps) -> L.EdgeDesc:
    De = e.shape[-1]
    rows = e.numel() // De
    flags = (L.EP_LAYERNORM if use_ln else 0) | (L.EP_GATES if gates else 0)
    return L.EdgeDesc(rows=rows, De=De, H=8, dtype=L.EGT_F32, flags=flags, act=act,
                      act_alpha=alpha, ln_eps=eps, reserved=0)


class _EdgeProj(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, gamma, beta, Wg, bg, We, be, use_ln, edge_activation, eps, passthrough):
        _need_gpu(e)
        lib = L.load()
        e_in = e
        e = _f32c(e)
        gates = Wg is not None
        act, alpha = _act_code(edge_activation)
        H = We.shape[1]
        if H != 8:
            raise AssertionError("edge kernels are built for num_heads=8")
        desc = _edge_desc(e, use_ln, gates, act, alpha, eps)
        shp = e.shape[:-1] + (H,)
        G = torch.empty(shp, device=e.device, dtype=torch.float32) if gates else None
        E = torch.empty(shp, device=e.device, dtype=torch.float32)
        ps = [_f32c(t) for t in (gamma, beta, Wg, bg, We, be)]
        L.check(lib.egt_edge_proj_fwd(C.byref(desc), L.ptr(e), *[L.ptr(p) for p in ps],
                                      L.ptr(G), L.ptr(E), L.current_stream()))
        ctx.desc = desc
        ctx.gates = gates
        ctx.use_ln = use_ln
        ctx.save_for_backward(e, ps[0], ps[1], ps[2], ps[4], E if act != L.ACT_NONE else None)
        ctx.set_materialize_grads(False)
        if G is None:
            G = torch.empty(0, device=e.device)
            ctx.mark_non_differentiable(G)
        if passthrough:
            # e handed on to the residual update: its gradient comes back into THIS backward and is
            # folded into d_e by the kernel (egt_edge_proj_bwd_acc) instead of a separate e-sized add
            thru = e_in.view_as(e_in)
        else:
            thru = torch.empty(0, device=e.device)
            ctx.mark_non_differentiable(thru)
        return G, E, thru

    @staticmethod
    def backward(ctx, dG, dE, d_thru):
        lib = L.load()
        e, gamma, beta, Wg, We, E_out = ctx.saved_tensors
        desc = ctx.desc
        De = e.shape[-1]
        if dE is None:
            dE = torch.zeros(e.shape[:-1] + (8,), device=e.device)
        if ctx.gates and dG is None:
            dG = torch.zeros(e.shape[:-1] + (8,), device=e.device)
        dE = _f32c(dE); dG = _f32c(dG) if ctx.gates else None
        d_e = torch.empty_like(e)
        mk = lambda *s: torch.empty(*s, device=e.device, dtype=torch.float32)
        d_gamma = mk(De) if ctx.use_ln else None
        d_beta = mk(De) if ctx.use_ln else None
        d_Wg = mk(De, 8) if ctx.gates else None
        d_bg = mk(8) if ctx.gates else None
        d_We, d_be = mk(De, 8), mk(8)
        ws = torch.empty(lib.egt_edge_proj_bwd_workspace_bytes(C.byref(desc)), device=e.device,
                         dtype=torch.uint8)
        base = _f32c(d_thru) if d_thru is not None else None
        L.check(lib.egt_edge_proj_bwd_acc(C.byref(desc), L.ptr(e), L.ptr(gamma), L.ptr(beta), L.ptr(Wg),
                                          L.ptr(We), L.ptr(E_out), L.ptr(dG), L.ptr(dE), L.ptr(base), L.ptr(d_e),
                                          L.ptr(d_gamma), L.ptr(d_beta), L.ptr(d_Wg), L.ptr(d_bg),
                                          L.ptr(d_We), L.ptr(d_be), L.ptr(ws), L.current_stream()))
        return d_e, d_gamma, d_beta, d_Wg, d_bg, d_We, d_be, None, None, None, None


def edge_proj(e, gamma, beta, Wg, bg, We, be, *, use_ln=True, edge_activation=None, eps=1e-3,
              passthrough=False):
    """(G, E): [norm_edge] -> attention_gates, dense_edge_b
    (graph_xformer_model_base.py:195,201-204,149-162).  Wg/bg None => no gates.
    passthrough=True returns (G, E, e_thru): e_thru is `e` itself, to be used by the residual
    update that follows (:218) so that the two gradient branches of e are summed inside the
    projection-backward kernel rather than by a separate elementwise add over [B,N,N,De]."""
    G, E, thru = _EdgeProj.apply(e, gamma, beta, Wg, bg, We, be, use_ln, edge_activation, eps, passthrough)
    G = G if Wg is not None else None
    return (G, E, thru) if passthrough else (G, E)


class _EdgeUpdate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e, h_hat, Wr, br):
        _need_gpu(e, h_hat)
        lib = L.load()
        e = _f32c(e); h_hat = _f32c(h_hat); Wr = _f32c(Wr); br = _f32c(br)
        desc = _edge_desc(e, False, False, L.ACT_NONE, 0.0, 1e-3)
        out = torch.empty_like(e)
        L.check(lib.egt_edge_update_fwd(C.byref(desc), L.ptr(e), L.ptr(h_hat), L.ptr(Wr), L.ptr(br),
                                        L.ptr(out), L.current_stream()))
        ctx.desc = desc
        ctx.save_for_backward(h_hat, Wr)
        return out

    @staticmethod
    def backward(ctx, d_out):
        lib = L.load()
        h_hat, Wr = ctx.saved_tensors
        desc = ctx.desc
        d_out = _f32c(d_out)
        d_h = torch.empty_like(h_hat)
        d_Wr = torch.empty_like(Wr)
        d_br = torch.empty(Wr.shape[1], device=Wr.device, dtype=torch.float32)
        ws = torch.empty(lib.egt_edge_update_bwd_workspace_bytes(C.byref(desc)), device=Wr.device,
                         dtype=torch.uint8)
        L.check(lib.egt_edge_update_bwd(C.byref(desc), L.ptr(d_out), L.ptr(h_hat), L.ptr(Wr),
                                        L.ptr(d_h), L.ptr(d_Wr), L.ptr(d_br), L.ptr(ws),
                                        L.current_stream()))
        return d_out, d_h, d_Wr, d_br


def edge_update(e, h_hat, Wr, br):
    """e' = e + H_hat·Wr + br  (dense_edge_r + res_edge, :214-218)."""
    return _EdgeUpdate.apply(e, h_hat, Wr, br)
