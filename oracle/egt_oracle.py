"""CPU oracle for the EGT edge-augmented attention hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``egt_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / the timed CPU baseline.

PARITY UNPINNED: the reference (shamim-hussain/egt) ships no tests, golden
vectors or saved weights, and its arithmetic lives in TensorFlow
(tensorflow-gpu 2.1.0, environment.yml:187-189), which is not installed in this
image.  This file therefore restates the reference's op sequence from its
source, op by op, in torch-CPU (dtype-parametric: fp64 for checking, fp32 for
the timed baseline).  It is pinned only by self-consistency (hand-derived
backward vs autograd, algebraic identities, SDPA cross-check) — see
tests/test_oracle.py — and by the committed fixtures generated from it.

Reference lines followed (relative to /root/reference):
  * inner op   lib/models/egt_layers.py:57-143 (gated), :145-213 (ungated)
  * outer op   lib/models/graph_xformer_model_base.py:106-145 (mha_block),
               :149-162 (edge_channel_contrib), :164-171 (edge_update_none),
               :173-190 (edge_update_bias), :192-223 (edge_update_residual)
  * masks      lib/base/xformer_layers/masking.py:35-43,
               lib/models/graph_model_base.py:131-142
Keras defaults hard-coded because Keras is absent: LayerNormalization(axis=-1,
epsilon=1e-3), Dense = x @ W + b with W:[in,out], softmax max-subtracted,
clip_by_value gradient passes where lo <= x <= hi, inverted dropout scaling.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch

NEG = 1e9  # the reference's additive mask constant (egt_layers.py:92,99,106)


# --------------------------------------------------------------------------
# inner op: EGT.call_gated / call_ungated
# --------------------------------------------------------------------------
def _add_mask(x, m):
    """x + m as the reference computes it.  The reference runs in fp32, where a logit plus -1e9
    ROUNDS to exactly -1e9 (|x| < 32; SURVEY.md 8(a) a5, probed): a row whose keys are all masked
    therefore gets a UNIFORM softmax over its least-masked keys (egt_layers.py:111 on such rows;
    visible in the ungated variant, :145-213).  Evaluated in fp64 the plain sum would keep the
    logit differences, so at masked positions the fp32 rounding is reproduced; unmasked positions
    keep the evaluation precision.  Differentiable like the plain add."""
    if x.dtype == torch.float64:
        y32 = (x.to(torch.float32) + m.to(torch.float32)).to(torch.float64)
        return torch.where(m != 0, y32, x)
    return x + m


def egt_forward(QKV: torch.Tensor,
                E: Optional[torch.Tensor],
                G: Optional[torch.Tensor],
                M: Optional[torch.Tensor],
                mask: Optional[torch.Tensor],
                *,
                num_heads: int = 8,
                clip_logits_value: Optional[Sequence[float]] = (-5.0, 5.0),
                scale_degree: bool = False,
                scaler_type: str = "log",
                num_virtual_nodes: int = 0,
                rand_mask: Optional[torch.Tensor] = None,
                drop_keep: Optional[torch.Tensor] = None,
                attn_dropout: float = 0.0):
    """Restatement of egt_layers.py:57-143 (G given) / :145-213 (G None).

    QKV [B,N,3*d*H]; E,G,M [B,N,N,H] or None; mask [B,N] bool or None.
    rand_mask [B,N,N,H] bool, True = "uniform_noise < random_mask_prob"
    (egt_layers.py:103-108) — injected instead of sampled so the branch is
    testable.  drop_keep [B,N,N,H] {0,1}: the keep mask of tf.nn.dropout
    (egt_layers.py:116-117).  Returns (V_att [B,N,d*H], H_hat, A_tild).
    """
    dt = QKV.dtype
    B, N, C = QKV.shape
    H = num_heads
    assert C % (H * 3) == 0                                   # :70
    d = C // (H * 3)                                          # :71
    Q, K, V = QKV.reshape(B, N, 3, d, H).unbind(2)            # :73-76  b,l,d,h

    A_hat = torch.einsum("bldh,bmdh->blmh", Q, K) * (d ** -0.5)   # :79
    if clip_logits_value is not None:                         # :81-82
        A_hat = torch.clamp(A_hat, clip_logits_value[0], clip_logits_value[1])

    H_hat = A_hat                                             # :85
    if E is not None:
        H_hat = H_hat + E                                     # :86

    H_hat_ = H_hat                                            # :89
    G_ = G                                                    # :90
    if mask is not None:                                      # :91-94
        mask_ = (mask[:, None, :, None].to(dt) - 1) * NEG
        H_hat_ = _add_mask(H_hat_, mask_)
        if G is not None:
            G_ = _add_mask(G_, mask_)
    if M is not None:                                         # :96-101
        M_ = (M.to(dt) - 1) * NEG
        H_hat_ = _add_mask(H_hat_, M_)
        if G is not None:
            G_ = _add_mask(G_, M_)
    if rand_mask is not None:                                 # :103-108
        random_mask_ = torch.where(rand_mask, torch.tensor(-NEG, dtype=dt),
                                   torch.tensor(0.0, dtype=dt))
        H_hat_ = _add_mask(H_hat_, random_mask_)
        if G is not None:
            G_ = _add_mask(G_, random_mask_)

    A_tild = torch.softmax(H_hat_, dim=2)                     # :111
    gates = None
    if G is not None:
        gates = torch.sigmoid(G_)                             # :112
        A_tild = A_tild * gates                               # :113

    A_drop = A_tild
    if drop_keep is not None and attn_dropout > 0.0:          # :116-117
        A_drop = A_tild * drop_keep.to(dt) / (1.0 - attn_dropout)

    V_att = torch.einsum("blmh,bmdh->bldh", A_drop, V)        # :120

    if scale_degree:                                          # :123-136
        if G is None:
            raise ValueError("scale_degree requires gate_input")
        degrees = gates.sum(dim=2, keepdim=True)              # b,l,1,h
        if scaler_type == "log":
            degree_scalers = torch.log(1 + degrees)
        elif scaler_type == "linear":
            degree_scalers = degrees
        else:
            raise ValueError(f"Unknown scaler type {scaler_type}")
        if num_virtual_nodes > 0:
            degree_scalers = torch.cat(
                [torch.ones_like(degree_scalers[:, :num_virtual_nodes]),
                 degree_scalers[:, num_virtual_nodes:]], dim=1)
        V_att = V_att * degree_scalers

    V_att = V_att.reshape(B, N, d * H)                        # :139-141
    # :116-117 / :202-203 REASSIGN A_tild = tf.nn.dropout(A_tild, ...): the third output is the
    # post-dropout matrix (identical to A_tild when attention dropout is off)
    return V_att, H_hat, A_drop


def egt_backward(QKV, E, G, M, mask, dV_att, dH_ext, *, num_heads=8,
                 clip_logits_value=(-5.0, 5.0), scale_degree=False,
                 scaler_type="log", num_virtual_nodes=0, rand_mask=None,
                 drop_keep=None, attn_dropout=0.0):
    """Hand-derived backward of egt_forward: what TF autodiff
    computes for egt_layers.py:57-143.  Checked against torch.autograd in
    tests/test_oracle.py; it is the formula sheet the HIP backward follows.

    Returns (dQKV, dE, dG); dE/dG are None when the input was None.
    """
    dt = QKV.dtype
    B, N, C = QKV.shape
    H = num_heads
    d = C // (3 * H)
    scale = d ** -0.5
    Q, K, V = QKV.reshape(B, N, 3, d, H).unbind(2)
    A_raw = torch.einsum("bldh,bmdh->blmh", Q, K) * scale
    if clip_logits_value is not None:
        lo, hi = clip_logits_value
        cpred = ((A_raw >= lo) & (A_raw <= hi)).to(dt)
        A_hat = torch.clamp(A_raw, lo, hi)
    else:
        cpred = torch.ones_like(A_raw)
        A_hat = A_raw
    H_hat = A_hat if E is None else A_hat + E
    Hm, Gm = H_hat, G                                         # masks added one by one, as in egt_forward
    for m_ in ([] if mask is None else [(mask[:, None, :, None].to(dt) - 1) * NEG]) + \
              ([] if M is None else [(M.to(dt) - 1) * NEG]) + \
              ([] if rand_mask is None else [torch.where(rand_mask, torch.tensor(-NEG, dtype=dt),
                                                         torch.tensor(0.0, dtype=dt))]):
        Hm = _add_mask(Hm, m_.expand_as(Hm))
        if G is not None:
            Gm = _add_mask(Gm, m_.expand_as(Gm))
    S = torch.softmax(Hm, dim=2)
    if G is not None:
        g = torch.sigmoid(Gm)
    else:
        g = torch.ones_like(S)
    A_tild = S * g
    Dk = torch.ones_like(S)
    if drop_keep is not None and attn_dropout > 0.0:
        Dk = drop_keep.to(dt) / (1.0 - attn_dropout)
    A_drop = A_tild * Dk
    O = torch.einsum("blmh,bmdh->bldh", A_drop, V)            # pre-scaler

    dOut = dV_att.reshape(B, N, d, H)
    ddeg = None
    if scale_degree:
        deg = g.sum(dim=2, keepdim=True)                      # b,l,1,h
        sc = torch.log(1 + deg) if scaler_type == "log" else deg
        live = torch.ones_like(sc)
        if num_virtual_nodes > 0:
            sc = sc.clone()
            sc[:, :num_virtual_nodes] = 1.0
            live[:, :num_virtual_nodes] = 0.0
        dsc = (dOut * O).sum(dim=2, keepdim=True).permute(0, 1, 2, 3)  # b,l,1,h
        dO = dOut * sc
        ddeg = dsc / (1 + deg) if scaler_type == "log" else dsc
        ddeg = ddeg * live
    else:
        dO = dOut

    dA_drop = torch.einsum("bldh,bmdh->blmh", dO, V)
    dV = torch.einsum("blmh,bldh->bmdh", A_drop, dO)
    dA_t = dA_drop * Dk
    dS = dA_t * g
    dg = dA_t * S
    if ddeg is not None:
        dg = dg + ddeg
    dGl = dg * g * (1 - g)
    rowdot = (S * dS).sum(dim=2, keepdim=True)
    dH = S * (dS - rowdot)
    if dH_ext is not None:
        dH = dH + dH_ext
    dA = dH * cpred * scale
    dQ = torch.einsum("blmh,bmdh->bldh", dA, K)
    dK = torch.einsum("blmh,bldh->bmdh", dA, Q)
    dQKV = torch.stack([dQ, dK, dV], dim=2).reshape(B, N, 3 * d * H)
    dE = dH if E is not None else None
    dG = dGl if G is not None else None
    return dQKV, dE, dG


# --------------------------------------------------------------------------
# Keras-default building blocks
# --------------------------------------------------------------------------
def layer_norm(x, gamma, beta, eps=1e-3):
    """keras.layers.LayerNormalization(axis=-1, epsilon=1e-3): moments over the
    last axis (biased variance), then scale/center."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * gamma + beta


def dense(x, W, b):
    """keras.layers.Dense: x @ W + b with W:[in,out]."""
    return x @ W + b


def edge_activation_fn(x, edge_activation):
    """graph_xformer_model_base.py:149-162: None, 'lreluN' (alpha=N/10), or a
    Keras activation name (only 'elu'/'relu' restated)."""
    if edge_activation is None:
        return x
    ea = edge_activation.lower()
    if ea.startswith("lrelu"):
        alpha = float(ea[-1]) / 10
        return torch.where(x >= 0, x, alpha * x)
    if ea == "elu":
        return torch.nn.functional.elu(x)
    if ea == "relu":
        return torch.relu(x)
    raise ValueError(f"unsupported edge_activation {edge_activation}")


BLOCK_PARAM_NAMES = (
    # Keras layer name stems (graph_xformer_model_base.py:109-218); XX = layer tag
    "norm_edge.gamma", "norm_edge.beta",            # :195
    "attention_gates.kernel", "attention_gates.bias",  # :201-204  De->H
    "dense_edge_b.kernel", "dense_edge_b.bias",     # :159-161  De->H
    "norm_mha.gamma", "norm_mha.beta",              # :109
    "dense_qkv.kernel", "dense_qkv.bias",           # :113      Dh->3Dh
    "dense_mha.kernel", "dense_mha.bias",           # :136      Dh->Dh
    "dense_edge_r.kernel", "dense_edge_r.bias",     # :214      H->De
)


def init_block_params(Dh, De, H, *, dtype=torch.float32, generator=None,
                      randomize_norm=False):
    """Keras defaults: Dense kernel Glorot-uniform, bias 0, LN gamma 1 beta 0.
    randomize_norm perturbs LN/bias params so parity tests exercise them."""
    def glorot(fi, fo):
        lim = math.sqrt(6.0 / (fi + fo))
        return (torch.rand(fi, fo, generator=generator, dtype=torch.float64) * 2 - 1) * lim

    def vec(n, base):
        if randomize_norm:
            return base + 0.2 * torch.randn(n, generator=generator, dtype=torch.float64)
        return torch.full((n,), float(base), dtype=torch.float64)

    p = {
        "norm_edge.gamma": vec(De, 1.0), "norm_edge.beta": vec(De, 0.0),
        "attention_gates.kernel": glorot(De, H), "attention_gates.bias": vec(H, 0.0),
        "dense_edge_b.kernel": glorot(De, H), "dense_edge_b.bias": vec(H, 0.0),
        "norm_mha.gamma": vec(Dh, 1.0), "norm_mha.beta": vec(Dh, 0.0),
        "dense_qkv.kernel": glorot(Dh, 3 * Dh), "dense_qkv.bias": vec(3 * Dh, 0.0),
        "dense_mha.kernel": glorot(Dh, Dh), "dense_mha.bias": vec(Dh, 0.0),
        "dense_edge_r.kernel": glorot(H, De), "dense_edge_r.bias": vec(De, 0.0),
    }
    return {k: v.to(dtype) for k, v in p.items()}


# --------------------------------------------------------------------------
# outer op: edge_update_{residual,bias,none} + mha_block
# --------------------------------------------------------------------------
def block_forward(h, e, mask, params, *, num_heads=8,
                  edge_channel_type="residual", gate_attention=True,
                  clip_logits_value=(-5.0, 5.0), scale_degree=False,
                  scaler_type="log", num_virtual_nodes=0, edge_activation=None,
                  attn_mask=None, rand_mask=None, drop_keep=None,
                  attn_dropout=0.0, add_n_norm=False, return_inner=False,
                  node_keep=None, node_dropout=0.0, edge_keep=None, edge_dropout=0.0):
    """(h, e, mask) -> (h', e'): graph_xformer_model_base.py:192-223 (residual /
    constrained), :173-190 (bias), :164-171 (none), with mha_block :106-145.
    Node/edge dropout (drp_mha :138-139, drp_edge :216-217; rate 0 in every shipped
    config, scheme_base.py:22) take an INJECTED keep mask (Keras Dropout = inverted
    scaling 1/(1-rate) on the kept elements); identity when no mask is given."""
    p = params
    H = num_heads

    def drop(x, keep, rate):
        if keep is None or rate <= 0.0:
            return x
        return x * keep.to(x.dtype) / (1.0 - rate)

    def mha_block(h, e_b, gates):
        y = h                                                       # :107
        hn = h
        if not add_n_norm:                                          # :108-109
            hn = layer_norm(h, p["norm_mha.gamma"], p["norm_mha.beta"])
        qkv = dense(hn, p["dense_qkv.kernel"], p["dense_qkv.bias"])  # :113
        v_att, h_hat, a_tild = egt_forward(
            qkv, e_b, gates, attn_mask, mask, num_heads=H,
            clip_logits_value=clip_logits_value, scale_degree=scale_degree,
            scaler_type=scaler_type, num_virtual_nodes=num_virtual_nodes,
            rand_mask=rand_mask, drop_keep=drop_keep, attn_dropout=attn_dropout)
        ho = dense(v_att, p["dense_mha.kernel"], p["dense_mha.bias"])  # :136
        ho = drop(ho, node_keep, node_dropout)                      # :138-139
        ho = ho + y                                                 # :140
        if add_n_norm:                                              # :142-143
            ho = layer_norm(ho, p["norm_mha.gamma"], p["norm_mha.beta"])
        return ho, h_hat, a_tild

    if edge_channel_type == "none":                                 # :164-171
        h2, h_hat, a_tild = mha_block(h, None, None)
        out = (h2, e)
    elif edge_channel_type == "bias":                               # :173-190
        gates = None
        if gate_attention:
            gates = dense(e, p["attention_gates.kernel"], p["attention_gates.bias"])
        e_b = edge_activation_fn(
            dense(e, p["dense_edge_b.kernel"], p["dense_edge_b.bias"]), edge_activation)
        h2, h_hat, a_tild = mha_block(h, e_b, gates)
        out = (h2, e)
    elif edge_channel_type in ("residual", "constrained"):          # :192-223
        y = e
        en = e
        if not add_n_norm:
            en = layer_norm(e, p["norm_edge.gamma"], p["norm_edge.beta"])
        gates = None
        if gate_attention:
            gates = dense(en, p["attention_gates.kernel"], p["attention_gates.bias"])
        e_b = edge_activation_fn(
            dense(en, p["dense_edge_b.kernel"], p["dense_edge_b.bias"]), edge_activation)
        h2, h_hat, a_tild = mha_block(h, e_b, gates)
        e2 = dense(h_hat, p["dense_edge_r.kernel"], p["dense_edge_r.bias"])  # :214
        e2 = drop(e2, edge_keep, edge_dropout)                      # :216-217
        e2 = e2 + y                                                 # :218
        if add_n_norm:
            e2 = layer_norm(e2, p["norm_edge.gamma"], p["norm_edge.beta"])
        out = (h2, e2)
    else:
        raise KeyError(edge_channel_type)
    if return_inner:
        return out + (h_hat, a_tild)
    return out


def ffn_forward(x, p, activation="elu"):
    """One channel type of ffn_block: graph_xformer_model_base.py:230-258 (ffnlr1 :230-239,
    ffnact :241-246 -- identity without cross-talk, the activation then sits in fnn_lr1 --,
    ffnlr2 :248-258), pre-norm (add_n_norm False), dropout 0, as called by :309-324.
    p: norm_gamma, norm_beta, lr1_kernel [W,2W], lr1_bias, lr2_kernel [2W,W], lr2_bias."""
    y = x
    xn = layer_norm(x, p["norm_gamma"], p["norm_beta"])
    hid = dense(xn, p["lr1_kernel"], p["lr1_bias"])
    if activation == "elu":
        hid = torch.nn.functional.elu(hid)
    elif activation == "relu":
        hid = torch.relu(hid)
    elif activation is not None:
        raise ValueError(activation)
    return dense(hid, p["lr2_kernel"], p["lr2_bias"]) + y


def stack_forward(h, e, mask, layer_params, **kw):
    """The Ly-layer attention stack of graph_xformer_model_base.py:336-339
    (edge_update only; the ffn_block at :340-341 is outside this path)."""
    rand_masks = kw.pop("rand_masks", None)
    for i, p in enumerate(layer_params):
        rm = None if rand_masks is None else rand_masks[i]
        h, e = block_forward(h, e, mask, p, rand_mask=rm, **kw)
    return h, e


# --------------------------------------------------------------------------
# mask producers
# --------------------------------------------------------------------------
def node_mask_from_features(node_features: torch.Tensor, num_virtual_nodes: int = 0) -> torch.Tensor:
    """Neg1MaskedEmbedding.compute_mask (masking.py:35-43): Embedding(mask_zero=True) on
    inputs+1, i.e. (x+1) != 0.  VirtualNodeEmbedding.compute_mask (virtual_nodes.py:47-50)
    prepends num_virtual_nodes True entries."""
    m = (node_features + 1) != 0
    if num_virtual_nodes > 0:
        m = torch.cat([torch.ones(m.shape[0], num_virtual_nodes, dtype=torch.bool), m], dim=1)
    return m


def node_mask_from_masking(node_features: torch.Tensor, mask_value: float = -1.0,
                           num_virtual_nodes: int = 0) -> torch.Tensor:
    """keras.layers.Masking(mask_value) (cifar10/dc.py:69): a step is kept when any of its
    features differs from mask_value."""
    m = (node_features != mask_value).any(dim=-1)
    if num_virtual_nodes > 0:
        m = torch.cat([torch.ones(m.shape[0], num_virtual_nodes, dtype=torch.bool), m], dim=1)
    return m


def constrained_edge_mask(adj: torch.Tensor, num_heads: int, num_virtual_nodes: int = 0) -> torch.Tensor:
    """AdjMatModel.get_edge_mask (graph_model_base.py:131-142): tile adjacency over heads;
    VNModel.get_edge_mask (:248-268): concat ones rows (axis 1), then ones columns (axis 2)."""
    M = adj[..., None].repeat(1, 1, 1, num_heads)
    nv = num_virtual_nodes
    if nv > 0:
        B, N1, N2, H = M.shape
        M = torch.cat([torch.ones(B, nv, N2, H, dtype=M.dtype), M], dim=1)
        M = torch.cat([torch.ones(B, N1 + nv, nv, H, dtype=M.dtype), M], dim=2)
    return M
