"""Seeded parity cases shared by the golden-fixture generator and the tests.

Inputs are fp32 (what the HIP path sees); the oracle evaluates them in fp64.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from oracle import egt_oracle as O  # noqa: E402

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# name -> dict(shape + operator attributes + input recipe)
ATTN_CASES = {
    # ZINC-500K head geometry (d=8), ragged key padding, logits pushed past the clip
    "gated_d8_clip": dict(B=2, N=5, H=8, d=8, qscale=3.0, nodes=[5, 3]),
    # ZINC-100K geometry: d = 48/8 = 6 (not a power of two), N=37 (not a multiple of 8)
    "gated_d6_n37": dict(B=1, N=37, H=8, d=6, nodes=[29]),
    # padded N=64 tile-aligned
    "gated_n64": dict(B=2, N=64, H=8, d=8, nodes=[37, 9]),
    # a graph whose every key is masked (all-masked rows: softmax uniform, gates 0)
    "gated_allmasked": dict(B=2, N=6, H=8, d=8, nodes=[6, 0]),
    # 'constrained': attention mask M with empty rows (graph_model_base.py:131-142)
    "constrained": dict(B=2, N=7, H=8, d=8, nodes=[7, 4], attn_mask=True, m_zero_rows=True),
    # ungated (gate_attention:false ablation / edge_update_none)
    "ungated": dict(B=2, N=9, H=8, d=8, nodes=[9, 5], gate=False),
    "ungated_noedge": dict(B=1, N=8, H=8, d=8, nodes=[6], gate=False, edge=False),
    # degree scalers with virtual nodes
    "scale_log_vn2": dict(B=2, N=7, H=8, d=8, nodes=[7, 5], scale_degree=True, nvn=2),
    "scale_linear": dict(B=1, N=6, H=8, d=8, nodes=[4], scale_degree=True, scaler_type="linear"),
    # stochastic branches with injected samples
    "randmask_dropout": dict(B=2, N=8, H=8, d=8, nodes=[8, 6], rand_p=0.3, drop_p=0.2),
    # no clip, no key mask, other head counts
    "noclip_nomask_h4": dict(B=1, N=10, H=4, d=16, nodes=None, clip=None),
    # stress geometry (config 5 scaled down): d=64
    "d64": dict(B=1, N=24, H=8, d=64, nodes=[20]),
}

BLOCK_CASES = {
    "residual_zinc500k": dict(B=2, N=9, Dh=64, De=64, nodes=[9, 6]),
    "residual_zinc100k": dict(B=2, N=11, Dh=48, De=48, nodes=[11, 7]),          # d=6
    "residual_pattern": dict(B=1, N=20, Dh=64, De=8, nodes=[17]),               # De=8
    "residual_randmask": dict(B=2, N=16, Dh=64, De=64, nodes=[16, 11], rand_p=0.25),
    "constrained": dict(B=2, N=8, Dh=64, De=32, nodes=[8, 5], ect="constrained"),
    "bias": dict(B=2, N=7, Dh=64, De=16, nodes=[7, 4], ect="bias"),
    "bias_lrelu": dict(B=1, N=6, Dh=64, De=16, nodes=[5], ect="bias", edge_activation="lrelu2"),
    "none": dict(B=2, N=6, Dh=64, De=8, nodes=[6, 3], ect="none"),
    "ungated_residual": dict(B=1, N=8, Dh=64, De=64, nodes=[6], gate=False),
    "residual_n64": dict(B=2, N=64, Dh=64, De=64, nodes=[37, 20]),
}


def _key_mask(B, N, nodes):
    if nodes is None:
        return None
    m = torch.zeros(B, N, dtype=torch.bool)
    for b, n in enumerate(nodes):
        m[b, :n] = True
    return m


def make_attn_case(name, seed=1234):
    c = dict(ATTN_CASES[name])
    g = torch.Generator().manual_seed(seed + sum(map(ord, name)))
    B, N, H, d = c["B"], c["N"], c["H"], c["d"]
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)
    inp = dict(QKV=r(B, N, 3 * d * H) * c.get("qscale", 1.0))
    inp["E"] = r(B, N, N, H) if c.get("edge", True) else None
    inp["G"] = r(B, N, N, H) if c.get("gate", True) else None
    inp["mask"] = _key_mask(B, N, c.get("nodes"))
    M = None
    if c.get("attn_mask"):
        M = (torch.rand(B, N, N, generator=g) > 0.4).to(torch.float32)
        if c.get("m_zero_rows"):
            M[:, 1, :] = 0.0
        M = M[..., None].repeat(1, 1, 1, H).contiguous()
    inp["M"] = M
    inp["rand_mask"] = (torch.rand(B, N, N, H, generator=g) < c["rand_p"]) if c.get("rand_p") else None
    inp["drop_keep"] = (torch.rand(B, N, N, H, generator=g) >= c["drop_p"]) if c.get("drop_p") else None
    inp["dV"] = r(B, N, d * H)
    inp["dH"] = r(B, N, N, H)
    attrs = dict(num_heads=H, clip_logits_value=c.get("clip", (-5.0, 5.0)),
                 scale_degree=c.get("scale_degree", False), scaler_type=c.get("scaler_type", "log"),
                 num_virtual_nodes=c.get("nvn", 0), attn_dropout=c.get("drop_p", 0.0))
    return inp, attrs, c


def attn_oracle(inp, attrs, dtype=torch.float64):
    cv = lambda t: None if t is None else t.to(dtype)
    QKV = cv(inp["QKV"]).requires_grad_()
    E = cv(inp["E"]); G = cv(inp["G"])
    if E is not None:
        E.requires_grad_()
    if G is not None:
        G.requires_grad_()
    V, Hh, At = O.egt_forward(QKV, E, G, cv(inp["M"]), inp["mask"], rand_mask=inp["rand_mask"],
                              drop_keep=inp["drop_keep"], **attrs)
    wrt = [t for t in (QKV, E, G) if t is not None]
    grads = torch.autograd.grad((V * cv(inp["dV"])).sum() + (Hh * cv(inp["dH"])).sum(), wrt)
    gi = iter(grads)
    out = dict(V_att=V.detach(), H_hat=Hh.detach(), A_tild=At.detach(), dQKV=next(gi))
    out["dE"] = next(gi) if E is not None else None
    out["dG"] = next(gi) if G is not None else None
    return out


def make_block_case(name, seed=4321):
    c = dict(BLOCK_CASES[name])
    g = torch.Generator().manual_seed(seed + sum(map(ord, name)))
    B, N, Dh, De, H = c["B"], c["N"], c["Dh"], c["De"], 8
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float32)
    ect = c.get("ect", "residual")
    inp = dict(h=r(B, N, Dh), e=r(B, N, N, De) * 1.5 + 0.3, mask=_key_mask(B, N, c.get("nodes")))
    inp["attn_mask"] = None
    if ect == "constrained":
        adj = (torch.rand(B, N, N, generator=g) > 0.5).to(torch.float32)
        inp["attn_mask"] = O.constrained_edge_mask(adj, H).contiguous()
    inp["rand_mask"] = (torch.rand(B, N, N, H, generator=g) < c["rand_p"]) if c.get("rand_p") else None
    inp["dh"] = r(B, N, Dh)
    inp["de"] = r(B, N, N, De)
    params = O.init_block_params(Dh, De, H, dtype=torch.float32, generator=g, randomize_norm=True)
    attrs = dict(num_heads=H, edge_channel_type=ect, gate_attention=c.get("gate", True),
                 edge_activation=c.get("edge_activation"))
    return inp, params, attrs, c


def block_oracle(inp, params, attrs, dtype=torch.float64):
    cv = lambda t: None if t is None else t.to(dtype)
    h = cv(inp["h"]).requires_grad_()
    e = cv(inp["e"]).requires_grad_()
    p = {k: v.to(dtype).requires_grad_() for k, v in params.items()}
    h2, e2 = O.block_forward(h, e, inp["mask"], p, attn_mask=cv(inp["attn_mask"]),
                             rand_mask=inp["rand_mask"], **attrs)
    loss = (h2 * cv(inp["dh"])).sum() + (e2 * cv(inp["de"])).sum()
    names = list(p.keys())
    grads = torch.autograd.grad(loss, [h, e] + [p[k] for k in names], allow_unused=True)
    out = dict(h_out=h2.detach(), e_out=e2.detach(), dh=grads[0], de=grads[1])
    out["dparams"] = {k: g for k, g in zip(names, grads[2:])}
    return out


def to_np(t):
    if t is None:
        return None
    return t.detach().cpu().numpy()


def save_npz(path, tree):
    flat = {}
    for k, v in tree.items():
        if isinstance(v, dict):
            for k2, v2 in v.items():
                if v2 is not None:
                    flat[f"{k}/{k2}"] = to_np(v2)
        elif v is not None:
            flat[k] = to_np(v)
    np.savez_compressed(path, **flat)


# ----------------------------------------------------------------- channel FFN -----
FFN_NAMES = ("norm_gamma", "norm_beta", "lr1_kernel", "lr1_bias", "lr2_kernel", "lr2_bias")
FFN_CASES = {
    "edge_w64_elu": dict(shape=(2, 12, 12), W=64, act="elu"),
    "node_w48_relu": dict(shape=(3, 11), W=48, act="relu"),
    "edge_w16_elu": dict(shape=(2, 9, 9), W=16, act="elu"),
}


def make_ffn_case(name, seed=777):
    c = FFN_CASES[name]
    g = torch.Generator().manual_seed(seed + c["W"])
    W, H = c["W"], 2 * c["W"]
    lim = (6.0 / (W + H)) ** 0.5
    params = {
        "norm_gamma": 1.0 + 0.3 * torch.randn(W, generator=g), "norm_beta": 0.3 * torch.randn(W, generator=g),
        "lr1_kernel": (torch.rand(W, H, generator=g) * 2 - 1) * lim, "lr1_bias": 0.2 * torch.randn(H, generator=g),
        "lr2_kernel": (torch.rand(H, W, generator=g) * 2 - 1) * lim, "lr2_bias": 0.2 * torch.randn(W, generator=g),
    }
    inp = {"x": torch.randn(*c["shape"], W, generator=g) * 1.5 + 0.2, "dy": torch.randn(*c["shape"], W, generator=g)}
    return inp, params, c


def ffn_oracle(inp, params, c, dtype=torch.float64):
    from oracle import egt_oracle as O
    x = inp["x"].to(dtype).requires_grad_()
    p = {k: v.to(dtype).requires_grad_() for k, v in params.items()}
    y = O.ffn_forward(x, p, activation=c["act"])
    gr = torch.autograd.grad(y, [x] + [p[k] for k in FFN_NAMES], inp["dy"].to(dtype))
    return {"y": y.detach(), "dx": gr[0], "dparams": dict(zip(FFN_NAMES, gr[1:]))}


# ----------------------------------------------------------------- whole ZINC model -----
MODEL_CASES = {
    # name: model config (reference model_config keys) + batch geometry
    "zinc_small": dict(cfg=dict(model_width=16, edge_width=16, model_height=2, upto_hop=4), B=3, N=11, nodes=(4, 11)),
    "zinc_w64": dict(cfg=dict(model_width=64, edge_width=64, model_height=2, upto_hop=16), B=2, N=21, nodes=(9, 21)),
}


def make_model_case(name, seed=99):
    from oracle import egt_model_oracle as MO
    c = MODEL_CASES[name]
    g = torch.Generator().manual_seed(seed + sum(map(ord, name)))
    B, N = c["B"], c["N"]
    lo, hi = c["nodes"]
    n = torch.randint(lo, hi + 1, (B,), generator=g); n[0] = N
    real = torch.arange(N)[None, :] < n[:, None]
    nf = torch.randint(0, 28, (B, N), generator=g)
    nf[~real] = -1
    adj = (torch.rand(B, N, N, generator=g) > 0.7).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float() * (real[:, :, None] & real[:, None, :]).float()
    adj = adj * (1 - torch.eye(N))[None]
    fm = torch.where(adj > 0, torch.randint(0, 4, (B, N, N), generator=g), torch.tensor(-1))
    fm = torch.minimum(fm, fm.transpose(1, 2)) if False else fm
    y = torch.randn(B, 1, generator=g)
    params = MO.init_zinc_params(c["cfg"], dtype=torch.float32, generator=g)
    return dict(node_features=nf, feature_matrix=fm, graph_matrix=adj, target=y), params, c


def model_oracle(inp, params, c, dtype=torch.float64, rand_masks=None):
    from oracle import egt_model_oracle as MO
    p = {k: v.to(dtype).requires_grad_() for k, v in params.items()}
    y = MO.zinc_forward(inp["node_features"], inp["feature_matrix"], inp["graph_matrix"], p, c["cfg"], rand_masks=rand_masks)
    loss = MO.mae_loss(y, inp["target"].to(dtype))
    names = list(p.keys())
    grads = torch.autograd.grad(loss, [p[k] for k in names], allow_unused=True)
    return dict(y=y.detach(), loss=loss.detach().reshape(1)), {k: g for k, g in zip(names, grads)}
