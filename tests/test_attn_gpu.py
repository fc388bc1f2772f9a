"""GPU parity: inner op EGT([QKV,E,G,M],mask) fwd+bwd through the C-ABI vs the
fp64 oracle and the committed golden fixtures."""
import os

import numpy as np
import pytest
import torch

import cases as CS
from util import assert_close, load_golden, FWD, BWD

pytestmark = pytest.mark.gpu


def run_hip(inp, attrs, dev, a_tild=True):
    from egt_amd import egt_attention, AttnConfig
    cu = lambda t: None if t is None else t.to(dev)
    QKV = cu(inp["QKV"]).requires_grad_()
    E = cu(inp["E"]); G = cu(inp["G"])
    if E is not None:
        E.requires_grad_()
    if G is not None:
        G.requires_grad_()
    stochastic = inp["rand_mask"] is not None or inp["drop_keep"] is not None
    cfg = AttnConfig(num_heads=attrs["num_heads"], clip_logits_value=attrs["clip_logits_value"],
                     scale_degree=attrs["scale_degree"], scaler_type=attrs["scaler_type"],
                     num_virtual_nodes=attrs["num_virtual_nodes"],
                     random_mask_prob=0.5 if inp["rand_mask"] is not None else 0.0,
                     attn_dropout=attrs["attn_dropout"], training=stochastic, seed=7,
                     need_a_tild=a_tild)
    V, Hh, At = egt_attention(QKV, E, G, cu(inp["M"]), cu(inp["mask"]), cfg=cfg,
                              rand_mask=cu(inp["rand_mask"]), drop_keep=cu(inp["drop_keep"]))
    wrt = [t for t in (QKV, E, G) if t is not None]
    grads = torch.autograd.grad((V * cu(inp["dV"])).sum() + (Hh * cu(inp["dH"])).sum(), wrt)
    gi = iter(grads)
    out = dict(V_att=V.detach(), H_hat=Hh.detach(), A_tild=At.detach(), dQKV=next(gi))
    out["dE"] = next(gi) if E is not None else None
    out["dG"] = next(gi) if G is not None else None
    return out


def compare(out, ref):
    for k in ("V_att", "H_hat", "A_tild"):
        assert_close(out[k], ref[k], name=k, **FWD)
    for k in ("dQKV", "dE", "dG"):
        if ref.get(k) is not None:
            assert_close(out[k], ref[k], name=k, **BWD)


@pytest.mark.parametrize("name", list(CS.ATTN_CASES))
def test_attn_vs_oracle(name, gpu, egt_lib):
    inp, attrs, _ = CS.make_attn_case(name)
    compare(run_hip(inp, attrs, gpu), CS.attn_oracle(inp, attrs))


@pytest.mark.parametrize("name", [n for n in CS.ATTN_CASES if n != "gated_n64"])
def test_attn_vs_golden(name, gpu, egt_lib):
    g = load_golden(os.path.join(CS.GOLDEN_DIR, f"attn_{name}.npz"))
    inp, attrs, _ = CS.make_attn_case(name)
    for k, v in g["in"].items():
        assert np.array_equal(CS.to_np(inp[k]), v)
    out = run_hip(inp, attrs, gpu)
    compare(out, {k: torch.from_numpy(v) for k, v in g["out"].items()})


def test_masked_positions_bit_exact_zero(gpu, egt_lib):
    inp, attrs, _ = CS.make_attn_case("gated_d8_clip")
    out = run_hip(inp, attrs, gpu)
    pad = ~inp["mask"]
    for b in range(pad.shape[0]):
        assert (out["A_tild"][b][:, pad[b].to(gpu), :] == 0).all()
        assert (out["dG"][b][:, pad[b].to(gpu), :] == 0).all()
    inp, attrs, _ = CS.make_attn_case("gated_allmasked")
    out = run_hip(inp, attrs, gpu)
    assert (out["A_tild"][1] == 0).all() and (out["V_att"][1] == 0).all()


def test_a_tild_optional(gpu, egt_lib):
    inp, attrs, _ = CS.make_attn_case("gated_d8_clip")
    a = run_hip(inp, attrs, gpu, a_tild=True)
    b = run_hip(inp, attrs, gpu, a_tild=False)
    assert b["A_tild"].numel() == 0
    assert torch.equal(a["V_att"], b["V_att"]) and torch.equal(a["dQKV"], b["dQKV"])


def test_layer_call_convention(gpu, egt_lib):
    """EGT(...)([QKV,E,G,M], mask) with list-wrapped mask (egt_layers.py:62-66)."""
    from egt_amd import EGT
    inp, attrs, _ = CS.make_attn_case("constrained")
    layer = EGT(num_heads=8, attn_mask=True, name="mha_00").to(gpu).eval()
    V, Hh, At = layer([inp["QKV"].to(gpu), inp["E"].to(gpu), inp["G"].to(gpu), inp["M"].to(gpu)],
                      mask=[inp["mask"].to(gpu)])
    ref = CS.attn_oracle(inp, attrs)
    assert_close(V, ref["V_att"], name="V_att", **FWD)
    assert_close(At, ref["A_tild"], name="A_tild", **FWD)
    assert layer.compute_mask(None, [inp["mask"]])[1:] == [None, None]
    with pytest.raises(AssertionError):     # egt_layers.py:70
        layer([torch.zeros(1, 4, 50, device=gpu)] + [torch.zeros(1, 4, 4, 8, device=gpu)] * 3)


def test_device_rng_stream_bit_exact(gpu, egt_lib):
    from egt_amd import mask_sample
    from oracle import rng_ref
    B, N, H = 3, 17, 8
    for seed, p in ((0, 0.1), (0x1234567890ABCDEF, 0.1), (42, 0.37)):
        dev = mask_sample(0, seed, p, B, N, H, gpu).cpu().numpy().astype(bool)
        assert np.array_equal(dev, rng_ref.random_mask(seed, B, N, H, p))
        devk = mask_sample(1, seed, p, B, N, H, gpu).cpu().numpy().astype(bool)
        assert np.array_equal(devk, rng_ref.dropout_keep(seed, B, N, H, p))


def test_in_kernel_random_mask_matches_injected(gpu, egt_lib):
    """The in-kernel sample (seeded counter hash) gives the same result as
    injecting the oracle-side replica of that sample."""
    from egt_amd import egt_attention, AttnConfig
    from oracle import rng_ref
    inp, attrs, _ = CS.make_attn_case("gated_n64")
    B, N, H = 2, 64, 8
    seed, p, pd = 99, 0.1, 0.15
    cfg = AttnConfig(random_mask_prob=p, attn_dropout=pd, training=True, seed=seed, need_a_tild=True)
    cu = lambda t: t.to(gpu)
    V, Hh, At = egt_attention(cu(inp["QKV"]), cu(inp["E"]), cu(inp["G"]), None, cu(inp["mask"]), cfg=cfg)
    inp2 = dict(inp)
    inp2["rand_mask"] = torch.from_numpy(rng_ref.random_mask(seed, B, N, H, p))
    inp2["drop_keep"] = torch.from_numpy(rng_ref.dropout_keep(seed, B, N, H, pd))
    attrs2 = dict(attrs, attn_dropout=pd)
    ref = CS.attn_oracle(inp2, attrs2)
    assert_close(V, ref["V_att"], name="V_att", **FWD)
    assert_close(At, ref["A_tild"], name="A_tild", **FWD)
    frac = inp2["rand_mask"].float().mean().item()
    assert abs(frac - p) < 0.01
    # eval mode: no mask
    cfg_eval = AttnConfig(random_mask_prob=p, training=False, need_a_tild=True)
    V2, _, _ = egt_attention(cu(inp["QKV"]), cu(inp["E"]), cu(inp["G"]), None, cu(inp["mask"]), cfg=cfg_eval)
    assert_close(V2, CS.attn_oracle(inp, attrs)["V_att"], name="V_eval", **FWD)


@pytest.mark.parametrize("B,N,d,opts", [
    (1, 24, 64, {}), (2, 37, 16, {}), (1, 64, 32, dict(attn_mask=True)), (1, 40, 64, dict(gate=False)),
    (2, 16, 16, dict(rand_p=0.3)), (1, 33, 64, dict(clip=None, nomask=True)), (1, 128, 64, {}),
])
def test_attn_mfma_kernel_vs_oracle(B, N, d, opts, gpu, egt_lib):
    """MFMA-tiled inner op (d in {16,32,64}) forward + general backward vs the fp64 oracle, and
    bit-level agreement of its H_hat with the general kernel's."""
    from egt_amd import egt_attention, AttnConfig
    H = 8
    g = torch.Generator().manual_seed(B * 100 + N + d)
    QKV = torch.randn(B, N, 3 * d * H, generator=g) * 0.7
    E = torch.randn(B, N, N, H, generator=g)
    G = torch.randn(B, N, N, H, generator=g) if opts.get("gate", True) else None
    mask = None
    if not opts.get("nomask"):
        mask = torch.ones(B, N, dtype=torch.bool); mask[0, N - 5:] = False
    M = None
    if opts.get("attn_mask"):
        M = (torch.rand(B, N, N, generator=g) > 0.4).float()[..., None].repeat(1, 1, 1, H).contiguous()
    rm = (torch.rand(B, N, N, H, generator=g) < opts["rand_p"]) if opts.get("rand_p") else None
    dV = torch.randn(B, N, d * H, generator=g); dH = torch.randn(B, N, N, H, generator=g)
    inp = dict(QKV=QKV, E=E, G=G, M=M, mask=mask, rand_mask=rm, drop_keep=None, dV=dV, dH=dH)
    attrs = dict(num_heads=H, clip_logits_value=opts.get("clip", (-5.0, 5.0)), scale_degree=False,
                 scaler_type="log", num_virtual_nodes=0, attn_dropout=0.0)
    ref = CS.attn_oracle(inp, attrs)
    cu = lambda t: None if t is None else t.to(gpu)
    outs = {}
    for use_mfma in (True, False):
        q = cu(QKV).requires_grad_(); e_ = cu(E).requires_grad_()
        g_ = cu(G).requires_grad_() if G is not None else None
        cfg = AttnConfig(num_heads=H, clip_logits_value=attrs["clip_logits_value"],
                         random_mask_prob=0.5 if rm is not None else 0.0, training=rm is not None,
                         need_a_tild=False, use_mfma=use_mfma)
        V, Hh, _ = egt_attention(q, e_, g_, cu(M), cu(mask), cfg=cfg, rand_mask=cu(rm))
        torch.autograd.backward([V, Hh], [cu(dV), cu(dH)])
        outs[use_mfma] = (V.detach(), Hh.detach(), q.grad, e_.grad, None if g_ is None else g_.grad)
    V, Hh, dq, de_, dg = outs[True]
    assert_close(V, ref["V_att"], name="V_att", **FWD)
    assert_close(Hh, ref["H_hat"], name="H_hat", **FWD)
    assert_close(dq, ref["dQKV"], name="dQKV", **BWD)
    assert_close(de_, ref["dE"], name="dE", **BWD)
    if dg is not None:
        assert_close(dg, ref["dG"], name="dG", **BWD)
    assert_close(outs[True][0], outs[False][0], name="V_att(mfma vs general)", rtol=1e-4, arel=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("N,d", [(48, 64), (37, 16)])
def test_attn_mfma_in_kernel_random_mask(N, d, gpu, egt_lib):
    """The straight-line MFMA instances with the in-kernel random-mask stream (training, no injected
    bytes): forward + backward equal the fp64 oracle fed the same mask, materialised by mask_sample."""
    from egt_amd import egt_attention, mask_sample, AttnConfig
    B, H, seed, p = 2, 8, 4242, 0.25
    g = torch.Generator().manual_seed(N + d)
    QKV = torch.randn(B, N, 3 * d * H, generator=g) * 0.7
    E = torch.randn(B, N, N, H, generator=g); G = torch.randn(B, N, N, H, generator=g)
    mask = torch.ones(B, N, dtype=torch.bool); mask[1, N - 7:] = False
    dV = torch.randn(B, N, d * H, generator=g); dH = torch.randn(B, N, N, H, generator=g)
    rm = mask_sample(0, seed, p, B, N, H, gpu).cpu().bool()
    inp = dict(QKV=QKV, E=E, G=G, M=None, mask=mask, rand_mask=rm, drop_keep=None, dV=dV, dH=dH)
    attrs = dict(num_heads=H, clip_logits_value=(-5.0, 5.0), scale_degree=False, scaler_type="log",
                 num_virtual_nodes=0, attn_dropout=0.0)
    ref = CS.attn_oracle(inp, attrs)
    q = QKV.to(gpu).requires_grad_(); e_ = E.to(gpu).requires_grad_(); g_ = G.to(gpu).requires_grad_()
    cfg = AttnConfig(num_heads=H, random_mask_prob=p, training=True, seed=seed, need_a_tild=False, use_mfma=True)
    V, Hh, _ = egt_attention(q, e_, g_, None, mask.to(gpu), cfg=cfg)
    torch.autograd.backward([V, Hh], [dV.to(gpu), dH.to(gpu)])
    assert_close(V, ref["V_att"], name="V_att", **FWD)
    assert_close(Hh, ref["H_hat"], name="H_hat", **FWD)
    assert_close(q.grad, ref["dQKV"], name="dQKV", **BWD)
    assert_close(e_.grad, ref["dE"], name="dE", **BWD)
    assert_close(g_.grad, ref["dG"], name="dG", **BWD)


@pytest.mark.gpu
@pytest.mark.parametrize("N,d", [(48, 64), (37, 16)])
def test_attn_mfma_shared_workspace_fwd_to_bwd(N, d, gpu, egt_lib, monkeypatch):
    """EGT_ATTN_WS_SHARED: the forward packs the q / k / v operand copies of BOTH directions into one workspace and the backward
    adds only dO.  The autograd front-end takes that path whenever an input needs a gradient (ctx.needs_input_grad); the
    workspace travels with the node's SAVED tensors (freed with the graph, reused by a second backward under retain_graph);
    with the opt-out EGT_ATTN_WS_SHARED=0 nothing is kept and the backward re-packs: all three must give the same gradients,
    bit for bit, and the C-ABI pair called by hand with the two workspace modes likewise."""
    import ctypes as C
    from egt_amd import egt_attention, AttnConfig, _lib as L
    B, H = 2, 8
    g = torch.Generator().manual_seed(N * 3 + d)
    QKV = (torch.randn(B, N, 3 * d * H, generator=g) * 0.7).to(gpu)
    E = torch.randn(B, N, N, H, generator=g).to(gpu); G = torch.randn(B, N, N, H, generator=g).to(gpu)
    mask = torch.ones(B, N, dtype=torch.bool); mask[1, N - 7:] = False
    dV = torch.randn(B, N, d * H, generator=g).to(gpu); dH = torch.randn(B, N, N, H, generator=g).to(gpu)
    cfg = AttnConfig(num_heads=H, need_a_tild=False, use_mfma=True)
    q = QKV.clone().requires_grad_(); e_ = E.clone().requires_grad_(); g_ = G.clone().requires_grad_()
    V, Hh, _ = egt_attention(q, e_, g_, None, mask.to(gpu), cfg=cfg)
    ws_saved = V.grad_fn.saved_tensors[-1]
    assert ws_saved is not None and ws_saved.dtype == torch.uint8, "a forward that will be differentiated must take the shared workspace"
    torch.autograd.backward([V, Hh], [dV, dH], retain_graph=True)
    first = (q.grad.clone(), e_.grad.clone(), g_.grad.clone())
    q.grad = e_.grad = g_.grad = None
    torch.autograd.backward([V, Hh], [dV, dH])                       # second pass through the retained node: same workspace again
    for a, b, n in zip(first, (q.grad, e_.grad, g_.grad), ("dQKV", "dE", "dG")):
        assert torch.equal(a, b), n
    monkeypatch.setenv("EGT_ATTN_WS_SHARED", "0")                    # the opt-out: nothing kept, the backward re-packs
    q2 = QKV.clone().requires_grad_(); e2 = E.clone().requires_grad_(); g2 = G.clone().requires_grad_()
    Vn, Hn, _ = egt_attention(q2, e2, g2, None, mask.to(gpu), cfg=cfg)
    assert Vn.grad_fn.saved_tensors[-1] is None
    torch.autograd.backward([Vn, Hn], [dV, dH])
    for a, b, n in zip(first, (q2.grad, e2.grad, g2.grad), ("dQKV", "dE", "dG")):
        assert torch.equal(a, b), n + " (re-packing backward)"
    monkeypatch.delenv("EGT_ATTN_WS_SHARED")
    with torch.no_grad():                                            # no gradient wanted: the forward-only workspace
        V2, Hh2, _ = egt_attention(QKV, E, G, None, mask.to(gpu), cfg=cfg)
    assert torch.equal(V2, V) and torch.equal(Hh2, Hh)
    # the workspace-size contract at the C-ABI: the forward's size follows the SHARED bit
    desc = L.AttnDesc(B=B, N=N, H=H, d=d, dtype=L.EGT_F32, flags=L.F_EDGE_INPUT | L.F_GATE_INPUT | L.F_CLIP, clip_lo=-5.0, clip_hi=5.0,
                      random_mask_prob=0.0, attn_dropout=0.0, num_virtual_nodes=0, reserved=0, seed=0)
    small = egt_lib.egt_attn_mfma_fwd_workspace_bytes(C.byref(desc))
    desc.reserved = L.ATTN_WS_SHARED
    assert egt_lib.egt_attn_mfma_fwd_workspace_bytes(C.byref(desc)) == egt_lib.egt_attn_mfma_workspace_bytes(C.byref(desc)) > small
    desc.reserved = 0x40
    ws = torch.empty(egt_lib.egt_attn_mfma_workspace_bytes(C.byref(desc)) or 16, dtype=torch.uint8, device=gpu)
    rc = egt_lib.egt_attn_mfma_fwd(C.byref(desc), L.ptr(QKV), L.ptr(E), L.ptr(G), None, None, None, L.ptr(V2), L.ptr(Hh2),
                                   L.ptr(torch.empty(B, N, H, 4, device=gpu)), L.ptr(ws), L.current_stream())
    assert rc == L.EGT_E_FLAGS                                      # unknown reserved bits are refused, not ignored
