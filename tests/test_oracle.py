"""CPU tests that pin the oracle (it is 'parity unpinned' by the reference, which
ships no vectors): hand-derived backward vs autograd, the algebraic identities
and probed facts of SURVEY §8(a)/(c), and the committed golden fixtures."""
import glob
import os

import numpy as np
import pytest
import torch

import cases as CS
from oracle import egt_oracle as O
from util import assert_close, load_golden


@pytest.mark.parametrize("name", list(CS.ATTN_CASES))
def test_hand_backward_matches_autograd(name):
    inp, attrs, _ = CS.make_attn_case(name)
    ref = CS.attn_oracle(inp, attrs)
    cv = lambda t: None if t is None else t.double()
    dQ, dE, dG = O.egt_backward(cv(inp["QKV"]), cv(inp["E"]), cv(inp["G"]), cv(inp["M"]), inp["mask"],
                                cv(inp["dV"]), cv(inp["dH"]), rand_mask=inp["rand_mask"],
                                drop_keep=inp["drop_keep"], **attrs)
    assert_close(dQ, ref["dQKV"], rtol=1e-10, arel=1e-12, name="dQKV")
    if dE is not None:
        assert_close(dE, ref["dE"], rtol=1e-10, arel=1e-12, name="dE")
    if dG is not None:
        assert_close(dG, ref["dG"], rtol=1e-10, arel=1e-12, name="dG")


def test_qkv_channel_layout():
    # c = s*d*H + k*H + h (egt_layers.py:73-76); output channel k*H + h (:139-141)
    B, N, H, d = 1, 3, 8, 6
    QKV = torch.arange(B * N * 3 * d * H, dtype=torch.float64).reshape(B, N, 3 * d * H)
    Q, K, V = QKV.reshape(B, N, 3, d, H).unbind(2)
    for s, T in enumerate((Q, K, V)):
        for k in (0, 3, 5):
            for h in (0, 7):
                assert T[0, 1, k, h] == QKV[0, 1, s * d * H + k * H + h]


def test_masked_positions_exactly_zero_fp32():
    inp, attrs, _ = CS.make_attn_case("gated_d8_clip")
    V, Hh, At = O.egt_forward(inp["QKV"], inp["E"], inp["G"], None, inp["mask"], **attrs)
    pad = ~inp["mask"]
    for b in range(pad.shape[0]):
        assert (At[b][:, pad[b], :] == 0).all()
    # H_hat - E lies in the clip range
    A = Hh - inp["E"]
    assert A.max() <= 5 + 1e-5 and A.min() >= -5 - 1e-5
    assert (A.abs() > 4.99).any(), "case must exercise clip saturation"


def test_all_masked_row_behaviour():
    inp, attrs, _ = CS.make_attn_case("gated_allmasked")
    V, Hh, At = O.egt_forward(inp["QKV"], inp["E"], inp["G"], None, inp["mask"], **attrs)
    assert (At[1] == 0).all() and (V[1] == 0).all()          # gated: zero
    Vu, _, Atu = O.egt_forward(inp["QKV"], inp["E"], None, None, inp["mask"], **attrs)
    assert torch.allclose(Atu[1], torch.full_like(Atu[1], 1.0 / 6))  # ungated: uniform 1/N


def test_reduces_to_sdpa():
    torch.manual_seed(0)
    B, N, H, d = 2, 9, 8, 8
    QKV = torch.randn(B, N, 3 * d * H, dtype=torch.float64)
    V, _, _ = O.egt_forward(QKV, None, None, None, None, num_heads=H, clip_logits_value=None)
    Q, K, Vv = QKV.reshape(B, N, 3, d, H).unbind(2)
    ref = torch.nn.functional.scaled_dot_product_attention(
        Q.permute(0, 3, 1, 2), K.permute(0, 3, 1, 2), Vv.permute(0, 3, 1, 2))  # b,h,l,d
    assert_close(V.reshape(B, N, d, H), ref.permute(0, 2, 3, 1), rtol=1e-10, arel=1e-12, name="sdpa")


def test_padded_key_permutation_invariance():
    inp, attrs, _ = CS.make_attn_case("gated_d8_clip")
    V0, _, _ = O.egt_forward(inp["QKV"], inp["E"], inp["G"], None, inp["mask"], **attrs)
    # scramble the padded keys of graph 1 (nodes 3,4 are padding)
    QKV = inp["QKV"].clone(); E = inp["E"].clone(); G = inp["G"].clone()
    QKV[1, 3:] = torch.randn_like(QKV[1, 3:]); E[1, :, 3:] = 7.0; G[1, :, 3:] = -3.0
    V1, _, _ = O.egt_forward(QKV, E, G, None, inp["mask"], **attrs)
    assert torch.allclose(V0[1, :3], V1[1, :3], atol=1e-6)


def test_head_independence():
    inp, attrs, _ = CS.make_attn_case("ungated")
    H = 8
    V0, _, _ = O.egt_forward(inp["QKV"], inp["E"], None, None, inp["mask"], **attrs)
    QKV = inp["QKV"].clone()
    QKV.reshape(2, 9, 3, 8, H)[..., 3] += 1.0   # perturb head 3 only
    V1, _, _ = O.egt_forward(QKV, inp["E"], None, None, inp["mask"], **attrs)
    diff = (V1 - V0).reshape(2, 9, 8, H).abs().amax(dim=(0, 1, 2))
    assert diff[3] > 0 and (diff[[0, 1, 2, 4, 5, 6, 7]] == 0).all()


def test_constructor_errors():
    with pytest.raises(ValueError):
        O.egt_forward(torch.zeros(1, 2, 48), None, None, None, None, scale_degree=True)


def test_mask_producers():
    x = torch.tensor([[3, 0, -1, -1]])
    assert O.node_mask_from_features(x).tolist() == [[True, True, False, False]]
    adj = torch.eye(3)[None]
    assert O.constrained_edge_mask(adj, 8).shape == (1, 3, 3, 8)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(CS.GOLDEN_DIR, "attn_*.npz"))))
def test_oracle_reproduces_attn_golden(path):
    name = os.path.basename(path)[len("attn_"):-4]
    g = load_golden(path)
    inp, attrs, _ = CS.make_attn_case(name)
    for k, v in g["in"].items():
        assert np.array_equal(CS.to_np(inp[k]), v), f"input {k} drifted"
    out = CS.attn_oracle(inp, attrs)
    for k, v in g["out"].items():
        assert_close(out[k].float(), v, rtol=1e-6, arel=1e-6, name=k)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(CS.GOLDEN_DIR, "block_*.npz"))))
def test_oracle_reproduces_block_golden(path):
    name = os.path.basename(path)[len("block_"):-4]
    g = load_golden(path)
    inp, params, attrs, _ = CS.make_block_case(name)
    out = CS.block_oracle(inp, params, attrs)
    for k, v in g["out"].items():
        assert_close(out[k].float(), v, rtol=1e-6, arel=1e-6, name=k)
    for k, v in g["dparams"].items():
        assert_close(out["dparams"][k].float(), v, rtol=1e-6, arel=1e-6, name=k)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(CS.GOLDEN_DIR, "ffn_*.npz"))))
def test_oracle_reproduces_ffn_golden(path):
    name = os.path.basename(path)[len("ffn_"):-4]
    g = load_golden(path)
    inp, params, c = CS.make_ffn_case(name)
    for k, v in g["in"].items():
        assert np.array_equal(CS.to_np(inp[k]), v), f"input {k} drifted"
    out = CS.ffn_oracle(inp, params, c)
    assert_close(out["y"].float(), g["out"]["y"], rtol=1e-6, arel=1e-6, name="y")
    assert_close(out["dx"].float(), g["out"]["dx"], rtol=1e-6, arel=1e-6, name="dx")
    for k, v in g["dparams"].items():
        assert_close(out["dparams"][k].float(), v, rtol=1e-6, arel=1e-6, name=k)


def test_ffn_oracle_backward_and_identities():
    """autograd of the restatement vs finite differences (fp64), and: zero second Dense => identity;
    the LayerNorm makes the FFN branch invariant to a per-row shift of x."""
    from oracle import egt_oracle as O
    inp, params, c = CS.make_ffn_case("edge_w16_elu")
    x = inp["x"][:1, :2, :2].double().requires_grad_()
    p = {k: v.double().requires_grad_() for k, v in params.items()}
    assert torch.autograd.gradcheck(lambda xx, *ps: O.ffn_forward(xx, dict(zip(CS.FFN_NAMES, ps)), activation="elu"),
                                    (x, *[p[k] for k in CS.FFN_NAMES]), eps=1e-6, atol=1e-6)
    p0 = dict(p); p0["lr2_kernel"] = torch.zeros_like(p["lr2_kernel"]); p0["lr2_bias"] = torch.zeros_like(p["lr2_bias"])
    assert torch.equal(O.ffn_forward(x, p0), x)
    y0 = O.ffn_forward(x, p) - x
    y1 = O.ffn_forward(x + 3.0, p) - (x + 3.0)
    assert_close(y1, y0, rtol=1e-9, arel=1e-9, name="shift invariance")


def test_all_masked_row_ungated_is_uniform_like_fp32_reference():
    """SURVEY 8(a) a5/a11 (probed fp32 facts): logit + (-1e9) rounds to exactly -1e9, so a row whose
    keys are all masked attends UNIFORMLY over its least-masked keys in the ungated variant -- the
    fp64 evaluation of the oracle must reproduce that rounding, not keep the logit differences."""
    from oracle import egt_oracle as O
    torch.manual_seed(0)
    B, N, H, d = 1, 4, 8, 2
    QKV = torch.randn(B, N, 3 * d * H, dtype=torch.float64)
    E = torch.randn(B, N, N, H, dtype=torch.float64)
    mask = torch.ones(B, N, dtype=torch.bool)
    rm = torch.zeros(B, N, N, H, dtype=torch.bool)
    rm[0, 1] = True                       # query row 1: every key hit by the random mask
    rm[0, 2, :3] = True                   # row 2: keys 0..2 masked, key 3 free
    V_att, _, A = O.egt_forward(QKV, E, None, None, mask, num_heads=H, rand_mask=rm)
    assert torch.allclose(A[0, 1], torch.full((N, H), 1.0 / N, dtype=torch.float64), atol=1e-12)
    assert torch.allclose(A[0, 2, 3], torch.ones(H, dtype=torch.float64)) and float(A[0, 2, :3].abs().max()) == 0.0
    A32 = O.egt_forward(QKV.float(), E.float(), None, None, mask, num_heads=H, rand_mask=rm)[2]
    assert torch.allclose(A32.double(), A, atol=1e-6)


def test_fp32_oracle_close_to_fp64():
    inp, params, attrs, _ = CS.make_block_case("residual_zinc500k")
    o64 = CS.block_oracle(inp, params, attrs, torch.float64)
    o32 = CS.block_oracle(inp, params, attrs, torch.float32)
    assert_close(o32["e_out"], o64["e_out"], rtol=1e-4, arel=2e-5, name="e_out")
    assert_close(o32["de"], o64["de"], rtol=1e-3, arel=1e-4, name="de")
