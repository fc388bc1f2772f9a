"""Branches of mha_block / edge_update_* that no shipped config switches on but the interface
carries: add_n_norm (post-norm, graph_xformer_model_base.py:108-109,142-143,220-221),
node / edge dropout (drp_mha :138-139, drp_edge :216-217) -- with the sample injected for parity and
with the device RNG for the statistics."""
import pytest
import torch

from oracle import egt_oracle as O
from util import assert_close, FWD, BWD
from test_block_gpu import PMAP

pytestmark = pytest.mark.gpu


def _mk(B, N, Dh, De, seed):
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(B, N, Dh, generator=g); e = torch.randn(B, N, N, De, generator=g) * 1.2
    dh = torch.randn(B, N, Dh, generator=g); de = torch.randn(B, N, N, De, generator=g)
    mask = torch.ones(B, N, dtype=torch.bool); mask[B - 1, N - 3:] = False
    params = O.init_block_params(Dh, De, 8, generator=g, randomize_norm=True)
    return g, h, e, dh, de, mask, params


def _load(blk, params, dev):
    with torch.no_grad():
        for k, (m, a) in PMAP.items():
            if hasattr(blk, m):
                getattr(getattr(blk, m), a).copy_(params[k].to(dev))


def _check(blk, dev, h, e, dh, de, mask, params, okw, fkw=None):
    fkw = fkw or {}
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    h64 = h.double().requires_grad_(); e64 = e.double().requires_grad_()
    ho, eo = O.block_forward(h64, e64, mask, p64, num_heads=8, **okw)
    names = [k for k, (m, _) in PMAP.items() if hasattr(blk, m)]
    gr = torch.autograd.grad([ho, eo], [h64, e64] + [p64[k] for k in names], [dh.double(), de.double()], allow_unused=True)
    hg = h.to(dev).requires_grad_(); eg = e.to(dev).requires_grad_()
    h2, e2 = blk(hg, eg, mask.to(dev), **{k: v.to(dev) for k, v in fkw.items()})
    torch.autograd.backward([h2, e2], [dh.to(dev), de.to(dev)])
    assert_close(h2, ho, name="h_out", **FWD)
    assert_close(e2, eo, name="e_out", **FWD)
    assert_close(hg.grad, gr[0], name="dh", **BWD)
    if gr[1] is not None:
        assert_close(eg.grad, gr[1], name="de", **BWD)
    for k, gref in zip(names, gr[2:]):
        m, a = PMAP[k]
        if gref is not None:
            assert_close(getattr(getattr(blk, m), a).grad, gref, name=k, **BWD)


@pytest.mark.parametrize("ect,gate", [("residual", True), ("residual", False), ("none", True), ("bias", True)])
def test_block_add_n_norm_vs_oracle(ect, gate, gpu, egt_lib):
    from egt_amd import EGTBlock
    _, h, e, dh, de, mask, params = _mk(2, 19, 64, 32, 31)
    blk = EGTBlock(model_width=64, edge_width=32, num_heads=8, gate_attention=gate, edge_channel_type=ect,
                   add_n_norm=True).to(gpu).eval()
    _load(blk, params, gpu)
    _check(blk, gpu, h, e, dh, de, mask, params,
           dict(edge_channel_type=ect, gate_attention=gate, add_n_norm=True))


@pytest.mark.parametrize("pn,pe", [(0.3, 0.0), (0.0, 0.25), (0.2, 0.4)])
def test_block_node_edge_dropout_injected_vs_oracle(pn, pe, gpu, egt_lib):
    from egt_amd import EGTBlock
    g, h, e, dh, de, mask, params = _mk(2, 17, 64, 16, 47)
    nk = torch.rand(h.shape, generator=g) >= pn
    ek = torch.rand(e.shape, generator=g) >= pe
    blk = EGTBlock(model_width=64, edge_width=16, num_heads=8, node_dropout=pn, edge_dropout=pe).to(gpu).train()
    _load(blk, params, gpu)
    _check(blk, gpu, h, e, dh, de, mask, params,
           dict(node_keep=nk, node_dropout=pn, edge_keep=ek, edge_dropout=pe),
           dict(node_keep=nk, edge_keep=ek))


def test_block_dropout_device_rng_and_eval_identity(gpu, egt_lib):
    """without an injected sample the device RNG draws it: about (1-p) of the update survives, the
    survivors are scaled by 1/(1-p); in eval mode dropout is the identity."""
    from egt_amd import EGTBlock
    _, h, e, dh, de, mask, params = _mk(4, 32, 64, 64, 3)
    pn, pe = 0.5, 0.5
    blk = EGTBlock(model_width=64, edge_width=64, num_heads=8, node_dropout=pn, edge_dropout=pe).to(gpu)
    ref = EGTBlock(model_width=64, edge_width=64, num_heads=8).to(gpu).eval()
    _load(blk, params, gpu); _load(ref, params, gpu)
    hg, eg, mg = h.to(gpu), e.to(gpu), mask.to(gpu)
    h0, e0 = ref(hg, eg, mg)
    blk.eval()
    h1, e1 = blk(hg, eg, mg)
    assert_close(h1, h0, name="eval h", rtol=1e-4, arel=5e-5); assert_close(e1, e0, name="eval e", rtol=1e-4, arel=5e-5)
    blk.train()
    torch.manual_seed(0)
    h2, e2 = blk(hg, eg, mg)
    ue, u0 = (e2 - eg), (e0 - eg)                 # the edge update with / without dropout
    dropped = (ue == 0) & (u0.abs() > 1e-3)
    kept = (ue != 0) & (u0.abs() > 1e-3)
    frac = float(dropped.sum()) / float((dropped | kept).sum())
    assert abs(frac - pe) < 0.01, frac
    assert_close(ue[kept], u0[kept] / (1 - pe), name="kept edge updates", rtol=1e-3, arel=1e-4)
