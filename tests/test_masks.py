"""Mask producers: the oracle's restatement against hand-built expectations on CPU;
the product's HIP producers bit-exact against the oracle on the GPU, then FED to the kernels
(the masks the kernels consume are the ones the product produces)."""
import pytest
import torch

from oracle import egt_oracle as O


def test_oracle_node_mask_semantics():
    x = torch.tensor([[3, 0, -1, -1], [-1, 5, 27, -1]])            # padding value -1 (datasets/zinc.py:69-77)
    assert O.node_mask_from_features(x).tolist() == [[True, True, False, False], [False, True, True, False]]
    assert O.node_mask_from_features(x, 2).tolist() == [[True, True, True, True, False, False],
                                                        [True, True, False, True, True, False]]
    f = torch.tensor([[[0.5, -1.0], [-1.0, -1.0], [-1.0, 0.0]]])   # Masking(mask_value=-1.): any feature != -1
    assert O.node_mask_from_masking(f).tolist() == [[True, False, True]]
    assert O.node_mask_from_masking(f, num_virtual_nodes=1).tolist() == [[True, True, False, True]]


def test_oracle_constrained_edge_mask_semantics():
    adj = torch.tensor([[[0., 1.], [1., 0.]]])
    M = O.constrained_edge_mask(adj, 8)
    assert M.shape == (1, 2, 2, 8) and torch.equal(M[..., 0], adj) and torch.equal(M[..., 7], adj)
    Mv = O.constrained_edge_mask(adj, 4, num_virtual_nodes=1)       # graph_model_base.py:248-268
    assert Mv.shape == (1, 3, 3, 4)
    assert torch.equal(Mv[0, :, :, 2], torch.tensor([[1., 1., 1.], [1., 0., 1.], [1., 1., 0.]]))


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,nv", [(1, 1, 0), (3, 37, 0), (128, 64, 0), (5, 150, 2), (2, 9, 1)])
def test_node_mask_producers_bit_exact(B, N, nv, gpu, egt_lib):
    from egt_amd import node_mask_from_features, node_mask_from_masking
    g = torch.Generator().manual_seed(B * 1000 + N)
    n = torch.randint(1, N + 1, (B,), generator=g)
    x = torch.randint(0, 28, (B, N), generator=g)
    x[torch.arange(N)[None, :] >= n[:, None]] = -1
    x[0, 0] = 0                                                     # feature id 0 is a REAL node ((0+1) != 0)
    got = node_mask_from_features(x.to(gpu), nv)
    assert got.dtype == torch.bool and torch.equal(got.cpu(), O.node_mask_from_features(x, nv))
    for F in (1, 3, 8, 11):
        f = torch.randn(B, N, F, generator=g)
        f[torch.arange(N)[None, :] >= n[:, None]] = -1.0
        if N > 1:
            f[0, N - 1] = -1.0; f[0, N - 1, F - 1] = 0.25           # a single differing feature keeps the step
        got = node_mask_from_masking(f.to(gpu), -1.0, nv)
        assert torch.equal(got.cpu(), O.node_mask_from_masking(f, -1.0, nv)), F


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,H,nv", [(1, 1, 8, 0), (4, 37, 8, 0), (16, 64, 8, 0), (2, 21, 4, 0), (3, 20, 8, 2), (2, 7, 3, 1)])
def test_constrained_edge_mask_bit_exact(B, N, H, nv, gpu, egt_lib):
    from egt_amd import constrained_edge_mask
    g = torch.Generator().manual_seed(N + H)
    adj = (torch.rand(B, N, N, generator=g) > 0.6).float()
    got = constrained_edge_mask(adj.to(gpu), H, nv)
    assert got.dtype == torch.float32 and torch.equal(got.cpu(), O.constrained_edge_mask(adj, H, nv))


@pytest.mark.gpu
def test_produced_masks_feed_the_kernels(gpu, egt_lib):
    """features == -1 -> node mask, adjacency -> M, both made by the product's HIP producers and handed
    to the fused 'constrained' block; result = oracle block on the oracle's masks.  Padded / disconnected
    keys get EXACTLY zero attention (checked through the composed inner op's A_tild)."""
    from egt_amd import EGTBlock, node_mask_from_features, constrained_edge_mask
    from util import assert_close, FWD, BWD
    from test_block_gpu import PMAP
    B, N, Dh, De, H = 3, 21, 64, 32, 8
    g = torch.Generator().manual_seed(77)
    n = torch.tensor([21, 9, 14])
    x = torch.randint(0, 28, (B, N), generator=g)
    x[torch.arange(N)[None, :] >= n[:, None]] = -1
    adj = (torch.rand(B, N, N, generator=g) > 0.5).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float()
    mask_g = node_mask_from_features(x.to(gpu))
    M_g = constrained_edge_mask(adj.to(gpu), H)
    mask_o, M_o = O.node_mask_from_features(x), O.constrained_edge_mask(adj, H)
    assert torch.equal(mask_g.cpu(), mask_o) and torch.equal(M_g.cpu(), M_o)
    torch.manual_seed(5)
    params = O.init_block_params(Dh, De, H, generator=g, randomize_norm=True)
    h = torch.randn(B, N, Dh, generator=g); e = torch.randn(B, N, N, De, generator=g)
    dh = torch.randn(B, N, Dh, generator=g); de = torch.randn(B, N, N, De, generator=g)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    h64 = h.double().requires_grad_(); e64 = e.double().requires_grad_()
    ho, eo, _, at = O.block_forward(h64, e64, mask_o, p64, num_heads=H, edge_channel_type="constrained",
                                    attn_mask=M_o.double(), return_inner=True)
    gr = torch.autograd.grad([ho, eo], [h64, e64], [dh.double(), de.double()])
    for fused in (True, False):
        blk = EGTBlock(model_width=Dh, edge_width=De, num_heads=H, edge_channel_type="constrained", fused=fused).to(gpu).eval()
        with torch.no_grad():
            for k, (m, a) in PMAP.items():
                getattr(getattr(blk, m), a).copy_(params[k].to(gpu))
        hg = h.to(gpu).requires_grad_(); eg = e.to(gpu).requires_grad_()
        h2, e2 = blk(hg, eg, mask_g, M_g)
        torch.autograd.backward([h2, e2], [dh.to(gpu), de.to(gpu)])
        assert_close(h2, ho, name=f"h_out(fused={fused})", **FWD)
        assert_close(e2, eo, name=f"e_out(fused={fused})", **FWD)
        assert_close(hg.grad, gr[0], name="dh", **BWD)
        assert_close(eg.grad, gr[1], name="de", **BWD)
    # exact zeros where a mask forbids the key
    forbidden = (~mask_o)[:, None, :, None] | (M_o == 0)
    assert (at[forbidden.expand_as(at)] == 0).all()
