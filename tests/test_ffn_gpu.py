"""GPU parity of the fused channel FFN vs the fp64 oracle."""
import pytest
import torch

from util import assert_close, FWD, BWD

pytestmark = pytest.mark.gpu

NAMES = ("norm_gamma", "norm_beta", "lr1_kernel", "lr1_bias", "lr2_kernel", "lr2_bias")


def _run(shape, act, gpu, seed=0, W=64, matmul="f32", fwd_tol=FWD, bwd_tol=BWD):
    from egt_amd import FFN
    from oracle import egt_oracle as O
    torch.manual_seed(seed)
    m = FFN(W, activation=act, matmul=matmul).to(gpu)
    with torch.no_grad():
        for n in ("norm_gamma", "norm_beta", "lr1_bias", "lr2_bias"):
            getattr(m, n).add_(0.3 * torch.randn_like(getattr(m, n)))
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(*shape, W, generator=g) * 1.5 + 0.2
    dy = torch.randn(*shape, W, generator=g)
    xg = x.to(gpu).requires_grad_()
    y = m(xg)
    y.backward(dy.to(gpu))
    p64 = {n: getattr(m, n).detach().double().cpu().requires_grad_() for n in NAMES}
    x64 = x.double().requires_grad_()
    yo = O.ffn_forward(x64, p64, activation=act)
    gr = torch.autograd.grad(yo, [x64] + [p64[n] for n in NAMES], dy.double())
    assert_close(y, yo, name="y", **fwd_tol)
    assert_close(xg.grad, gr[0], name="dx", **bwd_tol)
    for n, gref in zip(NAMES, gr[1:]):
        assert_close(getattr(m, n).grad, gref, name="d" + n, **bwd_tol)


@pytest.mark.parametrize("W,shape,act", [(64, (2, 24, 24), "elu"), (64, (3, 17, 17), "relu"), (64, (4, 64, 64), "elu"), (64, (1, 5), "elu"),
                                          (48, (2, 37, 37), "elu"), (32, (2, 19, 19), "elu"), (16, (2, 23, 23), "relu")])
def test_ffn_bf16x3_matmul_holds_the_fp32_tolerances(W, shape, act, gpu, egt_lib):
    """matmul="bf16x3": 3-term bfloat16 split products on the bf16 matrix pipe (per-product error 2^-16).
    SAME tolerances as the exact-fp32 kernels."""
    _run(shape, act, gpu, seed=W + 1, W=W, matmul="bf16x3")


@pytest.mark.parametrize("W,shape", [(64, (2, 24, 24)), (48, (2, 21, 21)), (16, (3, 9))])
def test_ffn_plain_bf16_matmul(W, shape, gpu, egt_lib):
    """matmul="bf16": plain bfloat16 products, fp32 accumulate -- SURVEY 8(c)'s bf16 tolerance (rtol 2e-2)."""
    tol = dict(rtol=2e-2, arel=1e-2, l2=2e-2)
    _run(shape, "elu", gpu, seed=W + 2, W=W, matmul="bf16", fwd_tol=tol, bwd_tol=tol)


@pytest.mark.parametrize("shape,act", [((2, 24, 24), "elu"), ((3, 17, 17), "elu"), ((2, 37), "relu"),
                                        ((1, 5), "elu"), ((4, 64, 64), "elu")])
def test_ffn_vs_oracle(shape, act, gpu, egt_lib):
    """edge [B,N,N,64] and node [B,N,64] shapes, ragged row counts (rows % 16 != 0), both activations"""
    _run(shape, act, gpu)


@pytest.mark.parametrize("W,shape,act", [(48, (2, 37, 37), "elu"), (48, (3, 21), "relu"), (32, (2, 19, 19), "elu"),
                                          (16, (2, 23, 23), "elu"), (16, (5, 7), "relu")])
def test_ffn_other_widths_vs_oracle(W, shape, act, gpu, egt_lib):
    """widths 16 / 32 / 48 (BASELINE config 1 is Dh = De = 48)"""
    _run(shape, act, gpu, seed=W, W=W)


@pytest.mark.parametrize("shape,act", [((2, 20, 20), "elu"), ((3, 15, 15), "relu"), ((1, 7), "elu"), ((1,), "elu"), ((128, 9), "elu"),
                                        ((2, 120, 120), "elu"), ((5, 251, 251), "relu")])
def test_ffn_width8_vs_oracle(shape, act, gpu, egt_lib):
    """edge_width 8 (BASELINE configs 3 / 4: CIFAR10, PATTERN): the row-per-lane kernels (k_ffn8_*), weight gradients through
    the transposed LDS images; odd and tiny row counts, a partial last 64-row chunk, more chunks than workgroups"""
    _run(shape, act, gpu, seed=8, W=8)


@pytest.mark.parametrize("W", [64, 8])
def test_ffn_bit_reproducible_and_linear_in_dy(W, gpu, egt_lib):
    from egt_amd import FFN
    torch.manual_seed(5)
    m = FFN(W).to(gpu)
    x = torch.randn(2, 40, 40, W, device=gpu)
    dy = torch.randn_like(x)
    outs = []
    for scale in (1.0, 1.0, 2.0):
        for prm in m.parameters():
            prm.grad = None
        xg = x.clone().requires_grad_()
        y = m(xg)
        y.backward(dy * scale)
        outs.append((y.detach().clone(), xg.grad.clone(), m.lr1_kernel.grad.clone(), m.lr2_kernel.grad.clone()))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)                       # deterministic partial reductions
    for a, b in zip(outs[0][1:], outs[2][1:]):
        assert_close(b, 2.0 * a, name="linearity", rtol=1e-5, arel=1e-6)


def test_ffn_rejects_uncovered(gpu, egt_lib):
    from egt_amd import ffn
    x = torch.randn(4, 40, device=gpu)
    z = torch.zeros(40, device=gpu)
    with pytest.raises(ValueError):
        ffn(x, z, z, torch.zeros(40, 80, device=gpu), torch.zeros(80, device=gpu), torch.zeros(80, 40, device=gpu), z)


@pytest.mark.parametrize("N,De", [(32, 64), (23, 8), (40, 8)])
def test_layer_stack_attention_plus_ffn_vs_oracle(N, De, gpu, egt_lib):
    """EGTLayerStack = the reference's full layer loop (attention block, then ffn_block on both
    channel types; graph_xformer_model_base.py:336-341) vs the fp64 oracle composition.  De = 8 is the
    edge width of BASELINE configs 3 / 4 (CIFAR10, PATTERN): whole layers run there since the width-8 FFN."""
    from egt_amd import EGTLayerStack
    from oracle import egt_oracle as O
    from test_block_gpu import PMAP
    torch.manual_seed(17)
    B, Ly = 2, 2
    st = EGTLayerStack(model_height=Ly, model_width=64, edge_width=De, num_heads=8, fused=True).to(gpu).eval()
    g = torch.Generator().manual_seed(2)
    h = torch.randn(B, N, 64, generator=g); e = torch.randn(B, N, N, De, generator=g)
    mask = torch.ones(B, N, dtype=torch.bool); mask[0, N - 4:] = False
    dh = torch.randn(B, N, 64, generator=g); de = torch.randn(B, N, N, De, generator=g)
    hg = h.to(gpu).requires_grad_(); eg = e.to(gpu).requires_grad_()
    h2, e2 = st(hg, eg, mask.to(gpu))
    torch.autograd.backward([h2, e2], [dh.to(gpu), de.to(gpu)])
    h64 = h.double().requires_grad_(); e64 = e.double().requires_grad_()
    ho, eo = h64, e64
    for i in range(Ly):
        bp = {k: getattr(getattr(st.blocks[i], m), a_).detach().double().cpu() for k, (m, a_) in PMAP.items()}
        ho, eo = O.block_forward(ho, eo, mask, bp, num_heads=8)
        eo = O.ffn_forward(eo, {n: getattr(st.ffn_edge[i], n).detach().double().cpu() for n in NAMES})
        ho = O.ffn_forward(ho, {n: getattr(st.ffn_node[i], n).detach().double().cpu() for n in NAMES})
    gr = torch.autograd.grad([ho, eo], [h64, e64], [dh.double(), de.double()])
    assert_close(h2, ho, name="h_out", rtol=3e-4, arel=1e-4)
    assert_close(e2, eo, name="e_out", rtol=3e-4, arel=1e-4)
    assert_close(hg.grad, gr[0], name="dh", **BWD)
    assert_close(eg.grad, gr[1], name="de", **BWD)


@pytest.mark.parametrize("name", ["edge_w64_elu", "node_w48_relu", "edge_w16_elu"])
def test_ffn_vs_golden(name, gpu, egt_lib):
    """the committed fixtures (tests/golden/ffn_*.npz, written by the oracle) through the C-ABI"""
    import os
    import cases as CS
    from util import load_golden
    from egt_amd import ffn
    g = load_golden(os.path.join(CS.GOLDEN_DIR, f"ffn_{name}.npz"))
    c = CS.FFN_CASES[name]
    prm = {k: torch.from_numpy(g["params"][k]).to(gpu).requires_grad_() for k in CS.FFN_NAMES}
    x = torch.from_numpy(g["in"]["x"]).to(gpu).requires_grad_()
    y = ffn(x, *[prm[k] for k in CS.FFN_NAMES], activation=c["act"])
    y.backward(torch.from_numpy(g["in"]["dy"]).to(gpu))
    assert_close(y, g["out"]["y"], name="y", **FWD)
    assert_close(x.grad, g["out"]["dx"], name="dx", **BWD)
    for k in CS.FFN_NAMES:
        assert_close(prm[k].grad, g["dparams"][k], name=k, **BWD)


@pytest.mark.parametrize("W,matmul", [(64, "f32"), (48, "bf16x3"), (8, "f32")])
def test_ffn_bwd_on_a_fresh_workspace_equals_the_prepared_one(W, matmul, gpu, egt_lib):
    """egt_ffn_bwd called the plain way (flags = 0: it prepares its own operands) gives the bits of the autograd path, which
    hands the forward's workspace back with EGT_FFN_WS_PREPARED"""
    import ctypes as C
    from egt_amd import FFN, _lib as L
    from egt_amd.ffn import _desc, _pstruct
    torch.manual_seed(11)
    m = FFN(W, matmul=matmul).to(gpu)
    x = torch.randn(3, 33, 33, W, device=gpu)
    dy = torch.randn_like(x)
    xg = x.clone().requires_grad_()
    m(xg).backward(dy)
    params = [getattr(m, n).detach().contiguous() for n in NAMES]
    desc = _desc(x.numel() // W, W, "elu", 1e-3, matmul)
    assert desc.flags == 0
    ws = torch.empty(egt_lib.egt_ffn_workspace_bytes(C.byref(desc)), dtype=torch.uint8, device=gpu)
    dx = torch.empty_like(x)
    grads = [torch.empty_like(p) for p in params]
    pst, gst = _pstruct(params), _pstruct(grads)
    L.check(egt_lib.egt_ffn_bwd(C.byref(desc), C.byref(pst), L.ptr(x), L.ptr(dy), L.ptr(dx), C.byref(gst), L.ptr(ws),
                                L.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(dx, xg.grad)
    for n, g in zip(NAMES, grads):
        assert torch.equal(g, getattr(m, n).grad), n
