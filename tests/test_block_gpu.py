"""GPU parity of the outer op (h, e, mask) -> (h', e') vs the fp64 oracle and the
golden fixtures, for every edge_channel_type, composed and fused paths."""
import os

import pytest
import torch

import cases as CS

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
from util import assert_close, load_golden, FWD, BWD

pytestmark = pytest.mark.gpu

PMAP = {  # oracle/Keras name -> (submodule, attr)
    "norm_edge.gamma": ("norm_edge", "gamma"), "norm_edge.beta": ("norm_edge", "beta"),
    "attention_gates.kernel": ("attention_gates", "kernel"), "attention_gates.bias": ("attention_gates", "bias"),
    "dense_edge_b.kernel": ("dense_edge_b", "kernel"), "dense_edge_b.bias": ("dense_edge_b", "bias"),
    "norm_mha.gamma": ("norm_mha", "gamma"), "norm_mha.beta": ("norm_mha", "beta"),
    "dense_qkv.kernel": ("dense_qkv", "kernel"), "dense_qkv.bias": ("dense_qkv", "bias"),
    "dense_mha.kernel": ("dense_mha", "kernel"), "dense_mha.bias": ("dense_mha", "bias"),
    "dense_edge_r.kernel": ("dense_edge_r", "kernel"), "dense_edge_r.bias": ("dense_edge_r", "bias"),
}


def build_block(c, attrs, params, dev, fused):
    from egt_amd import EGTBlock
    blk = EGTBlock(model_width=c["Dh"], edge_width=c["De"], num_heads=8,
                   gate_attention=attrs["gate_attention"], edge_activation=attrs["edge_activation"],
                   edge_channel_type=attrs["edge_channel_type"],
                   random_mask_prob=0.5 if c.get("rand_p") else 0.0, fused=fused).to(dev)
    with torch.no_grad():
        for k, (m, a) in PMAP.items():
            if hasattr(blk, m):
                getattr(getattr(blk, m), a).copy_(params[k].to(dev))
    return blk


def run_block(name, dev, fused):
    inp, params, attrs, c = CS.make_block_case(name)
    blk = build_block(c, attrs, params, dev, fused)
    blk.train(c.get("rand_p") is not None)
    cu = lambda t: None if t is None else t.to(dev)
    h = cu(inp["h"]).requires_grad_()
    e = cu(inp["e"]).requires_grad_()
    h2, e2 = blk(h, e, cu(inp["mask"]), cu(inp["attn_mask"]), rand_mask=cu(inp["rand_mask"]))
    loss = (h2 * cu(inp["dh"])).sum() + (e2 * cu(inp["de"])).sum()
    loss.backward()
    out = dict(h_out=h2.detach(), e_out=e2.detach(), dh=h.grad, de=e.grad)
    dparams = {}
    for k, (m, a) in PMAP.items():
        if hasattr(blk, m):
            dparams[k] = getattr(getattr(blk, m), a).grad
    return out, dparams, (inp, params, attrs)


def compare(out, dparams, ref, ref_dparams):
    assert_close(out["h_out"], ref["h_out"], name="h_out", **FWD)
    assert_close(out["e_out"], ref["e_out"], name="e_out", **FWD)
    assert_close(out["dh"], ref["dh"], name="dh", **BWD)
    if ref["de"] is not None:
        assert_close(out["de"], ref["de"], name="de", **BWD)
    for k, g in ref_dparams.items():
        if g is None:
            assert dparams.get(k) is None or float(dparams[k].abs().max()) == 0.0, k
            continue
        assert_close(dparams[k], g, name=k, **BWD)


@pytest.mark.parametrize("name", list(CS.BLOCK_CASES))
def test_block_composed_vs_oracle(name, gpu, egt_lib):
    out, dparams, (inp, params, attrs) = run_block(name, gpu, fused=False)
    ref = CS.block_oracle(inp, params, attrs)
    compare(out, dparams, ref, ref.pop("dparams"))


@pytest.mark.parametrize("name", [n for n in CS.BLOCK_CASES if n != "residual_n64"])
def test_block_composed_vs_golden(name, gpu, egt_lib):
    g = load_golden(os.path.join(CS.GOLDEN_DIR, f"block_{name}.npz"))
    out, dparams, _ = run_block(name, gpu, fused=False)
    ref = {k: torch.from_numpy(v) for k, v in g["out"].items()}
    ref.setdefault("de", None)
    compare(out, dparams, ref, {k: torch.from_numpy(v) for k, v in g["dparams"].items()})


FUSED_CASES = ["residual_zinc500k", "residual_zinc100k", "residual_pattern", "residual_randmask",
               "constrained", "ungated_residual", "residual_n64", "bias"]


@pytest.mark.parametrize("name", FUSED_CASES)
def test_block_fused_vs_oracle(name, gpu, egt_lib):
    out, dparams, (inp, params, attrs) = run_block(name, gpu, fused=True)
    ref = CS.block_oracle(inp, params, attrs)
    compare(out, dparams, ref, ref.pop("dparams"))


@pytest.mark.parametrize("name", [n for n in FUSED_CASES if n != "residual_n64"])
def test_block_fused_vs_golden(name, gpu, egt_lib):
    g = load_golden(os.path.join(CS.GOLDEN_DIR, f"block_{name}.npz"))
    out, dparams, _ = run_block(name, gpu, fused=True)
    ref = {k: torch.from_numpy(v) for k, v in g["out"].items()}
    compare(out, dparams, ref, {k: torch.from_numpy(v) for k, v in g["dparams"].items()})


def test_fused_refuses_uncovered_config(gpu, egt_lib):
    from egt_amd import EGTBlock
    blk = EGTBlock(model_width=64, edge_width=16, edge_channel_type="bias", edge_activation="lrelu2", fused=True).to(gpu)
    with pytest.raises(RuntimeError, match="not covered"):
        blk(torch.zeros(1, 4, 64, device=gpu), torch.zeros(1, 4, 4, 16, device=gpu))


def test_fused_in_kernel_random_mask(gpu, egt_lib):
    """Training-mode fused block with the in-kernel counter-hash mask == composed
    block fed the oracle-side replica of that mask; backward reuses the sample."""
    from egt_amd import EGTBlock
    from oracle import rng_ref
    inp, params, attrs, c = CS.make_block_case("residual_n64")
    p = 0.1
    blk = build_block(dict(c, rand_p=p), attrs, params, gpu, fused=True)
    blk.mha.random_mask_prob = p
    blk.train()
    cu = lambda t: None if t is None else t.to(gpu)
    h = cu(inp["h"]).requires_grad_(); e = cu(inp["e"]).requires_grad_()
    h2, e2 = blk(h, e, cu(inp["mask"]))
    seed = (blk.mha.seed * 0x9E3779B97F4A7C15 + blk.mha._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    (h2 * cu(inp["dh"])).sum().add((e2 * cu(inp["de"])).sum()).backward()
    inp2 = dict(inp)
    inp2["rand_mask"] = torch.from_numpy(rng_ref.random_mask(seed, c["B"], c["N"], 8, p))
    ref = CS.block_oracle(inp2, params, attrs)
    assert_close(h2, ref["h_out"], name="h_out", **FWD)
    assert_close(e2, ref["e_out"], name="e_out", **FWD)
    assert_close(e.grad, ref["de"], name="de", **BWD)
    assert_close(h.grad, ref["dh"], name="dh", **BWD)


def test_fused_stack_matches_composed(gpu, egt_lib):
    from egt_amd import EGTStack
    torch.manual_seed(3)
    kw = dict(model_height=3, model_width=64, edge_width=64, num_heads=8)
    a = EGTStack(fused=True, stack_call=False, **kw).to(gpu).eval()
    b = EGTStack(fused=False, **kw).to(gpu).eval()
    b.load_state_dict(a.state_dict())
    h = torch.randn(2, 24, 64, device=gpu); e = torch.randn(2, 24, 24, 64, device=gpu)
    mask = torch.ones(2, 24, dtype=torch.bool, device=gpu); mask[1, 17:] = False
    (h1, e1), (h2, e2) = a(h, e, mask), b(h, e, mask)
    assert_close(h1, h2, name="h", rtol=1e-4, arel=5e-5)
    assert_close(e1, e2, name="e", rtol=1e-4, arel=5e-5)


@pytest.mark.parametrize("Ly", [17, 40, 58])
def test_deep_stack_one_call_equals_block_calls(Ly, gpu, egt_lib):
    """the first launch of a stack's forward carries the edge-weight preparation of EVERY layer (k_node_pre_stack: instantiated for up
    to 16 / up to 40 layers per kernel-argument block -- Ly = 17 and Ly = 40 run the 40-layer instance, the second at its limit;
    deeper stacks, Ly = 58, fall back to the separate k_edge_prep launch): forward outputs and every gradient
    of the one-call stack equal the block-by-block fused calls bit for bit (same kernels, same order of operations)"""
    from egt_amd import EGTStack
    torch.manual_seed(Ly)
    kw = dict(model_height=Ly, model_width=64, edge_width=64, num_heads=8)
    a = EGTStack(fused=True, **kw).to(gpu).eval()
    b = EGTStack(fused=True, stack_call=False, **kw).to(gpu).eval()
    b.load_state_dict(a.state_dict())
    g = torch.Generator().manual_seed(Ly)
    h = (0.5 * torch.randn(2, 16, 64, generator=g)).to(gpu); e = (0.5 * torch.randn(2, 16, 16, 64, generator=g)).to(gpu)
    mask = torch.ones(2, 16, dtype=torch.bool, device=gpu); mask[1, 11:] = False
    dh = torch.randn(2, 16, 64, generator=g).to(gpu); de = torch.randn(2, 16, 16, 64, generator=g).to(gpu)
    out = []
    for st in (a, b):
        hh = h.clone().requires_grad_(); ee = e.clone().requires_grad_()
        h2, e2 = st(hh, ee, mask)
        torch.autograd.backward([h2, e2], [dh, de])
        out.append((h2.detach(), e2.detach(), hh.grad, ee.grad, {n: p.grad.clone() for n, p in st.named_parameters()}))
    assert a.last_path == "fused-stack"
    (h1, e1, dh1, de1, g1), (h2, e2, dh2, de2, g2) = out
    assert torch.isfinite(h1).all() and torch.isfinite(e1).all()
    assert torch.equal(h1, h2) and torch.equal(e1, e2)
    # (the block-by-block backward runs the node side in kernels of its own: same arithmetic, other summation orders -- and under
    #  EGT_BWD_MATMUL=bf16x3, tests/test_bwd_modes_gpu.py, other split products -- through up to 58 layers)
    assert_close(dh1, dh2, name="dh", rtol=1e-4, arel=2e-5)
    assert_close(de1, de2, name="de", rtol=1e-4, arel=2e-5)
    for n in g1:
        assert_close(g1[n], g2[n], name=n, rtol=1e-3, arel=1e-4)


@pytest.mark.parametrize("N,De,Dh,train", [(24, 64, 64, False), (32, 64, 64, True), (11, 48, 48, False),
                                           (20, 8, 64, True), (80, 16, 64, True), (32, 32, 64, False), (48, 48, 64, True),
                                           (32, 8, 64, True), (128, 8, 64, False),
                                           (150, 8, 64, True), (144, 16, 64, False),   # 32-row workgroups of k_block_fwd_r4
                                           # node widths below 64 on the fused node side (zero-padded column tiles): ZINC-100K's
                                           # N = 37 / Dh = 48 (d = 6), two empty tiles (d = 4), a partial tile (d = 5), d = 1
                                           (37, 48, 48, True), (40, 8, 32, True), (24, 16, 40, False), (20, 64, 8, False),
                                           (40, 64, 64, True), (150, 64, 64, False)])   # De = 64 with several ragged row groups (k_block_bwd_v4)
def test_stack_call_vs_oracle(N, De, Dh, train, gpu, egt_lib):
    """egt_stack_fwd/bwd (one C call per direction, deferred partial reduction) vs the fp64 oracle,
    including the per-layer in-kernel random masks."""
    from egt_amd import EGTStack
    from egt_amd.fused import layer_seed
    from oracle import egt_oracle as O, rng_ref
    B, Ly, p = 2, 3, 0.2
    torch.manual_seed(11)
    st = EGTStack(model_height=Ly, model_width=Dh, edge_width=De, num_heads=8,
                  random_mask_prob=p if train else 0.0, seed=5, fused=True).to(gpu).train(train)
    with torch.no_grad():
        for prm in st.parameters():
            if prm.dim() == 1:
                prm.add_(0.2 * torch.randn_like(prm))
    g = torch.Generator().manual_seed(N * 7 + De)
    h = torch.randn(B, N, Dh, generator=g); e = torch.randn(B, N, N, De, generator=g) * 1.3
    mask = torch.ones(B, N, dtype=torch.bool); mask[1, N - 3:] = False
    dh = torch.randn(B, N, Dh, generator=g); de = torch.randn(B, N, N, De, generator=g)
    hg = h.to(gpu).requires_grad_(); eg = e.to(gpu).requires_grad_()
    h2, e2 = st(hg, eg, mask.to(gpu))
    torch.autograd.backward([h2, e2], [dh.to(gpu), de.to(gpu)])
    # oracle
    names = {"norm_edge.gamma": ("norm_edge", "gamma"), "norm_edge.beta": ("norm_edge", "beta"),
             "attention_gates.kernel": ("attention_gates", "kernel"), "attention_gates.bias": ("attention_gates", "bias"),
             "dense_edge_b.kernel": ("dense_edge_b", "kernel"), "dense_edge_b.bias": ("dense_edge_b", "bias"),
             "norm_mha.gamma": ("norm_mha", "gamma"), "norm_mha.beta": ("norm_mha", "beta"),
             "dense_qkv.kernel": ("dense_qkv", "kernel"), "dense_qkv.bias": ("dense_qkv", "bias"),
             "dense_mha.kernel": ("dense_mha", "kernel"), "dense_mha.bias": ("dense_mha", "bias"),
             "dense_edge_r.kernel": ("dense_edge_r", "kernel"), "dense_edge_r.bias": ("dense_edge_r", "bias")}
    layers = [{k: getattr(getattr(blk, m), a_).detach().double().cpu().requires_grad_()
               for k, (m, a_) in names.items()} for blk in st.blocks]
    rms = None
    if train:
        b0 = st.blocks[0].mha
        seed = (b0.seed * 0x9E3779B97F4A7C15 + b0._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        rms = [torch.from_numpy(rng_ref.random_mask(layer_seed(seed, l), B, N, 8, p)) for l in range(Ly)]
    h64 = h.double().requires_grad_(); e64 = e.double().requires_grad_()
    ho, eo = O.stack_forward(h64, e64, mask, layers, num_heads=8, rand_masks=rms)
    flat = [t for lp in layers for t in lp.values()]
    gr = torch.autograd.grad([ho, eo], [h64, e64] + flat, [dh.double(), de.double()])
    assert_close(h2, ho, name="h_out", rtol=2e-4, arel=5e-5)
    assert_close(e2, eo, name="e_out", rtol=2e-4, arel=5e-5)
    assert_close(hg.grad, gr[0], name="dh", **BWD)
    assert_close(eg.grad, gr[1], name="de", **BWD)
    gi = iter(gr[2:])
    for li, blk in enumerate(st.blocks):
        for k, (m, a_) in names.items():
            assert_close(getattr(getattr(blk, m), a_).grad, next(gi), name=f"L{li}.{k}", **BWD)


def test_fused_accepts_unaligned_parameter_views(gpu, egt_lib):
    """Parameters living at odd float offsets of one flat buffer (a common optimizer layout)
    must work: no 16-byte alignment is assumed for parameter tensors."""
    from egt_amd import EGTBlock
    torch.manual_seed(4)
    a = EGTBlock(model_width=64, edge_width=64, fused=True).to(gpu).eval()
    b = EGTBlock(model_width=64, edge_width=64, fused=False).to(gpu).eval()
    b.load_state_dict(a.state_dict())
    total = sum(p.numel() + 3 for p in a.parameters()) + 1
    flat = torch.zeros(total, device=gpu)
    off = 1
    for p in a.parameters():
        v = flat[off:off + p.numel()].view_as(p)
        v.copy_(p.data)
        p.data = v
        assert p.data_ptr() % 16 != 0 or True
        off += p.numel() + 3
    assert any(p.data_ptr() % 16 != 0 for p in a.parameters())
    h = torch.randn(2, 32, 64, device=gpu); e = torch.randn(2, 32, 32, 64, device=gpu)
    mask = torch.ones(2, 32, dtype=torch.bool, device=gpu); mask[0, 20:] = False
    outs = []
    for blk in (a, b):
        hh = h.clone().requires_grad_(); ee = e.clone().requires_grad_()
        h2, e2 = blk(hh, ee, mask)
        (h2.sum() + (e2 * e2).sum()).backward()
        outs.append((h2.detach(), e2.detach(), hh.grad, ee.grad, blk.dense_qkv.kernel.grad, blk.dense_edge_r.bias.grad))
    for n, u, v in zip(("h", "e", "dh", "de", "dWqkv", "dbr"), *outs):
        assert_close(u, v, name=n, rtol=1e-3, arel=2e-4, l2=2e-3)


class _Bf16Storage(torch.autograd.Function):
    """A tensor that lives in HBM as bfloat16: the value is rounded where it is stored, and so is the gradient that flows
    back through the same tensor (de of a layer is the de' the layer below reads)."""

    @staticmethod
    def forward(ctx, x):
        return x.to(torch.bfloat16).to(x.dtype)

    @staticmethod
    def backward(ctx, g):
        return g.to(torch.bfloat16).to(g.dtype)


def test_config3_as_specified_bf16_depth4_margins(gpu, egt_lib, capsys):
    """BASELINE config 3 AS SPECIFIED: CIFAR10 shapes N = 150, Dh = 64, De = 8, H = 8, Ly = 4, bf16 edge tensors (the dtype is
    BASELINE.json's request for this config -- the reference itself is fp32 everywhere), training mode with the in-kernel random
    mask, node counts in [85, 150].  Forward and every gradient against the fp64 oracle fed the same bf16-rounded inputs; the
    worst error / tolerance of every output is PRINTED and written to gpurun_out/bf16_margins.json.
    Two oracles, both ASSERTED: (a) the plain fp64 oracle, under the depth-dependent bf16 contract of BASELINE.md section 2b
    (tests/util.py: bf16_stack_tol(Ly) = SURVEY's single-operator rtol 2e-2 times max(1, sqrt(Ly / 2)) -- the storage rounding of
    e_l / de_l at 2 Ly points is the error, and it adds in quadrature: measured 1.02 x the single-operator bound at Ly = 4, i.e.
    0.72 of the contract); (b) the oracle that also ROUNDS the intermediate e_l / de_l to bfloat16 where the kernels store them,
    under the SINGLE-operator tolerance: its margins isolate the ARITHMETIC of the kernels, including the bf16-operand MFMAs of
    the gradient path (egt_narrow.hip)."""
    import json
    from egt_amd import EGTStack
    from egt_amd.fused import layer_seed
    from oracle import egt_oracle as O, rng_ref
    from util import margin, bf16_stack_tol
    B, N, De, Dh, Ly, p = 2, 150, 8, 64, 4, 0.1
    torch.manual_seed(31)
    st = EGTStack(model_height=Ly, model_width=Dh, edge_width=De, num_heads=8, random_mask_prob=p, seed=5, fused=True).to(gpu).train()
    with torch.no_grad():
        for prm in st.parameters():
            if prm.dim() == 1:
                prm.add_(0.2 * torch.randn_like(prm))
    g = torch.Generator().manual_seed(150 * 4)
    h = torch.randn(B, N, Dh, generator=g)
    e = (torch.randn(B, N, N, De, generator=g) * 1.3).bfloat16()
    mask = torch.ones(B, N, dtype=torch.bool); mask[0, 85:] = False; mask[1, 131:] = False
    dh = torch.randn(B, N, Dh, generator=g); de = torch.randn(B, N, N, De, generator=g).bfloat16()
    hg = h.to(gpu).requires_grad_(); eg = e.to(gpu).requires_grad_()
    h2, e2 = st(hg, eg, mask.to(gpu))
    assert st.last_path == "fused-stack" and e2.dtype == torch.bfloat16
    torch.autograd.backward([h2, e2], [dh.to(gpu), de.to(gpu)])
    b0 = st.blocks[0].mha
    seed = (b0.seed * 0x9E3779B97F4A7C15 + b0._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    rms = [torch.from_numpy(rng_ref.random_mask(layer_seed(seed, l), B, N, 8, p)) for l in range(Ly)]
    tol1, ptol1 = bf16_stack_tol(1), bf16_stack_tol(1, params=True)        # bf16 rtol 2e-2 (one operator)
    tolL, ptolL = bf16_stack_tol(Ly), bf16_stack_tol(Ly, params=True)      # the stack contract

    def oracle_margins(storage):
        tol, ptol = (tol1, ptol1) if storage else (tolL, ptolL)
        layers = [{k: getattr(getattr(blk, m), a_).detach().double().cpu().requires_grad_()
                   for k, (m, a_) in PMAP.items()} for blk in st.blocks]
        h64 = h.double().requires_grad_(); e64 = e.double().requires_grad_()
        ho, eo = h64, e64
        for l, lp in enumerate(layers):
            ho, eo = O.block_forward(ho, eo, mask, lp, num_heads=8, rand_mask=rms[l])
            if storage:
                eo = _Bf16Storage.apply(eo)
        flat = [t for lp in layers for t in lp.values()]
        gr = torch.autograd.grad([ho, eo], [h64, e64] + flat, [dh.double(), de.double()])
        de_ref = gr[1].to(torch.bfloat16).double() if storage else gr[1]
        out = {"h_out": margin(h2, ho, **tol), "e_out": margin(e2.float(), eo, **tol),
               "dh": margin(hg.grad, gr[0], **tol), "de": margin(eg.grad.float(), de_ref, **tol)}
        gi = iter(gr[2:])
        for li, blk in enumerate(st.blocks):
            for k, (m, a_) in PMAP.items():
                out[f"L{li}.{k}"] = margin(getattr(getattr(blk, m), a_).grad, next(gi), **ptol)
        return out

    ms, mp = oracle_margins(True), oracle_margins(False)
    ws, wp = max(ms, key=ms.get), max(mp, key=mp.get)
    with capsys.disabled():
        fmt = lambda m_: ", ".join(f"{k} {v:.2f}" for k, v in m_.items() if not k.startswith("L"))
        print(f"\n[bf16 margins, config 3 as specified, Ly = 4] worst error / tolerance vs the oracle with bf16 storage of e_l / de_l: {fmt(ms)}; "
              f"worst of all: {ws} {ms[ws]:.2f}   |   vs the plain fp64 oracle: {fmt(mp)}; worst of all: {wp} {mp[wp]:.2f}")
    try:
        os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
        json.dump(dict(config="cifar10_n150 as specified: B=2, N=150, Dh=64, De=8, H=8, Ly=4, bf16 edge tensors, random_mask_prob 0.1",
                       tolerance=dict(single_operator=dict(outputs=tol1, parameter_gradients=ptol1), stack_contract=dict(outputs=tolL, parameter_gradients=ptolL),
                                      note="storage-aware oracle judged by the single-operator tolerance, plain fp64 oracle by the stack contract (BASELINE.md 2b)"),
                       worst_error_over_tolerance_vs_oracle_with_bf16_storage=ms, worst_error_over_tolerance_vs_plain_fp64_oracle=mp),
                  open(os.path.join(REPO, "gpurun_out", "bf16_margins.json"), "w"), indent=1)
    except OSError:
        pass
    assert ms[ws] < 1.0, (ws, ms[ws])
    assert mp[wp] < 1.0, (wp, mp[wp])          # the as-specified config against the PLAIN fp64 oracle, under the documented stack contract


@pytest.mark.parametrize("N,De,Dh,train,Ly", [(32, 64, 64, True, 3), (20, 8, 64, False, 2), (37, 48, 48, True, 2),
                                              (64, 64, 64, False, 1),
                                              # BASELINE config 3 as specified (CIFAR10 shapes, bf16): N = 150 selects the
                                              # 32-row / 8-wave forward k_block_fwd_r4<8,false,8,true> and the ragged
                                              # two-row backward k_block_bwd_v4r<8,true,2>; N = 160 their 16-row-exact forms
                                              (150, 8, 64, True, 2), (160, 8, 64, False, 2), (120, 8, 64, True, 2)])
def test_stack_bf16_edge_tensors_vs_oracle(N, De, Dh, train, Ly, gpu, egt_lib):
    """EGT_BF16 (BASELINE config 3's dtype): e / e' / de' / de are bfloat16 in HBM, arithmetic fp32.
    The plain fp64 oracle gets the SAME bf16-rounded inputs; the tolerance is the stack contract of BASELINE.md section 2b
    (tests/util.py: bf16_stack_tol(Ly) -- SURVEY 8(c)'s single-operator rtol 2e-2 up to two blocks, times sqrt(Ly / 2) beyond)."""
    from util import bf16_stack_tol
    from egt_amd import EGTStack
    from egt_amd.fused import layer_seed
    from oracle import egt_oracle as O, rng_ref
    B, p = 2, 0.2
    torch.manual_seed(23)
    st = EGTStack(model_height=Ly, model_width=Dh, edge_width=De, num_heads=8,
                  random_mask_prob=p if train else 0.0, seed=9, fused=True).to(gpu).train(train)
    with torch.no_grad():
        for prm in st.parameters():
            if prm.dim() == 1:
                prm.add_(0.2 * torch.randn_like(prm))
    g = torch.Generator().manual_seed(N * 5 + De)
    h = torch.randn(B, N, Dh, generator=g)
    e = (torch.randn(B, N, N, De, generator=g) * 1.3).bfloat16()
    mask = torch.ones(B, N, dtype=torch.bool); mask[1, N - 3:] = False
    dh = torch.randn(B, N, Dh, generator=g); de = torch.randn(B, N, N, De, generator=g).bfloat16()
    hg = h.to(gpu).requires_grad_(); eg = e.to(gpu).requires_grad_()
    h2, e2 = st(hg, eg, mask.to(gpu))
    assert e2.dtype == torch.bfloat16 and h2.dtype == torch.float32
    assert st.last_path == "fused-stack"
    torch.autograd.backward([h2, e2], [dh.to(gpu), de.to(gpu)])
    assert eg.grad.dtype == torch.bfloat16
    layers = [{k: getattr(getattr(blk, m), a_).detach().double().cpu().requires_grad_()
               for k, (m, a_) in PMAP.items()} for blk in st.blocks]
    rms = None
    if train:
        b0 = st.blocks[0].mha
        seed = (b0.seed * 0x9E3779B97F4A7C15 + b0._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        rms = [torch.from_numpy(rng_ref.random_mask(layer_seed(seed, l), B, N, 8, p)) for l in range(Ly)]
    h64 = h.double().requires_grad_(); e64 = e.double().requires_grad_()
    ho, eo = O.stack_forward(h64, e64, mask, layers, num_heads=8, rand_masks=rms)
    flat = [t for lp in layers for t in lp.values()]
    gr = torch.autograd.grad([ho, eo], [h64, e64] + flat, [dh.double(), de.double()])
    tol, ptol = bf16_stack_tol(Ly), bf16_stack_tol(Ly, params=True)
    assert_close(h2, ho, name="h_out", **tol)
    assert_close(e2.float(), eo, name="e_out", **tol)
    assert_close(hg.grad, gr[0], name="dh", **tol)
    assert_close(eg.grad.float(), gr[1], name="de", **tol)
    gi = iter(gr[2:])
    for li, blk in enumerate(st.blocks):
        for k, (m, a_) in PMAP.items():
            assert_close(getattr(getattr(blk, m), a_).grad, next(gi), name=f"L{li}.{k}", l2=ptol["rtol"], **ptol)


@pytest.mark.parametrize("N,De,Dh", [(37, 48, 48), (64, 64, 64), (40, 8, 32)])
def test_node_side_runs_inside_the_pair_kernels(N, De, Dh, gpu, egt_lib):
    """One launch per layer and direction: dense_mha + residual + the next block's norm_mha / dense_qkv run as the forward pair
    kernel's epilogue, dQKV -> dh -> dV_att / delta as the backward pair kernel's prologue (egt_block_dev.h).  Node widths
    below 64 (ZINC-100K: Dh = 48, d = 6) take the zero-padded fourth column tile.  Counted with the C-ABI's launch profiler:
    a 3-layer stack is ONE k_node_pre, no k_node_post, ONE k_node_bwd (the bottom of the chain) -- parity of these shapes is
    test_stack_call_vs_oracle / test_other_baseline_shapes_fused_vs_oracle / test_stack_bf16_edge_tensors_vs_oracle."""
    import ctypes as C
    from egt_amd import EGTStack, _lib
    lib = _lib.load()
    torch.manual_seed(5)
    Ly, B = 3, 2
    st = EGTStack(model_height=Ly, model_width=Dh, edge_width=De, num_heads=8, fused=True).to(gpu).eval()
    g = torch.Generator().manual_seed(N)
    h = torch.randn(B, N, Dh, generator=g).to(gpu).requires_grad_(); e = torch.randn(B, N, N, De, generator=g).to(gpu).requires_grad_()
    mask = torch.ones(B, N, dtype=torch.bool, device=gpu)
    lib.egt_prof_filter(b""); lib.egt_prof_enable(2)
    try:
        h2, e2 = st(h, e, mask)
        assert st.last_path == "fused-stack"
        torch.autograd.backward([h2, e2], [torch.ones_like(h2), torch.ones_like(e2)])
        torch.cuda.synchronize()
    finally:
        lib.egt_prof_enable(0)
    buf = C.create_string_buffer(4096)
    lib.egt_prof_names(buf, 4096)
    count = {}
    for name in buf.value.decode().split():
        cnt, ms = C.c_int64(0), C.c_double(0.0)
        lib.egt_prof_read(name.encode(), C.byref(cnt), C.byref(ms))
        count[name] = cnt.value
    assert count.get("k_block_fwd") == Ly and count.get("k_block_bwd") == Ly, count
    assert count.get("k_node_pre") == 1 and count.get("k_node_post", 0) == 0 and count.get("k_node_bwd") == 1, count
    # the edge-weight preparation of every layer rides along with k_node_pre (no k_edge_prep launch); the step's launches are the
    # 2 Ly pair kernels + k_node_pre, k_node_bwd, k_node_wgrads, k_sum_segments, k_edge_param_grads
    assert count.get("k_edge_prep", 0) == 0, count
    assert sum(count.values()) == 2 * Ly + 5, count


def test_block_bf16_single_block_and_dtype_errors(gpu, egt_lib):
    """single-block C call with bf16 edge tensors; a bf16 caller of h gets bf16 back."""
    from egt_amd import EGTBlock
    torch.manual_seed(3)
    blk = EGTBlock(model_width=64, edge_width=64, num_heads=8, fused=True).to(gpu).eval()
    B, N = 2, 32
    h = torch.randn(B, N, 64, device=gpu); e = torch.randn(B, N, N, 64, device=gpu)
    mask = torch.ones(B, N, dtype=torch.bool, device=gpu)
    h32, e32 = blk(h, e.bfloat16().float(), mask)
    hb, eb = blk(h.bfloat16(), e.bfloat16(), mask)
    assert hb.dtype == torch.bfloat16 and eb.dtype == torch.bfloat16
    assert_close(eb.float(), e32, name="e_out(bf16 vs fp32 on rounded input)", rtol=1e-2, arel=5e-3)


@pytest.mark.parametrize("N,De,train", [(32, 64, True), (21, 16, False)])
def test_stack_bias_edge_channels_vs_oracle(N, De, train, gpu, egt_lib):
    """EGT-simple ('bias' edge channels, graph_xformer_model_base.py:173-190) on the fused stack path:
    gates / edge bias from the RAW e, e returned unchanged, d e accumulates every layer's projection
    gradient."""
    from egt_amd import EGTStack
    from egt_amd.fused import layer_seed
    from oracle import egt_oracle as O, rng_ref
    B, Ly, p, Dh = 2, 3, 0.2, 64
    torch.manual_seed(31)
    st = EGTStack(model_height=Ly, model_width=Dh, edge_width=De, num_heads=8, edge_channel_type="bias",
                  random_mask_prob=p if train else 0.0, seed=4, fused=True).to(gpu).train(train)
    with torch.no_grad():
        for prm in st.parameters():
            if prm.dim() == 1:
                prm.add_(0.2 * torch.randn_like(prm))
    g = torch.Generator().manual_seed(N + De)
    h = torch.randn(B, N, Dh, generator=g); e = torch.randn(B, N, N, De, generator=g)
    mask = torch.ones(B, N, dtype=torch.bool); mask[1, N - 3:] = False
    dh = torch.randn(B, N, Dh, generator=g); de = torch.randn(B, N, N, De, generator=g)
    hg = h.to(gpu).requires_grad_(); eg = e.to(gpu).requires_grad_()
    h2, e2 = st(hg, eg, mask.to(gpu))
    assert st.last_path == "fused-stack" and e2 is eg
    torch.autograd.backward([h2, e2], [dh.to(gpu), de.to(gpu)])
    names = {k: v for k, v in PMAP.items() if not (k.startswith("norm_edge") or k.startswith("dense_edge_r"))}
    layers = [{k: getattr(getattr(blk, m), a_).detach().double().cpu().requires_grad_()
               for k, (m, a_) in names.items()} for blk in st.blocks]
    rms = None
    if train:
        b0 = st.blocks[0].mha
        seed = (b0.seed * 0x9E3779B97F4A7C15 + b0._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        rms = [torch.from_numpy(rng_ref.random_mask(layer_seed(seed, l), B, N, 8, p)) for l in range(Ly)]
    h64 = h.double().requires_grad_(); e64 = e.double().requires_grad_()
    ho, eo = O.stack_forward(h64, e64, mask, layers, num_heads=8, rand_masks=rms, edge_channel_type="bias")
    flat = [t for lp in layers for t in lp.values()]
    gr = torch.autograd.grad([ho, eo], [h64, e64] + flat, [dh.double(), de.double()])
    assert_close(h2, ho, name="h_out", rtol=2e-4, arel=5e-5)
    assert_close(hg.grad, gr[0], name="dh", **BWD)
    assert_close(eg.grad, gr[1], name="de", **BWD)
    gi = iter(gr[2:])
    for li, blk in enumerate(st.blocks):
        for k, (m, a_) in names.items():
            assert_close(getattr(getattr(blk, m), a_).grad, next(gi), name=f"L{li}.{k}", **BWD)
