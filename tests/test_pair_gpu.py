"""The fused pair operator of the large-head geometry (egt_pair_fwd / egt_pair_bwd: d = 64, De = 32 -- BASELINE config 5):
(h, e, mask) -> (h', e') of an EGTBlock that runs k_pair_fwd / k_pair_bwd (E, G, H_hat, dE, dG, dH_ext never in HBM) against
the fp64 oracle (graph_xformer_model_base.py:106-145,192-223 with egt_layers.py:57-143 inside) and against the composed HIP path."""
import ctypes as C

import pytest
import torch

import cases as CS
from util import assert_close, FWD, BWD
from test_block_gpu import build_block, PMAP

pytestmark = pytest.mark.gpu

ATTRS = dict(gate_attention=True, edge_activation=None, edge_channel_type="residual")


def _case(B, N, nodes, seed, rand_p=None):
    from oracle import egt_oracle as O
    g = torch.Generator().manual_seed(seed)
    Dh, De = 512, 32
    params = O.init_block_params(Dh, De, 8, generator=g, randomize_norm=True)
    h = torch.randn(B, N, Dh, generator=g); e = torch.randn(B, N, N, De, generator=g) * 1.2 + 0.2
    dh = torch.randn(B, N, Dh, generator=g); de = torch.randn(B, N, N, De, generator=g)
    mask = torch.zeros(B, N, dtype=torch.bool)
    for b, n in enumerate(nodes):
        mask[b, :n] = True
    return dict(h=h, e=e, mask=mask, attn_mask=None, rand_mask=None, dh=dh, de=de), params, dict(Dh=Dh, De=De, rand_p=rand_p)


def _names(lib):
    buf = C.create_string_buffer(4096)
    lib.egt_prof_names(buf, 4096)
    return buf.value.decode().split()


def _run(blk, inp, gpu):
    hg = inp["h"].to(gpu).requires_grad_(); eg = inp["e"].to(gpu).requires_grad_()
    for p in blk.parameters():
        p.grad = None
    h2, e2 = blk(hg, eg, inp["mask"].to(gpu))
    torch.autograd.backward([h2, e2], [inp["dh"].to(gpu), inp["de"].to(gpu)])
    return h2.detach(), e2.detach(), hg.grad, eg.grad


def _compare(blk, out, ref):
    h2, e2, dh, de = out
    assert_close(h2, ref["h_out"], name="h_out", **FWD)
    assert_close(e2, ref["e_out"], name="e_out", **FWD)
    assert_close(dh, ref["dh"], name="dh", **BWD)
    assert_close(de, ref["de"], name="de", **BWD)
    for k, (m, a) in PMAP.items():
        assert_close(getattr(getattr(blk, m), a).grad, ref["dparams"][k], name=k, **BWD)


@pytest.mark.parametrize("B,N,nodes", [(2, 48, [48, 31]), (1, 37, [29]), (2, 16, [16, 9]), (1, 100, [100]),
                                       (3, 24, [24, 0, 7]), (1, 8, [5]), (5, 32, [32, 17, 1, 32, 20])])
def test_pair_block_vs_oracle(B, N, nodes, gpu, egt_lib):
    """key padding inside and across key tiles, N not a multiple of 16 (ragged last tile), a single tile, several tiles; a graph without
    any node (every key masked: the reference's all-masked rows, egt_layers.py:91-108), a tile smaller than the MFMA tile, workgroup counts
    that are not a multiple of the 8 XCDs (no remap)"""
    inp, params, c = _case(B, N, nodes, seed=100 + N)
    blk = build_block(c, ATTRS, params, gpu, "auto").eval()
    egt_lib.egt_prof_filter(b""); egt_lib.egt_prof_enable(2)
    try:
        out = _run(blk, inp, gpu)
        torch.cuda.synchronize()
    finally:
        egt_lib.egt_prof_enable(0)
    assert blk.last_path == "fused-pair"
    names = _names(egt_lib)
    assert "k_pair_fwd" in names and "k_pair_bwd" in names and "k_attn_mfma_bwd_q" in names, names
    composed = {"k_edge_proj_fwd", "k_edge_proj_bwd", "k_edge_update_fwd", "k_edge_update_bwd", "k_attn_mfma_fwd", "k_attn_mfma_bwd_kv"}
    assert not composed & set(names), names   # none of the composed path's [B,N,N,8]-producing launches
    ref = CS.block_oracle(inp, params, dict(num_heads=8, **ATTRS))
    _compare(blk, out, ref)


def test_pair_block_equals_composed_and_is_deterministic(gpu, egt_lib):
    inp, params, c = _case(2, 64, [64, 50], seed=7)
    a = build_block(c, ATTRS, params, gpu, "auto").eval()
    b = build_block(c, ATTRS, params, gpu, False).eval()
    ra, ra2, rb = _run(a, inp, gpu), None, None
    ga = {k: getattr(getattr(a, m), n).grad.clone() for k, (m, n) in PMAP.items()}
    ra2 = _run(a, inp, gpu)
    for u, v in zip(ra, ra2):
        assert torch.equal(u, v), "the fused pair operator must be bit-reproducible"
    for k, (m, n) in PMAP.items():
        assert torch.equal(ga[k], getattr(getattr(a, m), n).grad), k
    rb = _run(b, inp, gpu)
    assert a.last_path == "fused-pair" and b.last_path == "composed"
    for n, u, v in zip(("h_out", "e_out"), ra[:2], rb[:2]):
        assert_close(u, v, name=n, rtol=1e-4, arel=5e-5)
    for n, u, v in zip(("dh", "de"), ra[2:], rb[2:]):
        assert_close(u, v, name=n, rtol=1e-3, arel=2e-4, l2=2e-3)
    for k, (m, n) in PMAP.items():
        assert_close(ga[k], getattr(getattr(b, m), n).grad, name=k, rtol=1e-3, arel=2e-4, l2=2e-3)


def test_pair_block_in_kernel_random_mask_vs_oracle(gpu, egt_lib):
    """training mode: the in-kernel counter-hash mask (forward and backward draw the same sample) == the oracle fed the
    rng_ref replica of that stream"""
    from oracle import rng_ref
    p = 0.15
    inp, params, c = _case(2, 48, [48, 40], seed=11, rand_p=p)
    blk = build_block(c, ATTRS, params, gpu, "auto")
    blk.mha.random_mask_prob = p
    blk.train()
    out = _run(blk, inp, gpu)
    assert blk.last_path == "fused-pair"
    m = blk.mha
    seed = (m.seed * 0x9E3779B97F4A7C15 + m._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    inp2 = dict(inp, rand_mask=torch.from_numpy(rng_ref.random_mask(seed, 2, 48, 8, p)))
    ref = CS.block_oracle(inp2, params, dict(num_heads=8, **ATTRS))
    _compare(blk, out, ref)


def test_pair_no_clip_and_refusals(gpu, egt_lib):
    """clip_logits_value=None runs the same kernels (run-time switch); geometries outside the instantiated one report unsupported"""
    from egt_amd import EGTBlock, _lib as L
    from oracle import egt_oracle as O
    inp, params, c = _case(1, 32, [27], seed=3)
    blk = EGTBlock(model_width=512, edge_width=32, num_heads=8, clip_logits_value=None, fused="auto").to(gpu).eval()
    with torch.no_grad():
        for k, (m, a) in PMAP.items():
            getattr(getattr(blk, m), a).copy_(params[k].to(gpu))
    out = _run(blk, inp, gpu)
    assert blk.last_path == "fused-pair"
    ref = CS.block_oracle(inp, params, dict(num_heads=8, clip_logits_value=None, **ATTRS))
    _compare(blk, out, ref)
    d = L.BlockDesc(B=1, N=32, H=8, d=64, De=32, dtype=L.EGT_F32, flags=L.BF_GATE | L.BF_CLIP, clip_lo=-5, clip_hi=5,
                    random_mask_prob=0.0, ln_eps=1e-3, reserved=0, seed=0, seed_device=None)
    assert egt_lib.egt_pair_supported(C.byref(d)) == 1
    for field, val in (("d", 32), ("De", 64), ("H", 4), ("flags", L.BF_CLIP), ("flags", L.BF_GATE | L.BF_ATTN_MASK), ("dtype", L.EGT_BF16)):
        d2 = L.BlockDesc.from_buffer_copy(d)
        setattr(d2, field, val)
        assert egt_lib.egt_pair_supported(C.byref(d2)) == 0, (field, val)
        assert egt_lib.egt_pair_workspace_bytes(C.byref(d2)) == 0


def test_random_pair_geometries():
    """tools/sweep_pair.py: random batch / N / node counts / clip / training-mode geometries against the fp64 oracle (a short run;
    the tool takes a case count and EGT_SWEEP_SEED for longer ones -- profiles/r06_sweep_pair.log: 40 cases)."""
    import os, subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "tools", "sweep_pair.py"), "12"], cwd=repo, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "failures: 0" in r.stdout
