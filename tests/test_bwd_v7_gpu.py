"""k_block_bwd_v7 (twelve waves per CU, 32-row workgroups: egt_amd/csrc/egt_block_bwd7.h) against the fp64 oracle and against
k_block_bwd_v5.  (With EGT_BWD_V7 = 2 it runs where a launch fills the chip -- B * N / 32 >= CUs: the headline batch, covered
by tests/test_fullsize_gpu.py); it is OPT-IN -- EGT_BWD_V7 = 1 (every geometry it covers) / 2 (launches that fill the chip), read once
per process -- so the cases run in subprocesses with the variable set."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

_CASE = r'''
import json, sys, torch
sys.path.insert(0, "{repo}"); sys.path.insert(0, "{repo}/tests")
import ctypes as C
from egt_amd import EGTStack, _lib as L
lib = L.load()
B, N, Ly, train = {B}, {N}, {Ly}, {train}
d = L.BlockDesc(B=B, N=N, H=8, d=8, De=64, dtype=L.EGT_F32, flags=L.BF_GATE | L.BF_CLIP | (L.BF_TRAINING if train else 0),
                clip_lo=-5.0, clip_hi=5.0, random_mask_prob=0.2 if train else 0.0, ln_eps=1e-3, reserved=0, seed=0, seed_device=None)
kern = lib.egt_block_bwd_kernel(C.byref(d)).decode()
dev = torch.device("cuda:0")
torch.manual_seed(11)
st = EGTStack(model_height=Ly, model_width=64, edge_width=64, num_heads=8, random_mask_prob=0.2 if train else 0.0, seed=5, fused=True).to(dev).train(bool(train))
with torch.no_grad():
    for prm in st.parameters():
        if prm.dim() == 1:
            prm.add_(0.2 * torch.randn_like(prm))
g = torch.Generator().manual_seed(N * 7 + 64)
h = torch.randn(B, N, 64, generator=g); e = torch.randn(B, N, N, 64, generator=g) * 1.3
mask = torch.ones(B, N, dtype=torch.bool); mask[1, N - 3:] = False
if B > 2:
    mask[2, 5:] = False
dh = torch.randn(B, N, 64, generator=g); de = torch.randn(B, N, N, 64, generator=g)
hg = h.to(dev).requires_grad_(); eg = e.to(dev).requires_grad_()
h2, e2 = st(hg, eg, mask.to(dev))
torch.autograd.backward([h2, e2], [dh.to(dev), de.to(dev)])
out = dict(kernel=kern, h_out=h2.detach().cpu(), e_out=e2.detach().cpu(), dh=hg.grad.cpu(), de=eg.grad.cpu(),
           grads={{n: p.grad.cpu() for n, p in st.named_parameters()}}, params={{n: p.detach().cpu() for n, p in st.named_parameters()}},
           seed=(st.blocks[0].mha.seed, st.blocks[0].mha._calls), inputs=(h, e, mask, dh, de))
torch.save(out, "{out}")
'''


def _run_case(tmp_path, v7, B, N, Ly, train):
    out = tmp_path / f"v7_{v7}_{B}_{N}_{Ly}_{int(train)}.pt"
    code = _CASE.format(repo=REPO, B=B, N=N, Ly=Ly, train=int(train), out=out)
    env = dict(os.environ, EGT_BWD_V7=str(v7))
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return torch.load(out, weights_only=False)


@pytest.mark.parametrize("B,N,Ly,train", [(2, 64, 3, True), (3, 32, 2, True), (2, 64, 1, False)])
def test_v7_stack_vs_oracle_and_v5(B, N, Ly, train, gpu, egt_lib, tmp_path):
    """forced v7 == the fp64 oracle (the tolerances of tests/test_block_gpu.py::test_stack_call_vs_oracle) and == forced v5 to a few
    ulps (same per-step arithmetic; the sums over row chunks associate differently)"""
    from util import assert_close, BWD
    from egt_amd.fused import layer_seed
    from oracle import egt_oracle as O, rng_ref
    a = _run_case(tmp_path, 1, B, N, Ly, train)
    b = _run_case(tmp_path, 0, B, N, Ly, train)
    assert a["kernel"] == "k_block_bwd_v7" and b["kernel"] == "k_block_bwd_v5"
    # v7 vs v5
    for k in ("h_out", "e_out"):
        assert torch.equal(a[k], b[k])                       # same forward
    assert_close(a["de"], b["de"], name="de v7 vs v5", rtol=1e-5, arel=1e-6)
    assert_close(a["dh"], b["dh"], name="dh v7 vs v5", rtol=1e-4, arel=1e-5)
    for n in a["grads"]:
        assert_close(a["grads"][n], b["grads"][n], name=f"{n} v7 vs v5", rtol=1e-4, arel=2e-5)
    # v7 vs the oracle
    h, e, mask, dh, de = a["inputs"]
    names = ["norm_edge.gamma", "norm_edge.beta", "attention_gates.kernel", "attention_gates.bias", "dense_edge_b.kernel", "dense_edge_b.bias",
             "norm_mha.gamma", "norm_mha.beta", "dense_qkv.kernel", "dense_qkv.bias", "dense_mha.kernel", "dense_mha.bias",
             "dense_edge_r.kernel", "dense_edge_r.bias"]
    layers = [{k: a["params"][f"blocks.{li}.{k}"].double().requires_grad_() for k in names} for li in range(Ly)]
    rms = None
    if train:
        s0, calls = a["seed"]
        seed = (s0 * 0x9E3779B97F4A7C15 + calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        rms = [torch.from_numpy(rng_ref.random_mask(layer_seed(seed, l), B, N, 8, 0.2)) for l in range(Ly)]
    h64 = h.double().requires_grad_(); e64 = e.double().requires_grad_()
    ho, eo = O.stack_forward(h64, e64, mask, layers, num_heads=8, rand_masks=rms)
    flat = [t for lp in layers for t in lp.values()]
    gr = torch.autograd.grad([ho, eo], [h64, e64] + flat, [dh.double(), de.double()])
    assert_close(a["h_out"], ho, name="h_out", rtol=2e-4, arel=5e-5)
    assert_close(a["e_out"], eo, name="e_out", rtol=2e-4, arel=5e-5)
    assert_close(a["dh"], gr[0], name="dh", **BWD)
    assert_close(a["de"], gr[1], name="de", **BWD)
    gi = iter(gr[2:])
    for li in range(Ly):
        for k in names:
            assert_close(a["grads"][f"blocks.{li}.{k}"], next(gi), name=f"L{li}.{k}", **BWD)


def test_v7_is_opt_in(gpu, egt_lib):
    """k_block_bwd_v7 measured slower than k_block_bwd_v5 at the headline batch: without EGT_BWD_V7 the dispatch never takes it"""
    from egt_amd import _lib as L
    if os.environ.get("EGT_BWD_V7"):
        pytest.skip("EGT_BWD_V7 is set: the default dispatch is not what runs")
    mk = lambda B, N, De=64, dt=L.EGT_F32, fl=0: L.BlockDesc(B=B, N=N, H=8, d=8, De=De, dtype=dt, flags=L.BF_GATE | fl, clip_lo=-5.0, clip_hi=5.0,
                                                             random_mask_prob=0.0, ln_eps=1e-3, reserved=0, seed=0, seed_device=None)
    k = lambda d: egt_lib.egt_block_bwd_kernel(C.byref(d)).decode()
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    assert k(mk(cus // 2, 64)) == "k_block_bwd_v5"          # the headline batch (B = 128 on MI355X)
    assert k(mk(2, 64)) == "k_block_bwd_v5"
    assert k(mk(cus // 2, 64, De=48)) == "k_block_bwd_v5"
    assert k(mk(cus // 2, 64, dt=L.EGT_BF16)) == "k_block_bwd_v4"
    assert k(mk(cus // 2, 64, fl=L.BF_ATTN_MASK)) == "k_block_bwd_v4"
    assert k(mk(cus // 2, 64, De=8)) == "k_narrow_bwd"


def test_v7_full_size_properties(gpu, egt_lib):
    """the full-size property tests (fused == composed, linearity, bit-reproducibility, padded-key invariance at B = 128, N = 64) on v7"""
    env = dict(os.environ, EGT_BWD_V7="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_fullsize_gpu.py"), "-m", "gpu", "-x", "-q", "-k", "zinc500k"],
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "passed" in r.stdout
