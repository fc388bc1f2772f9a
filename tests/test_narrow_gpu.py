"""De = 8 pair kernels (egt_narrow.hip): the driver's GPU suite runs the DEFAULT selection (k_narrow_fwd / k_narrow_bwd for
fp32 and bf16 edge tensors).  The kernel choice is read from the environment once per process, so the other selections
are exercised in child processes: the MFMA-tile backward (v4r) under the De = 8 forward, and the MFMA-tile kernels
(r4 / v4r) with both De = 8 kernels switched off -- all against the same oracle tests."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DE8 = ["tests/test_fullsize_gpu.py::test_other_baseline_shapes_fused_vs_oracle",
       "tests/test_block_gpu.py::test_stack_bf16_edge_tensors_vs_oracle",
       "tests/test_block_gpu.py::test_stack_call_vs_oracle",
       "tests/test_block_gpu.py::test_block_fused_vs_oracle",
       "tests/test_block_gpu.py::test_fused_in_kernel_random_mask"]


def _run(env_extra):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + DE8,
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout


def test_mfma_tile_backward_under_the_narrow_forward():
    _run({"EGT_NO_NARROW_BWD": "1"})


def test_mfma_tile_kernels_with_the_valu_kernels_off():
    _run({"EGT_NO_NARROW": "1"})


def test_random_de8_stacks_under_every_switch():
    """tools/sweep_de8.py: random N / batch / variant / edge dtype / depth under four kernel-selection settings (a short run;
    the tool takes a case count and EGT_SWEEP_SEED for longer ones)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "sweep_de8.py"), "4"], cwd=REPO, env=env, capture_output=True,
                       text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "settings with failures: 0" in r.stdout


@pytest.mark.parametrize("waves", ["4", "8"])
def test_forward_waves_per_workgroup(waves):
    """k_narrow_fwd splits a workgroup's key range over 4 waves, or over 8 when the launch has at most one workgroup per CU
    (egt_narrow_launch_fwd).  The tiny test batches take 8 by default from N = 64 up; both sizes are forced here for every
    geometry (8 waves on N = 37 leaves waves with an empty key range)."""
    _run({"EGT_NRW_FWD_WAVES": waves, "EGT_NRW_FWD_HALF": "0"})


def test_forward_half_row_workgroups():
    """With eight waves and a 16-row grid of at most half the CUs, k_narrow_fwd takes EIGHT query rows per workgroup (lanes p and p ^ 8
    share a row and split a step's key pair; egt_narrow_launch_fwd).  The tiny test batches take that form by default from N = 64 up
    (the run above switches it off); here it is forced together with the eight waves for every geometry, N = 37 and ragged N included."""
    _run({"EGT_NRW_FWD_WAVES": "8", "EGT_NRW_FWD_HALF": "1"})


@pytest.mark.parametrize("waves", ["4", "8"])
def test_backward_waves_per_workgroup(waves):
    """k_narrow_bwd splits a workgroup's key tiles over 4 waves, or over 8 when the launch has at most one workgroup per CU
    (egt_narrow_launch_bwd; the node-side prologue stays four waves' work).  The tiny test batches take 8 by default from N = 64
    up; both sizes are forced here for every geometry (balanced (tile, row) ranges with parked partials, waves without a key tile)."""
    _run({"EGT_NRW_BWD_WAVES": waves})
    _run({"EGT_NRW_BWD_WAVES": waves, "EGT_BWD_TL": "16"})


@pytest.mark.parametrize("rows", ["16", "4"])
def test_backward_rows_per_workgroup(rows):
    """The De = 8 backward takes 16, 8 or 4 query rows per workgroup (egt_block.hip: bwd_rows_per_wg; small batches get 8 so
    that every CU has work).  The test batches are tiny, so the default selection already runs 8 rows per workgroup
    everywhere; the other two sizes are forced here, for the De = 8 kernel and for the MFMA-tile fallback."""
    _run({"EGT_BWD_TL": rows})
    _run({"EGT_BWD_TL": rows, "EGT_NO_NARROW_BWD": "1"})


WIDE = ["tests/test_block_gpu.py::test_stack_call_vs_oracle", "tests/test_block_gpu.py::test_block_fused_vs_oracle",
        "tests/test_block_gpu.py::test_stack_bf16_edge_tensors_vs_oracle", "tests/test_block_gpu.py::test_fused_in_kernel_random_mask",
        "tests/test_graph_gpu.py"]


@pytest.mark.parametrize("rows", ["16", "5"])
def test_backward_rows_per_workgroup_every_edge_width(rows):
    """Every backward pair kernel takes 4 .. 16 query rows per workgroup; the tiny test batches select 8-row groups by default
    (bwd_rows_per_wg: a launch that leaves workgroup slots empty takes shorter groups), so the 16-row geometry of the full-size
    launches (the headline's) and an odd size are forced here for the De >= 16 kernels too."""
    env = dict(os.environ, EGT_BWD_TL=rows)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + WIDE,
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout


@pytest.mark.parametrize("rows", ["16", "13", "4"])
def test_forward_rows_per_workgroup(rows):
    """k_block_fwd takes 4 .. 16 query rows per workgroup (launch_fwd: the tiny test batches select shorter groups than the
    full-size launches, which take 16); forced here: the headline's 16, an odd size, and one row per wave."""
    env = dict(os.environ, EGT_FWD_ROWS=rows)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"] + WIDE,
                       cwd=REPO, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.stdout + r.stderr)[-4000:]
    assert " passed" in r.stdout


# ---- the less-travelled branches of the De = 8 kernels, in process (default selection) -----------------------------
import torch  # noqa: E402

from util import assert_close, BWD  # noqa: E402

PMAP = {"norm_edge.gamma": ("norm_edge", "gamma"), "norm_edge.beta": ("norm_edge", "beta"),
        "attention_gates.kernel": ("attention_gates", "kernel"), "attention_gates.bias": ("attention_gates", "bias"),
        "dense_edge_b.kernel": ("dense_edge_b", "kernel"), "dense_edge_b.bias": ("dense_edge_b", "bias"),
        "norm_mha.gamma": ("norm_mha", "gamma"), "norm_mha.beta": ("norm_mha", "beta"),
        "dense_qkv.kernel": ("dense_qkv", "kernel"), "dense_qkv.bias": ("dense_qkv", "bias"),
        "dense_mha.kernel": ("dense_mha", "kernel"), "dense_mha.bias": ("dense_mha", "bias"),
        "dense_edge_r.kernel": ("dense_edge_r", "kernel"), "dense_edge_r.bias": ("dense_edge_r", "bias")}


@pytest.mark.parametrize("variant,N,Dh,bf16,train", [
    ("ungated", 37, 64, False, True),        # call_ungated: no gate input (run-time feature instance of the kernels)
    ("ungated", 52, 64, True, False),
    ("noclip", 41, 64, False, True),         # clip_logits_value = None
    ("bias", 40, 64, False, True),           # EGT-simple: projections of the raw e (no norm_edge), e returned unchanged
    ("bias", 33, 64, True, False),
    ("d6", 37, 48, False, True),             # Dh = 48 -> d = 6: zero-padded packed QKV, node side on the zero-padded fourth column tile
    ("d6", 50, 48, True, True),
    ("plain", 188, 64, True, True),          # PATTERN's longest graphs: 12 row groups, ragged last key block
    ("plain", 90, 64, True, True),           # 6 key tiles over 4 waves: balanced (tile, row) ranges, key tiles 1 and 4 shared by two waves
    ("plain", 70, 64, True, False),          # 5 key tiles, ragged last row group (6 rows): every range boundary inside a tile
    ("plain", 90, 64, False, True),          # the same geometries with fp32 edge tensors
    ("plain", 70, 64, False, False),
    ("plain", 7, 64, False, False)])         # fewer keys than one wave's share: empty key ranges in the forward
def test_narrow_kernel_branches_vs_oracle(variant, N, Dh, bf16, train, gpu, egt_lib):
    check_de8_stack(variant, N, Dh, bf16, train, gpu)


def check_de8_stack(variant, N, Dh, bf16, train, gpu, B=2, Ly=2):
    """One De = 8 stack (forward + backward, every input and parameter gradient) against the fp64 oracle; also the body of
    tools/sweep_de8.py's random geometries."""
    from egt_amd import EGTStack
    from egt_amd.fused import layer_seed
    from oracle import egt_oracle as O, rng_ref
    p, De = 0.2, 8
    kw, okw = {}, {}
    if variant == "ungated":
        kw["gate_attention"] = False; okw["gate_attention"] = False
    if variant == "noclip":
        kw["clip_logits_value"] = None; okw["clip_logits_value"] = None
    if variant == "bias":
        kw["edge_channel_type"] = "bias"; okw["edge_channel_type"] = "bias"
    torch.manual_seed(101 + N)
    st = EGTStack(model_height=Ly, model_width=Dh, edge_width=De, num_heads=8, random_mask_prob=p if train else 0.0,
                  seed=3, fused=True, **kw).to(gpu).train(train)
    with torch.no_grad():
        for prm in st.parameters():
            if prm.dim() == 1:
                prm.add_(0.2 * torch.randn_like(prm))
    g = torch.Generator().manual_seed(N * 3 + Dh)
    h = torch.randn(B, N, Dh, generator=g); e = torch.randn(B, N, N, De, generator=g) * 1.3
    dh = torch.randn(B, N, Dh, generator=g); de = torch.randn(B, N, N, De, generator=g)
    if bf16:
        e, de = e.bfloat16(), de.bfloat16()
    mask = torch.ones(B, N, dtype=torch.bool); mask[B - 1, max(1, N - 3):] = False
    hg = h.to(gpu).requires_grad_(); eg = e.to(gpu).requires_grad_()
    h2, e2 = st(hg, eg, mask.to(gpu))
    assert st.last_path == "fused-stack"
    torch.autograd.backward([h2, e2], [dh.to(gpu), de.to(gpu)])
    names = dict(PMAP)
    if variant == "bias":
        names = {k: v for k, v in PMAP.items() if not (k.startswith("norm_edge") or k.startswith("dense_edge_r"))}
    if variant == "ungated":
        names = {k: v for k, v in names.items() if not k.startswith("attention_gates")}
    layers = [{k: getattr(getattr(blk, m), a_).detach().double().cpu().requires_grad_() for k, (m, a_) in names.items()}
              for blk in st.blocks]
    rms = None
    if train:
        b0 = st.blocks[0].mha
        seed = (b0.seed * 0x9E3779B97F4A7C15 + b0._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        rms = [torch.from_numpy(rng_ref.random_mask(layer_seed(seed, l), B, N, 8, p)) for l in range(Ly)]
    h64 = h.double().requires_grad_(); e64 = e.double().requires_grad_()
    ho, eo = O.stack_forward(h64, e64, mask, layers, num_heads=8, rand_masks=rms, **okw)
    flat = [t for lp in layers for t in lp.values()]
    gr = torch.autograd.grad([ho, eo], [h64, e64] + flat, [dh.double(), de.double()], allow_unused=True)
    if bf16:
        tol, ptol = dict(rtol=2e-2, arel=1e-2, zero_atol=2e-4), dict(rtol=3e-2, arel=2e-2, l2=3e-2, zero_atol=2e-4)   # (an analytically zero gradient, e.g. the edge-bias bias of the 'bias' variant, gets an absolute bound)
    else:
        tol, ptol = dict(rtol=2e-4, arel=5e-5), BWD
    gtol = tol if bf16 else BWD
    assert_close(h2, ho, name="h_out", **tol)
    assert_close(e2.float(), eo, name="e_out", **tol)
    assert_close(hg.grad, gr[0], name="dh", **gtol)
    assert_close(eg.grad.float(), gr[1], name="de", **gtol)
    gi = iter(gr[2:])
    for li, blk in enumerate(st.blocks):
        for k, (m, a_) in names.items():
            ref = next(gi)
            if ref is None:
                continue
            assert_close(getattr(getattr(blk, m), a_).grad, ref, name=f"L{li}.{k}", **ptol)


@pytest.mark.parametrize("bf16", [False, True])
def test_narrow_kernels_fully_masked_rows(bf16, gpu, egt_lib):
    """Rows whose keys are ALL masked (an empty graph; a one-node graph under a 50 % random mask): the additive -1e9 / -2e9
    semantics of egt_layers.py:89-113 -- uniform attention over the least-masked keys, gates exactly 0 -- through the
    De = 8 kernels' online softmax (forward) and saved-statistics recompute (backward)."""
    from egt_amd import EGTStack
    from egt_amd.fused import layer_seed
    from oracle import egt_oracle as O, rng_ref
    B, N, Dh, De, Ly, p = 3, 21, 64, 8, 2, 0.5
    torch.manual_seed(77)
    st = EGTStack(model_height=Ly, model_width=Dh, edge_width=De, num_heads=8, random_mask_prob=p, seed=6, fused=True).to(gpu).train(True)
    g = torch.Generator().manual_seed(5)
    h = torch.randn(B, N, Dh, generator=g); e = torch.randn(B, N, N, De, generator=g)
    dh = torch.randn(B, N, Dh, generator=g); de = torch.randn(B, N, N, De, generator=g)
    if bf16:
        e, de = e.bfloat16(), de.bfloat16()
    mask = torch.zeros(B, N, dtype=torch.bool)
    mask[0, :1] = True          # one real node: every (row, head) whose random mask hits it is fully masked
    mask[2, :13] = True         # graph 1 stays empty: every row fully masked by the key mask alone
    hg = h.to(gpu).requires_grad_(); eg = e.to(gpu).requires_grad_()
    h2, e2 = st(hg, eg, mask.to(gpu))
    assert st.last_path == "fused-stack"
    torch.autograd.backward([h2, e2], [dh.to(gpu), de.to(gpu)])
    layers = [{k: getattr(getattr(blk, m), a_).detach().double().cpu().requires_grad_() for k, (m, a_) in PMAP.items()}
              for blk in st.blocks]
    b0 = st.blocks[0].mha
    seed = (b0.seed * 0x9E3779B97F4A7C15 + b0._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
    rms = [torch.from_numpy(rng_ref.random_mask(layer_seed(seed, l), B, N, 8, p)) for l in range(Ly)]
    h64 = h.double().requires_grad_(); e64 = e.double().requires_grad_()
    ho, eo = O.stack_forward(h64, e64, mask, layers, num_heads=8, rand_masks=rms)
    gr = torch.autograd.grad([ho, eo], [h64, e64], [dh.double(), de.double()])
    tol = dict(rtol=2e-2, arel=1e-2, zero_atol=2e-4) if bf16 else dict(rtol=2e-4, arel=5e-5)
    gtol = tol if bf16 else BWD
    assert torch.isfinite(h2).all() and torch.isfinite(e2.float()).all()
    assert_close(h2, ho, name="h_out", **tol)
    assert_close(e2.float(), eo, name="e_out", **tol)
    assert_close(hg.grad, gr[0], name="dh", **gtol)
    assert_close(eg.grad.float(), gr[1], name="de", **gtol)
