"""The opt-in arithmetic mode of the fused backward (EGT_BWD_MATMUL=bf16x3: the channel contractions of
k_block_bwd_v5 as 3-term bfloat16 split products, per-product error 2^-16) must hold the SAME parity
tolerances as the exact fp32 kernels.  The switch is read once per process, so the block suite is re-run
in a subprocess with the variable set; only the absolute bound on analytically-zero gradient tensors is
relaxed (their value is pure rounding noise of the summands)."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_block_suite_with_bf16x3_backward(gpu, egt_lib):
    env = dict(os.environ, EGT_BWD_MATMUL="bf16x3", EGT_TEST_ZERO_ATOL="2e-4")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_block_gpu.py"),
                        os.path.join(REPO, "tests", "test_fullsize_gpu.py"), "-m", "gpu", "-x", "-q",
                        "-k", "stack or fused or fullsize or zinc500k"], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "passed" in r.stdout


def test_block_suite_with_tile_pair_backward(gpu, egt_lib):
    """EGT_BWD_V6=1: k_block_bwd_v6 (every 16-pair tile worked on by a pair of waves; csrc/egt_block_bwd6.hip) instead of
    k_block_bwd_v5 for the fp32 De >= 32 geometries.  Same suites, same tolerances -- and bit-identical gradients to v5 at
    the headline geometry (same operands, same association order)."""
    env = dict(os.environ, EGT_BWD_V6="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(REPO, "tests", "test_block_gpu.py"),
                        os.path.join(REPO, "tests", "test_fullsize_gpu.py"), "-m", "gpu", "-x", "-q",
                        "-k", "stack or fused or fullsize or zinc500k"], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "passed" in r.stdout
    code = ("import sys, torch; sys.path.insert(0, 'tests'); from test_block_gpu import run_block; "
            "o, dp, _ = run_block('residual_n64', torch.device('cuda:0'), fused=True); "
            "torch.save({**{k: v.cpu() for k, v in o.items()}, **{'p/' + k: v.cpu() for k, v in dp.items() if v is not None}}, sys.argv[1])")
    import tempfile
    import torch
    with tempfile.TemporaryDirectory() as td:
        outs = []
        for v in ("0", "1"):
            f = os.path.join(td, f"o{v}.pt")
            rr = subprocess.run([sys.executable, "-c", code, f], cwd=REPO, env=dict(os.environ, EGT_BWD_V6=v), capture_output=True, text=True, timeout=300)
            assert rr.returncode == 0, rr.stderr[-2000:]
            outs.append(torch.load(f))
        for k in outs[0]:
            assert torch.equal(outs[0][k], outs[1][k]), k
