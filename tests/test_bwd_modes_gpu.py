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
