"""De = 8 pair kernels (egt_narrow.hip): the driver's GPU suite runs the DEFAULT selection (VALU forward for fp32 and
bf16, VALU backward for bf16, v4r backward for fp32).  The kernel choice is read from the environment once per process,
so the other selections are exercised in child processes: the VALU backward forced for fp32, and the MFMA-tile kernels
(r4 / v4r) with the VALU kernels switched off -- all against the same oracle tests."""
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


def test_valu_backward_forced_for_fp32():
    _run({"EGT_NARROW_BWD": "1"})


def test_mfma_tile_kernels_with_the_valu_kernels_off():
    _run({"EGT_NO_NARROW": "1"})
