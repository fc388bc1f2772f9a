"""bench.py's launch contract, end to end on the GPU box: a launched 1-rank job goes through RCCL
(init, the flat-buffer collective, destroy) and reports what really ran; asking for more ranks than
there are GPUs is an error, never a silent 1-rank run (VERDICT r1 "what's missing" #2)."""
import json
import os
import subprocess
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _env():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def _json_line(out: str):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_bench_one_rank_under_launcher_uses_rccl():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29631", os.path.join(REPO, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--layers", "2"]
    r = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["parallelism"] == "dp1"
    assert line["config"]["backend"] == "rccl"
    assert line["config"]["grad_allreduce_us"] is not None and line["config"]["grad_allreduce_us"] > 0
    assert line["config"]["flat_grad_adopted"] is True
    assert line["value"] > 0 and line["scaling"] == "weak"


def test_bench_refuses_more_ranks_than_gpus():
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n), "--steps", "1",
                        "--warmup", "0", "--no-cpu-baseline"], cwd=REPO, env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0
    assert "refusing" in (r.stderr + r.stdout)
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]   # no bench line for a run that did not happen


def test_bench_world_size_mismatch_is_an_error():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29632", os.path.join(REPO, "bench.py"),
           "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "WORLD_SIZE=1" in (r.stderr + r.stdout)
