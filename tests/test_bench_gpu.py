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


def test_bench_default_calibrates_the_step_mode():
    """--graph auto (the default) at stack scope: a few steps of the eager stream and of the hipGraph replay are timed after the
    warm-up, the faster mode runs the timed region, and the line says which one and by how much (config.step_mode); the other
    mode's figure travels as the secondary leg."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--layers", "2"]
    r = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    sm = line["config"]["step_mode"]
    assert sm["chosen"] in ("eager", "graph") and sm["eager_ms_per_step"] > 0 and sm["graph_ms_per_step"] > 0
    assert (sm["chosen"] == "graph") == (sm["graph_ms_per_step"] < 0.99 * sm["eager_ms_per_step"])
    assert line["roofline"]["timed_in_region"] is True      # eager: per-launch hipEvents; replay: event-record nodes inside the captured graph
    assert line["roofline"]["launches"] >= 1 and line["median_ms_per_step"] > 0 and line["value_at_median"] > 0
    assert line["roofline"]["kernels_sum_ms_per_step"] > 0
    assert (line["config"]["hipgraph"] is not None) == (sm["chosen"] == "graph")
    other = "eager_step" if sm["chosen"] == "graph" else "hipgraph_replay"
    assert other in line and line[other] and line[other].get("value", 0) > 0, line.get(other)
    assert line["roofline"]["kernel"] == "k_block_bwd" and line["roofline"]["achieved"] > 0


def test_bench_graph_replay_under_launcher_with_the_eager_collective():
    """--graph on: forward + backward replayed from one hipGraph, the RCCL collective issued eagerly after each replay"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29634", os.path.join(REPO, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--layers", "2", "--graph", "on"]
    r = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["backend"] == "rccl"
    assert "hipGraph (6 replays)" in line["config"]["hipgraph"]            # 3 settle + 3 timed
    assert line["config"]["grad_allreduce_us"] is not None and line["config"]["grad_allreduce_us"] > 0
    assert line["roofline"]["timed_in_region"] is True and line["roofline"]["avg_launch_us"] > 0   # event-record nodes inside the graph
    assert "captured graph" in line["roofline"]["launches_sampled"] and line["config"]["rccl_nranks"] == 1
    assert line["value"] > 0


def test_bench_graph_replay_with_the_collective_captured():
    """--graph on --graph-collective: the RCCL all-reduce of the flat gradient buffer is a node of the step's hipGraph (a DP
    step is ONE host call); same line otherwise"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29635", os.path.join(REPO, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "2", "--no-cpu-baseline", "--layers", "2", "--graph", "on", "--graph-collective"]
    r = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["backend"] == "rccl"
    assert line["config"]["collective_in_graph"] is True and "gradient all-reduce" in line["config"]["hipgraph"]
    assert line["value"] > 0


def test_bench_one_rank_through_the_capi_communicator():
    """--dp-backend capi: the step's collective is egt_dp_allreduce (ncclAllReduce on the compute stream)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1",
           "--master-addr", "127.0.0.1", "--master-port", "29633", os.path.join(REPO, "bench.py"),
           "--gpus", "1", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--layers", "2", "--dp-backend", "capi"]
    r = subprocess.run(cmd, cwd=REPO, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["config"]["backend"] == "rccl (egt_dp_* C-ABI)"
    assert line["config"]["grad_allreduce_us"] is not None and line["config"]["grad_allreduce_us"] > 0
    assert line["value"] > 0


def test_capi_communicator_one_rank_allreduce():
    """egt_dp_unique_id / init / allreduce / finalize on a 1-rank RCCL communicator: SUM and AVG leave the buffer
    unchanged, the weighted form scales by local/global, a second init is refused, finalize allows a new one."""
    code = r"""
import torch, ctypes as C
from egt_amd import _lib as L
from egt_amd.dp import CapiComm, FlatGradAllReduce
torch.cuda.set_device(0)
c = CapiComm(rank=0, world=1)
lib = L.load()
assert lib.egt_dp_world() == 1 and lib.egt_dp_rank() == 0
x = torch.randn(100003, device="cuda"); ref = x.clone()
c.all_reduce_flat(x, average=False); c.all_reduce_flat(x, average=True)
torch.cuda.synchronize(); assert torch.equal(x, ref)
c.all_reduce_flat(x, True, local_count=3, global_count=4)
torch.cuda.synchronize(); assert torch.allclose(x, ref * 0.75)
ident = C.create_string_buffer(128)
assert lib.egt_dp_init(ident, 1, 0) == L.EGT_E_FLAGS
try:
    c.all_reduce_flat(x.double())
    raise SystemExit("dtype check missing")
except TypeError:
    pass
p = torch.nn.Parameter(torch.ones(5, device="cuda"))
fa = FlatGradAllReduce([p], comm=c); fa.flat.fill_(2.0); fa.all_reduce()
torch.cuda.synchronize(); assert torch.equal(p.grad, torch.full((5,), 2.0, device="cuda"))
c.close(); assert lib.egt_dp_world() == 0
c2 = CapiComm(rank=0, world=1); c2.close()
print("CAPI_DP_OK")
"""
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CAPI_DP_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]


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


def test_capi_allreduce_is_a_node_of_a_captured_graph():
    """egt_dp_allreduce under stream capture (1-rank RCCL communicator): the collective becomes part of the graph -- nothing runs at
    capture time, and every REPLAY applies it (the count-weighted form scales the buffer by local / global each time)."""
    code = r"""
import torch
from egt_amd.dp import CapiComm
torch.cuda.set_device(0)
c = CapiComm(rank=0, world=1)
x = torch.randn(100003, device="cuda"); ref = x.clone()
c.all_reduce_flat(x, True, local_count=1, global_count=1)       # un-captured first call: RCCL's lazy set-up
torch.cuda.synchronize(); assert torch.equal(x, ref)
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    c.all_reduce_flat(x, True, local_count=3, global_count=4)
torch.cuda.synchronize()
assert torch.equal(x, ref), "a captured collective must not execute at capture time"
g.replay(); torch.cuda.synchronize()
assert torch.allclose(x, ref * 0.75)
g.replay(); g.replay(); torch.cuda.synchronize()
assert torch.allclose(x, ref * 0.75 ** 3)
c.close()
print("CAPTURED_COLLECTIVE_OK")
"""
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=_env(), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "CAPTURED_COLLECTIVE_OK" in r.stdout, (r.stdout + r.stderr)[-3000:]
