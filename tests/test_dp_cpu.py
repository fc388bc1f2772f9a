"""world_size-2 gloo tests of the data-parallel plumbing on CPU: batch sharding
with no data-path collective + one flat gradient all-reduce reproduces the
full-batch gradient.  Per-rank gradients come from the CPU oracle (the HIP
kernels need a GPU); what is under test is egt_amd.dp."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from egt_amd.dp import FlatGradAllReduce, shard_batch
        from oracle import egt_oracle as O
        torch.manual_seed(0)
        Bg, N, Dh, De, H = 6, 5, 16, 8, 8
        g = torch.Generator().manual_seed(5)
        h = torch.randn(Bg, N, Dh, generator=g, dtype=torch.float64)
        e = torch.randn(Bg, N, N, De, generator=g, dtype=torch.float64)
        mask = torch.ones(Bg, N, dtype=torch.bool)
        mask[:, 4:] = False
        params = O.init_block_params(Dh, De, H, dtype=torch.float64, generator=g, randomize_norm=True)

        def grads(lo, hi, denom):
            ps = {k: torch.nn.Parameter(v.clone()) for k, v in params.items()}
            h2, e2 = O.block_forward(h[lo:hi], e[lo:hi], mask[lo:hi], ps, num_heads=H)
            loss = (h2.pow(2).sum() + e2.pow(2).sum()) / denom   # mean over the GLOBAL batch
            return ps, loss

        lo, hi = shard_batch(Bg, world, rank)
        ps, loss = grads(lo, hi, Bg)
        fa = FlatGradAllReduce(ps.values())
        loss.backward()
        # per-rank loss is scaled by 1/Bg, so SUM over ranks is the global gradient
        fa.all_reduce(average=False)
        ps_full, loss_full = grads(0, Bg, Bg)
        loss_full.backward()
        err = max(float((a.grad - b.grad).abs().max()) for a, b in zip(ps.values(), ps_full.values()))
        # average=True semantics: mean of per-rank grads
        fa2 = FlatGradAllReduce([torch.nn.Parameter(torch.full((3,), float(rank + 1)))])
        fa2.flat.copy_(torch.full((3,), float(rank + 1)))
        fa2.all_reduce(average=True)
        q.put((rank, lo, hi, err, fa2.flat.tolist(), fa.nbytes))
    finally:
        dist.destroy_process_group()


def test_flat_grad_allreduce_gloo_ws2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [(r[1], r[2]) for r in res] == [(0, 3), (3, 6)]
    for r in res:
        assert r[3] < 1e-12, f"rank {r[0]} grad mismatch {r[3]}"
        assert r[4] == [1.5, 1.5, 1.5]


def test_shard_batch_covers_everything():
    from egt_amd.dp import shard_batch
    for n in (1, 7, 128, 130):
        for w in (1, 2, 3, 8):
            spans = [shard_batch(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_flat_buffer_aliases_grads():
    from egt_amd.dp import FlatGradAllReduce
    ps = [torch.nn.Parameter(torch.randn(3, 4)), torch.nn.Parameter(torch.randn(5))]
    fa = FlatGradAllReduce(ps)
    (ps[0].sum() * 2 + ps[1].sum() * 3).backward()
    assert torch.equal(fa.flat, torch.cat([torch.full((12,), 2.0), torch.full((5,), 3.0)]))
    assert fa.nbytes == 17 * 4
    fa.zero()
    assert float(ps[0].grad.abs().sum()) == 0.0


def _worker_flat(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from egt_amd.dp import all_reduce_flat, flat_grad_view
        ps = [torch.nn.Parameter(torch.zeros(2, 3)), torch.nn.Parameter(torch.zeros(4))]
        flat = torch.arange(10, dtype=torch.float32) * (rank + 1)      # what the fused stack backward hands over
        ps[0].grad = flat[:6].view(2, 3); ps[1].grad = flat[6:]
        ok = flat_grad_view(ps, flat)
        other = torch.zeros(10)
        not_ok = flat_grad_view(ps, other)
        all_reduce_flat(flat, average=True)
        q.put((rank, ok, not_ok, flat.tolist(), ps[1].grad.tolist()))
    finally:
        dist.destroy_process_group()


def test_all_reduce_flat_adopted_views_gloo_ws2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_flat, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    want = [i * 1.5 for i in range(10)]            # mean of 1x and 2x
    for r in res:
        assert r[1] is True and r[2] is False
        assert r[3] == want and r[4] == want[6:]   # the .grad views see the reduced values


def _worker_uneven(rank, world, port, q):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from egt_amd.dp import FlatGradAllReduce, shard_batch
        Bg = 7                                           # world does not divide the global batch
        g = torch.Generator().manual_seed(11)
        x = torch.randn(Bg, 4, generator=g, dtype=torch.float64)
        w0 = torch.randn(4, 3, generator=g, dtype=torch.float64)
        lo, hi = shard_batch(Bg, world, rank)
        w = torch.nn.Parameter(w0.clone())
        fa = FlatGradAllReduce([w])
        (x[lo:hi] @ w).pow(2).sum(1).mean().backward()   # LOCAL-mean loss, as a replica computes it
        naive = fa.flat.clone()
        fa.all_reduce(average=True, local_count=hi - lo, global_count=Bg)
        wf = torch.nn.Parameter(w0.clone())
        (x @ wf).pow(2).sum(1).mean().backward()         # the global-batch mean
        err = float((fa.flat.view_as(wf) - wf.grad).abs().max())
        # the unweighted SUM / world differs whenever the shards differ in size
        t = naive.clone(); dist.all_reduce(t); t /= world
        err_naive = float((t.view_as(wf) - wf.grad).abs().max())
        q.put((rank, hi - lo, err, err_naive))
    finally:
        dist.destroy_process_group()


def test_uneven_shards_weighted_allreduce_gloo_ws2():
    """shards that differ by one graph must not over-weight the smaller shard."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_uneven, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [4, 3]
    for r in res:
        assert r[2] < 1e-12, f"weighted all-reduce is not the global mean: {r[2]}"
        assert r[3] > 1e-6, "control: the unweighted mean should differ for uneven shards"


# ---------------------------------------------------------------- the scheme driver under DP ---
class _StubModel(torch.nn.Module):
    """CPU stand-in with ZincDCTransformer's call convention"""
    def __init__(self, mc):
        super().__init__()
        self.emb = torch.nn.Parameter(torch.zeros(29))
        self.b = torch.nn.Parameter(torch.zeros(1))

    def forward(self, nf, fm, adj):
        return (self.emb[(nf + 1).long()] * (nf >= 0)).sum(1, keepdim=True) + self.b + adj.sum((1, 2))[:, None] / 40


def _worker_scheme(rank, world, port, q, store, save_path):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from egt_amd import training as T
        # 50 training graphs, global batch 16 -> 3 full batches + a remainder of 2 graphs (1 per rank);
        # 11 validation graphs, batch 16 -> one short batch split 6 / 5
        cfg = dict(scheme="zinc.svd", model_name="dp", num_epochs=3, initial_lr=0.05, batch_size=16, use_svd=False,
                   distributed=True, dataset_path=store, save_path=save_path, rlr_patience=1)
        logs = []
        s = T.ZincSVDScheme(cfg, model_factory=_StubModel, print_fn=logs.append)
        s.execute_training()
        steps = s.state.global_step
        q.put((rank, steps, [round(h["val_mae"], 10) for h in s.history], [h["lr"] for h in s.history],
               s.model.emb.detach().tolist(), s.state.save_best_epoch))
    finally:
        dist.destroy_process_group()


def test_scheme_driver_two_ranks_stay_in_lockstep(tmp_path):
    """execute_training under a 2-rank gloo group: the ranks shuffle alike (rank 0's seed), shard every global batch, take the
    SAME number of steps, see the same validation logs (all-reduced sums) and therefore the same lr / best-epoch decisions,
    end with identical weights, and only rank 0 writes the checkpoint / weight files."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    from test_data import _store
    store = _store(tmp_path, n_train=50, n_val=11)
    save = str(tmp_path / "run")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_scheme, args=(r, 2, port, q, store, save)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    r0, r1 = res
    assert r0[1] == r1[1] == 3 * 4                    # 4 steps per epoch on both ranks (the 2-graph remainder is kept: 1 + 1)
    assert r0[2] == r1[2] and r0[3] == r1[3] and r0[5] == r1[5]
    assert r0[4] == r1[4] and any(abs(x) > 0 for x in r0[4])
    assert os.path.exists(os.path.join(save, "checkpoint", "ckpt.pt")) and not os.path.exists(os.path.join(save, "checkpoint", "ckpt.pt.tmp"))
    assert os.path.exists(os.path.join(save, "saved", "dp.npz"))
