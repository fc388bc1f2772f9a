#!/usr/bin/env python
"""graphs/sec of the EGT attention-block stack, fwd+bwd, on MI355X.

A "step" = one pass of the hot path over one batch of synthetic padded graphs:
forward through Ly stacked (h,e,mask)->(h',e') blocks, backward from N(0,1)
upstream gradients down to (dh, de) and every parameter gradient, and — for
N>1 ranks — the single flat RCCL gradient all-reduce.  Inputs are resident in
HBM before the timed region.  Workload = BASELINE.json configs[1]:
ZINC-500K shapes (Dh=64, De=64, H=8, d=8, Ly=10), padded N=64, fp32, B=128
graphs per GPU (weak scaling), node counts ~U[9,37], random_mask_prob=0.1.

Usage: python bench.py --gpus N --steps K --warmup W
N>1 runs one rank per GPU over RCCL.  Either launch it under torch.distributed.run
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
--master-port P bench.py --gpus N ...`) or just run `python bench.py --gpus N`: without a
launcher in the environment the script re-executes itself under torch.distributed.run.  A
world size that differs from --gpus, or fewer visible GPUs than ranks, is an ERROR (never a
silent 1-rank run).  --scaling weak (default): 128 graphs per GPU; strong: 128 graphs split
over the ranks (lib/training/training_base.py:230-247: Keras splits the global batch).
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0   # same guide: what a float4 copy sustains (79 % of the spec)

WORKLOADS = {
    # BASELINE.json configs[1]
    "zinc500k_n64": dict(B=128, N=64, Dh=64, De=64, H=8, Ly=10, nodes=(9, 37), rand_p=0.1),
    "zinc500k_n64_b1024": dict(B=1024, N=64, Dh=64, De=64, H=8, Ly=10, nodes=(9, 37), rand_p=0.1),   # large-batch sanity
    # the full-occupancy variant of config 2 (every padded slot a real node: n_b = 64)
    "zinc500k_n64_nomask": dict(B=128, N=64, Dh=64, De=64, H=8, Ly=10, nodes=(9, 37), rand_p=0.0),   # diagnosis: what the in-kernel mask RNG costs
    "zinc500k_n64_full": dict(B=128, N=64, Dh=64, De=64, H=8, Ly=10, nodes=(64, 64), rand_p=0.1),
    # the other BASELINE.json configs' shapes, for reference runs -- not bench lines
    "zinc100k_n37": dict(B=128, N=37, Dh=48, De=48, H=8, Ly=4, nodes=(9, 37), rand_p=0.1),
    "cifar10_n150_fp32": dict(B=128, N=150, Dh=64, De=8, H=8, Ly=4, nodes=(85, 150), rand_p=0.1),
    "cifar10_n150": dict(B=128, N=150, Dh=64, De=8, H=8, Ly=4, nodes=(85, 150), rand_p=0.1, edge_dtype="bf16"),
    "pattern500k_n120": dict(B=16, N=120, Dh=64, De=8, H=8, Ly=16, nodes=(44, 120), rand_p=0.1),
    "pattern500k_n120_b128": dict(B=128, N=120, Dh=64, De=8, H=8, Ly=16, nodes=(44, 120), rand_p=0.1),
    # SURVEY 8(d) for config 4: n_b ~ U[44,188] padded to the PER-BATCH maximum (what the reference's padded_batch does,
    # scheme_base.py:66 + dataset_base.py:106-111: N = None -> the batch's largest graph), and the fixed N = 188 row
    "pattern500k_bmax": dict(B=16, N=None, Dh=64, De=8, H=8, Ly=16, nodes=(44, 188), rand_p=0.1),
    "pattern500k_bmax_b128": dict(B=128, N=None, Dh=64, De=8, H=8, Ly=16, nodes=(44, 188), rand_p=0.1),
    "pattern500k_n188": dict(B=16, N=188, Dh=64, De=8, H=8, Ly=16, nodes=(44, 188), rand_p=0.1),
    "pattern500k_n188_b128": dict(B=128, N=188, Dh=64, De=8, H=8, Ly=16, nodes=(44, 188), rand_p=0.1),
    # the same graphs padded to the next multiple of 16 (what a caller's padded_batch can do for free):
    # unlocks the 16-row-exact backward (prologue inside the pair kernel) at the price of 14 % more pairs
    "cifar10_n150_pad160": dict(B=128, N=160, Dh=64, De=8, H=8, Ly=4, nodes=(85, 150), rand_p=0.1),
    "pattern500k_n120_pad128_b128": dict(B=128, N=128, Dh=64, De=8, H=8, Ly=16, nodes=(44, 120), rand_p=0.1),
    # BASELINE.json configs[4]: synthetic dense graphs, the MFMA / HBM roofline stress.  CORE-OP scope of SURVEY 8(d)(i):
    # ([QKV,E,G],mask) -> (V_att,H_hat) forward + backward on the MFMA inner-op kernels (egt_attn_mfma.hip), every key real
    "synthetic_n512": dict(B=8, N=512, Dh=512, De=32, H=8, Ly=1, nodes=(512, 512), rand_p=0.1, scope="core"),
    "synthetic_n512_b32": dict(B=32, N=512, Dh=512, De=32, H=8, Ly=1, nodes=(512, 512), rand_p=0.1, scope="core"),
    # the same config at the BLOCK scope of SURVEY 8(d)(ii): (h, e, mask) -> (h', e') of ONE attention block, forward + backward with
    # all parameter gradients.  d = 64 is outside the fused pair kernels (d <= 8): the composed path -- HIP edge projections / edge
    # update (egt_edge.hip) + the MFMA inner op + rocBLAS node-side Dense
    "synthetic_n512_block": dict(B=8, N=512, Dh=512, De=32, H=8, Ly=1, nodes=(512, 512), rand_p=0.1, scope="block"),
    "synthetic_n512_block_b32": dict(B=32, N=512, Dh=512, De=32, H=8, Ly=1, nodes=(512, 512), rand_p=0.1, scope="block"),   # four rounds of workgroups, node-side GEMMs with 16 k rows
    "synthetic_n512_block_nomask": dict(B=8, N=512, Dh=512, De=32, H=8, Ly=1, nodes=(512, 512), rand_p=0.0, scope="block"),   # diagnosis: what the in-kernel mask RNG costs
}
# what the metric string says after "graphs/sec EGT fwd+bwd, " (BASELINE.json's metric is quoted on the first)
METRIC_OF = {"zinc500k_n64": "ZINC-500K padded N=64", "zinc500k_n64_nomask": "ZINC-500K padded N=64 (random_mask_prob = 0)", "zinc500k_n64_b1024": "ZINC-500K padded N=64 (B=1024)",
             "zinc500k_n64_full": "ZINC-500K shapes N=64, every node real (full occupancy)",
             "pattern500k_bmax": "PATTERN-500K shapes padded to the per-batch max (B=16)", "pattern500k_bmax_b128": "PATTERN-500K shapes padded to the per-batch max (B=128)",
             "pattern500k_n188": "PATTERN-500K shapes padded N=188 (B=16)", "pattern500k_n188_b128": "PATTERN-500K shapes padded N=188 (B=128)",
             "zinc100k_n37": "ZINC-100K shapes padded N=37", "cifar10_n150_fp32": "CIFAR10-500K shapes padded N=150 (fp32 edge tensors)",
             "cifar10_n150": "CIFAR10-500K shapes padded N=150 (bf16 edge tensors)", "pattern500k_n120": "PATTERN-500K shapes padded N=120 (B=16)",
             "pattern500k_n120_b128": "PATTERN-500K shapes padded N=120 (B=128)", "cifar10_n150_pad160": "CIFAR10-500K shapes padded N=160",
             "pattern500k_n120_pad128_b128": "PATTERN-500K shapes padded N=128 (B=128)",
             "synthetic_n512": "synthetic dense N=512 heads=8 d=64 (core op)", "synthetic_n512_b32": "synthetic dense N=512 heads=8 d=64 (core op, B=32)",
             "synthetic_n512_block": "synthetic dense N=512 heads=8 d=64 (block op)",
             "synthetic_n512_block_b32": "synthetic dense N=512 heads=8 d=64 (block op, B=32)",
             "synthetic_n512_block_nomask": "synthetic dense N=512 heads=8 d=64 (block op, random_mask_prob = 0)"}
FP32_MFMA_PEAK_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD x 1024 SIMDs x 2.4 GHz


def algorithmic_bytes(kernel: str, w: dict) -> float:
    """ALGORITHMIC HBM bytes one launch of `kernel` must move (fp32, s=4): the
    per-graph-per-layer figures of SURVEY §8(d) split per launch."""
    B, N, Dh, De, H = w["B"], w["N"], w["Dh"], w["De"], w["H"]
    pairs = B * N * N
    s = 4
    se = 2 if w.get("edge_dtype", "f32") == "bf16" else 4   # edge tensors in HBM
    node = B * N * Dh * s
    table = {
        # fused block: fwd reads e, writes e' (+ node rows); bwd reads e, de', writes de
        "k_block_fwd": pairs * De * se * 2 + 3 * node,
        "k_block_bwd": pairs * De * se * 3 + 3 * node,
        # composed path
        "k_edge_proj_fwd": pairs * (De + 2 * H) * s,
        "k_edge_proj_bwd": pairs * (2 * De + 2 * H) * s,
        "k_edge_update_fwd": pairs * (2 * De + H) * s,
        "k_edge_update_bwd": pairs * (De + 2 * H) * s,
        "k_attn_fwd": pairs * 3 * H * s + 4 * node,
        "k_attn_bwd_row": pairs * 7 * H * s + 4 * node,
        "k_attn_bwd_dq": pairs * H * s + 2 * node,
        "k_attn_bwd_dkv": pairs * 2 * H * s + 3 * node,
    }
    return float(table.get(kernel, 0.0))


def make_inputs(w, dev, seed=1234):
    g = torch.Generator().manual_seed(seed)
    B, N, Dh, De = w["B"], w["N"], w["Dh"], w["De"]
    lo, hi = w["nodes"]
    n = torch.randint(lo, hi + 1, (B,), generator=g)
    mask = torch.arange(N)[None, :] < n[:, None]
    h = torch.randn(B, N, Dh, generator=g)
    e = torch.randn(B, N, N, De, generator=g)
    dh = torch.randn(B, N, Dh, generator=g)
    de = torch.randn(B, N, N, De, generator=g)
    return [t.to(dev) for t in (h, e, mask, dh, de)]


def make_zinc_inputs(w, dev, seed=1234):
    """synthetic molecules in the reference's input format (lib/data/datasets/zinc.py): node_features [B,N] int
    (28 atom types, padding -1), feature_matrix [B,N,N] int (bond type on edges, -1 elsewhere), graph_matrix
    [B,N,N] 0/1 symmetric adjacency (random sparse, average degree ~2.2 like ZINC), target [B,1]."""
    g = torch.Generator().manual_seed(seed)
    B, N = w["B"], w["N"]
    lo, hi = w["nodes"]
    n = torch.randint(lo, hi + 1, (B,), generator=g)
    real = torch.arange(N)[None, :] < n[:, None]
    nf = torch.randint(0, 28, (B, N), generator=g)
    nf[~real] = -1
    pr = (1.1 / n.float().clamp(min=2))[:, None, None]
    adj = (torch.rand(B, N, N, generator=g) < pr).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float() * (real[:, :, None] & real[:, None, :]).float()
    adj = adj * (1 - torch.eye(N))[None]
    bond = torch.randint(0, 4, (B, N, N), generator=g)
    bond = torch.triu(bond, 1); bond = bond + bond.transpose(1, 2)
    fm = torch.where(adj > 0, bond, torch.tensor(-1))
    tgt = torch.randn(B, 1, generator=g)
    return [t.to(dev) for t in (nf.int(), fm.int(), adj, tgt)]


def algorithmic_flops(kernel: str, w: dict, scope: str) -> float:
    """ALGORITHMIC flops of ONE STEP's launches of an MFMA-bound kernel (the channel FFN costs
    24 W^2 flop per row fwd+bwd = 8 W^2 forward + 16 W^2 backward)."""
    B, N, Dh, De, Ly = w["B"], w["N"], w["Dh"], w["De"], w["Ly"]
    n_edge = Ly - 1 if scope == "model" else Ly        # the model's last edge FFN is not on a path to the output
    per_row = {"k_ffn_fwd": 8.0, "k_ffn_bwd": 16.0}.get(kernel)
    if per_row is None:
        return 0.0
    return per_row * (n_edge * B * N * N * De * De + Ly * B * N * Dh * Dh)


def event_pair_overhead_us(n=64):
    """what an EMPTY hipEvent pair on the launch stream measures (two records back to back): the share of a per-launch figure that
    is the event machinery, not the kernel and its launch gap.  Median of n pairs, after a synchronisation (untimed phase only)."""
    st = torch.cuda.current_stream()
    pairs = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(st); b.record(st)
        pairs.append((a, b))
    st.synchronize()
    ts = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
    return ts[len(ts) // 2]


def prof_read_all(lib):
    buf = C.create_string_buffer(4096)
    lib.egt_prof_names(buf, 4096)
    out = {}
    for name in buf.value.decode().split():
        cnt, ms = C.c_int64(0), C.c_double(0.0)
        lib.egt_prof_read(name.encode(), C.byref(cnt), C.byref(ms))
        if cnt.value:
            out[name] = (cnt.value, ms.value)
    return out


def cpu_baseline_model(w, seconds=12.0, pattern=False):
    """whole-model scope: the torch-CPU restatement of the ZINC / PATTERN model (oracle/egt_model_oracle.py), fwd + loss + bwd."""
    from oracle import egt_model_oracle as MO
    Bs = 8
    cfg = dict(model_width=w["Dh"], edge_width=w["De"], num_heads=w["H"], model_height=w["Ly"], upto_hop=16)
    if pattern:
        cfg.update(num_node_features=3, num_edge_features=0, num_targets=2)
    nf, fm, adj, tgt = make_zinc_inputs(dict(w, B=Bs), "cpu", seed=77)
    g = torch.Generator().manual_seed(3)
    p = {k: v.requires_grad_() for k, v in MO.init_zinc_params(cfg, generator=g).items()}
    rms = [torch.rand(Bs, w["N"], w["N"], w["H"], generator=g) < w["rand_p"] for _ in range(w["Ly"])]
    ycls = torch.randint(0, 2, (Bs, w["N"]), generator=g)
    cw = MO.class_weights_from_sizes([979220, 209900]).float()
    nf3 = torch.where(nf >= 0, nf % 3, nf)

    def step():
        if pattern:
            lo, m = MO.pattern_forward(nf3, adj, p, cfg, rand_masks=rms)
            loss = MO.weighted_sparse_xent_loss(lo, ycls, m, cw)
        else:
            loss = MO.mae_loss(MO.zinc_forward(nf, fm, adj, p, cfg, rand_masks=rms), tgt)
        torch.autograd.grad(loss, [v for k, v in p.items()], allow_unused=True)

    nthr = torch.get_num_threads()
    thr = min(nthr, 8)
    torch.set_num_threads(thr)
    try:
        step()
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < seconds:
            step(); reps += 1
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(nthr)
    return dict(value=Bs * reps / dt, unit="graphs/s", cores=thr, kind="port",
                sample=f"{reps} fwd+loss+bwd steps of the whole {'PATTERN' if pattern else 'ZINC'} model (Ly={w['Ly']}) on B={Bs} graphs "
                       f"(N={w['N']}, fp32, torch-CPU restatement, {dt:.1f}s, host has {os.cpu_count()} cpus)")


def cpu_baseline(w, seconds=12.0, Bs=8):
    """Reference-equivalent CPU path: the torch-CPU fp32 op-by-op restatement of
    the TF op sequence (oracle/egt_oracle.py), fwd+bwd by autograd, on the host
    cores.  Bounded sample of the same workload (fewer graphs per batch)."""
    from oracle import egt_oracle as O
    ws = dict(w, B=Bs)
    h, e, mask, dh, de = make_inputs(ws, "cpu", seed=77)
    g = torch.Generator().manual_seed(3)
    layers = [{k: v.requires_grad_() for k, v in
               O.init_block_params(w["Dh"], w["De"], w["H"], generator=g).items()}
              for _ in range(w["Ly"])]
    rms = [torch.rand(Bs, w["N"], w["N"], w["H"], generator=g) < w["rand_p"] for _ in range(w["Ly"])]
    h.requires_grad_(); e.requires_grad_()
    flat = [p for L_ in layers for p in L_.values()]

    def step():
        h2, e2 = O.stack_forward(h, e, mask, layers, num_heads=w["H"], rand_masks=rms)
        torch.autograd.grad([h2, e2], [h, e] + flat, [dh, de])

    def timed(sec):
        step()  # warm
        t0 = time.perf_counter()
        reps = 0
        while True:
            step()
            reps += 1
            if time.perf_counter() - t0 >= sec:
                break
        return reps, time.perf_counter() - t0

    nthr = torch.get_num_threads()
    ncpu = os.cpu_count() or nthr
    try:                                                 # cores this process may really use (affinity mask, cgroup quota)
        ncpu = min(ncpu, len(os.sched_getaffinity(0)))
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            ncpu = max(1, min(ncpu, int(float(q[0]) / float(q[1]) + 0.5)))
    except Exception:  # noqa: BLE001
        pass
    # the all-cores figure (every core this process may use) and the 8-core figure.  Each leg is timed TWICE; when the
    # two runs differ by more than 10 % a third decides (median) -- a baseline that wanders between runs of the same code is not one.
    # The op sequence works on small tensors: beyond a handful of threads torch's intra-op pool only adds hand-off cost, and on a
    # 256-CPU host ONE step at 256 threads took over a minute (round 5).  So the all-cores leg starts with a probe -- one block on one
    # graph, at all cores and at 8 threads; when all cores run the probe at less than half the 8-thread rate the full leg is not run
    # and the all-cores figure is the 8-thread figure scaled by the probe's ratio (marked `extrapolated`): the default bench.py run
    # has to finish within minutes.
    def probe(thr):
        wp = dict(w, Ly=1)
        hp, ep, mp, dhp, dep = make_inputs(dict(wp, B=1), "cpu", seed=78)
        hp.requires_grad_(); ep.requires_grad_()
        lp = [layers[0]]
        rp = [rms[0][:1]]
        def one():
            a2, b2 = O.stack_forward(hp, ep, mp, lp, num_heads=w["H"], rand_masks=rp)
            torch.autograd.grad([a2, b2], [hp, ep] + list(lp[0].values()), [dhp, dep])
        torch.set_num_threads(thr)
        try:
            one()
            t_ = time.perf_counter(); n_ = 0
            while n_ < 3 or (time.perf_counter() - t_ < 0.5 and n_ < 50):
                one(); n_ += 1
            return n_ / (time.perf_counter() - t_)
        finally:
            torch.set_num_threads(nthr)

    legs = [ncpu, 8] if ncpu > 8 else [ncpu]
    ratio = None
    if len(legs) > 1:
        ratio = probe(ncpu) / probe(8)
    runs = []                                            # (threads, graphs/s, reps, seconds, spread, extrapolated)
    per = max(2.0, seconds / (2.0 * len(legs)))
    for thr in reversed(legs):                           # the 8-thread leg first: the extrapolation needs it
        if thr == ncpu and ratio is not None and ratio < 0.5:
            eight = runs[0]
            runs.append((thr, eight[1] * ratio, 0, 0.0, 0.0, True))
            continue
        torch.set_num_threads(thr)
        try:
            rates, reps_t, secs_t = [], 0, 0.0
            for _ in range(2):
                reps, dt = timed(per)
                rates.append(Bs * reps / dt); reps_t += reps; secs_t += dt
            if abs(rates[0] - rates[1]) > 0.10 * max(rates):
                reps, dt = timed(per)
                rates.append(Bs * reps / dt); reps_t += reps; secs_t += dt
        finally:
            torch.set_num_threads(nthr)
        rs = sorted(rates)
        runs.append((thr, rs[len(rs) // 2], reps_t, secs_t, (rs[-1] - rs[0]) / rs[len(rs) // 2], False))
    runs = runs[::-1] if len(runs) > 1 else runs         # [all cores, 8 threads]
    best = max(runs, key=lambda r: r[1])                 # the baseline is the FASTER thread count
    out = dict(value=best[1], unit="graphs/s", cores=best[0], kind="port",
               all_cores=dict(threads=runs[0][0], value=runs[0][1], spread=runs[0][4], extrapolated=runs[0][5],
                              probe_ratio_vs_8_threads=ratio),
               eight_cores=(dict(threads=runs[1][0], value=runs[1][1], spread=runs[1][4]) if len(runs) > 1 else None),
               sample=f"{best[2]} fwd+bwd steps of the Ly={w['Ly']} block stack on B={Bs} graphs "
                      f"(N={w['N']}, fp32, torch-CPU restatement of the TF op sequence, "
                      f"{best[3]:.1f}s, host has {os.cpu_count()} cpus, {ncpu} usable; median of 2-3 timed runs per thread count: "
                      + ", ".join(f"{t} threads: {v:.1f} graphs/s ({'extrapolated from a one-block probe' if ex else f'spread {sp:.0%}'})"
                                  for t, v, _, _, sp, ex in runs) + ")")
    return out


def cpu_baseline_core(w, seconds=12.0):
    """core-op scope: the torch-CPU fp32 op-by-op restatement of egt_layers.py:57-143 (oracle/egt_oracle.egt_forward),
    fwd + bwd by autograd, on the host cores; bounded sample (one graph per step)."""
    from oracle import egt_oracle as O
    Bs, N, H, d = 1, w["N"], w["H"], w["Dh"] // w["H"]
    g = torch.Generator().manual_seed(77)
    QKV = torch.randn(Bs, N, 3 * d * H, generator=g).requires_grad_()
    E = torch.randn(Bs, N, N, H, generator=g).requires_grad_()
    G = torch.randn(Bs, N, N, H, generator=g).requires_grad_()
    mask = torch.ones(Bs, N, dtype=torch.bool)
    rm = torch.rand(Bs, N, N, H, generator=g) < w["rand_p"]
    dV = torch.randn(Bs, N, d * H, generator=g); dH = torch.randn(Bs, N, N, H, generator=g)

    def step():
        V, Hh, _ = O.egt_forward(QKV, E, G, None, mask, rand_mask=rm, drop_keep=None, num_heads=H, clip_logits_value=(-5.0, 5.0),
                                 scale_degree=False, scaler_type="log", num_virtual_nodes=0, attn_dropout=0.0)
        torch.autograd.grad([V, Hh], [QKV, E, G], [dV, dH])

    nthr = torch.get_num_threads()
    thr = min(nthr, 8)
    torch.set_num_threads(thr)
    try:
        step()
        t0 = time.perf_counter(); reps = 0
        while time.perf_counter() - t0 < seconds:
            step(); reps += 1
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(nthr)
    return dict(value=Bs * reps / dt, unit="graphs/s", cores=thr, kind="port",
                sample=f"{reps} fwd+bwd steps of the inner op on B={Bs} graph (N={N}, H={H}, d={d}, fp32, torch-CPU restatement of the TF op "
                       f"sequence egt_layers.py:57-143, {dt:.1f}s, host has {os.cpu_count()} cpus)")


def run_core(args, w, dev, lib, rank, world, use_dist):
    """BASELINE config 5 at the core-op scope of SURVEY 8(d)(i): a step = egt_attn_mfma_fwd + egt_attn_mfma_bwd through the
    C-ABI on buffers resident in HBM (one shared workspace: q/k/v are packed once), training mode with the in-kernel
    random mask (a fresh seed per step).  No parameters at this scope: N > 1 ranks are independent shards (no collective)."""
    from egt_amd import _lib as L
    B, N, H, Dh = w["B"], w["N"], w["H"], w["Dh"]
    d = Dh // H
    g = torch.Generator().manual_seed(1234 + rank)
    mk = lambda *s: torch.randn(*s, generator=g).to(dev)
    qkv, E, G = mk(B, N, 3 * Dh), mk(B, N, N, H), mk(B, N, N, H)
    dV, dHx = mk(B, N, Dh), mk(B, N, N, H)
    km = torch.ones(B, N, dtype=torch.uint8, device=dev)
    desc = L.AttnDesc(B=B, N=N, H=H, d=d, dtype=L.EGT_F32, flags=L.F_EDGE_INPUT | L.F_GATE_INPUT | L.F_CLIP | L.F_TRAINING,
                      clip_lo=-5.0, clip_hi=5.0, random_mask_prob=float(w["rand_p"]), attn_dropout=0.0, num_virtual_nodes=0,
                      reserved=L.ATTN_WS_SHARED, seed=0)
    assert lib.egt_attn_mfma_supported(C.byref(desc), 0) == 1
    v_att = torch.empty(B, N, Dh, device=dev); h_hat = torch.empty(B, N, N, H, device=dev)
    rowstats = torch.empty(B, N, H, 4, device=dev)
    ws = torch.empty(lib.egt_attn_mfma_workspace_bytes(C.byref(desc)), dtype=torch.uint8, device=dev)
    d_qkv = torch.empty_like(qkv); d_E = torch.empty_like(E); d_G = torch.empty_like(G)
    st = L.current_stream()
    state = {"i": 0}

    def step():
        state["i"] += 1
        desc.seed = (1 + rank) * 1000003 + state["i"]      # the replicas and the steps draw independent random masks
        L.check(lib.egt_attn_mfma_fwd(C.byref(desc), L.ptr(qkv), L.ptr(E), L.ptr(G), L.ptr(km), None, None, L.ptr(v_att), L.ptr(h_hat),
                                      L.ptr(rowstats), L.ptr(ws), st))
        L.check(lib.egt_attn_mfma_bwd(C.byref(desc), L.ptr(qkv), L.ptr(E), L.ptr(G), L.ptr(km), None, None, L.ptr(v_att), L.ptr(rowstats),
                                      L.ptr(dV), L.ptr(dHx), L.ptr(d_qkv), L.ptr(d_E), L.ptr(d_G), L.ptr(ws), st))

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    dominant = "k_attn_mfma_bwd_kv"
    for _ in range(args.warmup):
        step()
    fence()
    if not args.no_prof:
        lib.egt_prof_filter(dominant.encode()); lib.egt_prof_enable(2)
    sev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step events: the median beside the wall-clock mean
    cur = torch.cuda.current_stream()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sev[i].record(cur)
        step()
    sev[args.steps].record(cur)
    fence()
    elapsed = time.perf_counter() - t0
    lib.egt_prof_enable(0)
    step_ms = sorted(sev[i].elapsed_time(sev[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if step_ms else None
    graphs_step = B
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
        gs = torch.tensor([B], device=dev, dtype=torch.int64)
        dist.all_reduce(gs); graphs_step = int(gs.item())
    dom_prof = prof_read_all(lib) if not args.no_prof else {}
    prof = {}
    if not args.no_prof:
        lib.egt_prof_filter(b""); lib.egt_prof_enable(2)
        for _ in range(min(args.steps, 10)):
            step()
        fence()
        lib.egt_prof_enable(0)
        prof = prof_read_all(lib)
    if rank != 0:
        return
    pairs = B * N * N
    # ALGORITHMIC flops (core op 12 N^2 Dh per graph fwd+bwd): forward QK^T + A.V = 4, backward dP + dV + dK = 6
    # in k_attn_mfma_bwd_kv, dQ = 2 in k_attn_mfma_bwd_q; the backward's recompute of S (2 more) is executed, not counted
    fl = {"k_attn_mfma_fwd": 4.0 * pairs * Dh, "k_attn_mfma_bwd_kv": 6.0 * pairs * Dh, "k_attn_mfma_bwd_q": 2.0 * pairs * Dh}
    roof = None
    if prof:
        cnt, ms = dom_prof.get(dominant, prof[dominant])
        avg_s = ms / cnt / 1e3
        ach = fl[dominant] / avg_s / 1e12
        pmc = {}
        try:
            pmc = json.load(open(os.path.join(REPO, "profiles", "pmc_mfma.json"))).get(args.workload, {})
        except Exception:  # noqa: BLE001
            pmc = {}
        roof = dict(bound="mfma", kernel=dominant, timed_in_region=dominant in dom_prof, achieved=ach, peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s",
                    frac=ach / FP32_MFMA_PEAK_TF, executed_frac=ach * (8.0 / 6.0) / FP32_MFMA_PEAK_TF,
                    step_frac=12.0 * pairs * Dh / (elapsed / args.steps) / 1e12 / FP32_MFMA_PEAK_TF / max(world, 1) * (graphs_step / B),
                    traffic=(pmc.get(dominant) or {}).get("hbm_bytes_per_launch"),
                    mfma_busy=(pmc.get(dominant) or {}).get("mfma_busy"),
                    pmc_source=("static: profiles/pmc_mfma.json = committed rocprofv3 --pmc passes of this workload (SQ_VALU_MFMA_BUSY_CYCLES / "
                                "(4 x SQ_BUSY_CU_CYCLES); HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024); not re-measured in this run") if pmc else None,
                    avg_launch_us=avg_s * 1e6, launches=cnt, algorithmic_flops_per_launch=fl[dominant],
                    kernels={k: dict(launches=v[0], avg_us=v[1] / v[0] * 1e3, share=v[1] / sum(x[1] for x in prof.values()),
                                     tflops=(fl[k] / (v[1] / v[0] / 1e3) / 1e12) if k in fl else None)
                             for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])})
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline_core(w, args.cpu_seconds)
    line = {
        "metric": "graphs/sec EGT fwd+bwd, " + METRIC_OF[args.workload],
        "value": graphs_step * args.steps / elapsed, "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "median_ms_per_step": median_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: core op ([QKV,E,G],mask) -> (V_att,H_hat) fwd+bwd (egt_layers.py:57-143 under autodiff), "
                               "training mode, in-kernel random mask", "scope": "core", "graphs_per_gpu": B, "global_batch": graphs_step,
                   "N": N, "Dh": Dh, "H": H, "d": d, "De": w["De"], "random_mask_prob": w["rand_p"], "nodes": list(w["nodes"]),
                   "path": "mfma inner op (pack + k_attn_mfma_fwd; pack + k_attn_mfma_bwd_kv + k_attn_mfma_bwd_q), one shared workspace",
                   "parallelism": f"dp{world}", "backend": "independent shards (the core op has no parameters: no collective)",
                   "tflops_step": 12.0 * pairs * Dh * (graphs_step / B) / (elapsed / args.steps) / 1e12},
        "roofline": roof, "cpu_baseline": cpu,
    }
    print(json.dumps(line))


def run_block(args, w, dev, lib, rank, world, use_dist):
    """BASELINE config 5 at the block scope of SURVEY 8(d)(ii): a step = forward + backward of ONE EGTBlock (composed path) with
    every parameter gradient, training mode with the in-kernel random mask; N > 1: batch-DP, the gradients all-reduced through one
    flat buffer per step.  Roofline: the step's ALGORITHMIC flops (3 (6 N^2 De H + 8 N Dh^2 + 4 N^2 Dh) per graph)
    against the fp32 matrix peak -- the scope is MFMA-bound by arithmetic intensity (34.7 flop/B) -- and the dominant HIP kernel's own."""
    from egt_amd import EGTBlock
    from egt_amd.dp import FlatGradAllReduce
    B, N, H, Dh, De = w["B"], w["N"], w["H"], w["Dh"], w["De"]
    blas = os.environ.get("EGT_BENCH_BLAS", "hipblaslt")   # library of the node-side fp32 GEMMs ([B N, 512] x [512, 1536] ...): hipBLASLt measured ~1 % ahead of the default
    try:
        torch.backends.cuda.preferred_blas_library(blas)
    except Exception:  # noqa: BLE001
        blas = "default"
    g = torch.Generator().manual_seed(1234 + rank)
    torch.manual_seed(7)                                  # the same parameters on every rank
    blk = EGTBlock(model_width=Dh, edge_width=De, num_heads=H, random_mask_prob=w["rand_p"]).to(dev).train()
    h = torch.randn(B, N, Dh, generator=g).to(dev).requires_grad_()
    e = torch.randn(B, N, N, De, generator=g).to(dev).requires_grad_()
    mask = torch.ones(B, N, dtype=torch.bool, device=dev)
    dh = torch.randn(B, N, Dh, generator=g).to(dev); de = torch.randn(B, N, N, De, generator=g).to(dev)
    params = [p for p in blk.parameters()]
    fa = FlatGradAllReduce(params) if use_dist else None

    def step():
        h.grad = e.grad = None
        if fa is not None:
            fa.zero(); fa.rebind()
        else:
            for p in params:
                p.grad = None
        h2, e2 = blk(h, e, mask)
        torch.autograd.backward([h2, e2], [dh, de])
        if fa is not None:
            fa.all_reduce(average=True, force=True)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 6)):                  # rocBLAS picks its kernels on the first calls
        step()
    fence()
    fused_pair = getattr(blk, "last_path", "") == "fused-pair"   # d = 64 / De = 32: the fused pair operator (k_pair_fwd / k_pair_bwd)
    dominant = "k_pair_bwd" if fused_pair else "k_attn_mfma_bwd_kv"
    if not args.no_prof:
        lib.egt_prof_filter(dominant.encode()); lib.egt_prof_enable(2)
    sev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]   # per-step events: the median beside the wall-clock mean
    cur = torch.cuda.current_stream()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sev[i].record(cur)
        step()
    sev[args.steps].record(cur)
    fence()
    elapsed = time.perf_counter() - t0
    lib.egt_prof_enable(0)
    step_ms = sorted(sev[i].elapsed_time(sev[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if step_ms else None
    graphs_step = B
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX); elapsed = float(t.item())
        gs = torch.tensor([B], device=dev, dtype=torch.int64)
        dist.all_reduce(gs); graphs_step = int(gs.item())
    dom_prof = prof_read_all(lib) if not args.no_prof else {}
    prof = {}
    if not args.no_prof:
        lib.egt_prof_filter(b""); lib.egt_prof_enable(2)
        for _ in range(min(args.steps, 10)):
            step()
        fence()
        lib.egt_prof_enable(0)
        prof = prof_read_all(lib)
    if rank != 0:
        return
    pairs = B * N * N
    step_s = elapsed / args.steps
    flops_blk = 3.0 * (6.0 * pairs * De * H + 8.0 * B * N * Dh * Dh + 4.0 * pairs * Dh)
    bytes_blk = B * (5.0 * N * N * De * 4 + 6.0 * N * Dh * 4)
    # algorithmic flops per launch.  The fused pair kernels carry the edge-channel contractions of SURVEY 8(d)'s block count as well:
    # forward LN-folded projections 2 De 16 + dense_edge_r 2 H De per pair, backward twice that (input + weight gradients)
    edge_f = 2.0 * De * 2 * H + 2.0 * H * De
    fl = {"k_attn_mfma_fwd": 4.0 * pairs * Dh, "k_attn_mfma_bwd_kv": 6.0 * pairs * Dh, "k_attn_mfma_bwd_q": 2.0 * pairs * Dh,
          "k_pair_fwd": pairs * (4.0 * Dh + edge_f), "k_pair_bwd": pairs * (6.0 * Dh + 2.0 * edge_f)}
    hbm = {"k_pair_fwd": pairs * De * 4 * 2.0, "k_pair_bwd": pairs * (De * 4 * 3.0 + H * 4.0)}   # e in, e' out / e, de' in, de out + the dA tiles
    roof = None
    if prof and dominant in prof:
        cnt, ms = dom_prof.get(dominant, prof[dominant])
        avg_s = ms / cnt / 1e3
        hip_ms = sum(v[1] / v[0] * (v[0] / min(args.steps, 10)) for v in prof.values())   # HIP kernels of this library per step
        roof = dict(bound="mfma", kernel=dominant, timed_in_region=dominant in dom_prof, achieved=fl[dominant] / avg_s / 1e12,
                    peak=FP32_MFMA_PEAK_TF, unit="TFLOP/s", frac=fl[dominant] / avg_s / 1e12 / FP32_MFMA_PEAK_TF,
                    step_frac=flops_blk * (graphs_step / B) / step_s / 1e12 / FP32_MFMA_PEAK_TF / max(world, 1),
                    step_hbm_frac=bytes_blk / step_s / 1e9 / HBM_PEAK_GBS,
                    traffic=None, avg_launch_us=avg_s * 1e6, launches=cnt, algorithmic_flops_per_launch=fl[dominant],
                    algorithmic_flops_per_step=flops_blk, algorithmic_bytes_per_step=bytes_blk,
                    library_kernels_ms_per_step=hip_ms, other_ms_per_step=step_s * 1e3 - hip_ms,
                    note="other_ms_per_step = the rocBLAS node-side GEMMs, torch glue and launch gaps (not behind the C-ABI's launch profiler)",
                    kernels={k: dict(launches=v[0], avg_us=v[1] / v[0] * 1e3, share_of_step=(v[1] / min(args.steps, 10)) / (step_s * 1e3),
                                     tflops=(fl[k] / (v[1] / v[0] / 1e3) / 1e12) if k in fl else None,
                                     hbm_frac=(hbm[k] / (v[1] / v[0] / 1e3) / 1e9 / HBM_PEAK_GBS) if k in hbm else None)
                             for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])})
        try:   # HBM bytes / matrix-pipe busy per launch of the dominant kernel from the committed counter passes of this workload (static citation)
            pmc = (json.load(open(os.path.join(REPO, "profiles", "pmc_mfma.json"))).get(args.workload) or {}).get(dominant) or {}
            roof["traffic"] = pmc.get("hbm_bytes_per_launch")
            roof["mfma_busy"] = pmc.get("mfma_busy")
            roof["issue"] = pmc.get("issue")
            if pmc:
                roof["pmc_source"] = ("static: profiles/pmc_mfma.json = committed rocprofv3 --pmc passes of this workload (tools/profile_r06.sh; "
                                      "HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024); not re-measured in this run")
        except Exception:  # noqa: BLE001
            pass
        if dominant in hbm:   # the fused pair kernels sit between both roofs: algorithmic HBM bytes of the dominant one beside its flops
            roof["hbm_achieved_GBs"] = hbm[dominant] / avg_s / 1e9
            roof["hbm_frac"] = hbm[dominant] / avg_s / 1e9 / HBM_PEAK_GBS
            roof["algorithmic_bytes_per_launch"] = hbm[dominant]
    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(w, args.cpu_seconds, Bs=1)
    line = {
        "metric": "graphs/sec EGT fwd+bwd, " + METRIC_OF[args.workload],
        "value": graphs_step * args.steps / elapsed, "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": step_s * 1e3, "median_ms_per_step": median_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: ONE attention block (h,e,mask)->(h',e') fwd+bwd + param grads "
                               "(graph_xformer_model_base.py:192-223 around egt_layers.py:57-143), training mode, in-kernel random mask",
                   "scope": "block", "graphs_per_gpu": B, "global_batch": graphs_step, "N": N, "Dh": Dh, "De": De, "H": H, "d": Dh // H,
                   "random_mask_prob": w["rand_p"], "nodes": list(w["nodes"]),
                   "path": ("fused pair operator: k_pair_fwd / k_pair_bwd (LN + gates / edge bias -> MFMA inner op -> dense_edge_r + residual in one pair "
                            "kernel per direction: E, G, H_hat, dE, dG, dH_ext stay in LDS) + k_attn_mfma_bwd_q; node-side Dense = torch library GEMMs (" + blas + ")") if fused_pair else
                           "composed: k_edge_proj (LN + gates / edge bias) -> MFMA inner op -> k_edge_update (dense_edge_r + residual), rocBLAS node-side Dense",
                   "parallelism": f"dp{world}", "backend": "rccl" if use_dist else "none (single process)",
                   "tflops_step": flops_blk * (graphs_step / B) / step_s / 1e12},
        "roofline": roof, "cpu_baseline": cpu,
    }
    print(json.dumps(line))


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(n: int):
    """`python bench.py --gpus N` with no launcher in the environment: become
    `python -m torch.distributed.run --nproc-per-node N bench.py <same args>` (one rank per GPU)."""
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py: --gpus {n} but only {have} GPU(s) visible; refusing to run fewer ranks")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC (RCCL across processes)
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="zinc500k_n64", choices=list(WORKLOADS))
    ap.add_argument("--edge-dtype", default="", choices=["", "f32", "bf16"],
                    help="storage type of the edge tensors (EGT_BF16: bf16 in HBM, fp32 arithmetic); default: the workload's")
    ap.add_argument("--with-ffn", action="store_true",
                    help="whole-layer scope: every attention block is followed by the fused node + edge FFN "
                         "(graph_xformer_model_base.py:336-341); NOT the headline workload")
    ap.add_argument("--scope", default="", choices=["", "stack", "layers", "model"],
                    help="stack (default, the headline): the attention-block stack; layers (= --with-ffn): attention block + "
                         "node/edge FFN per layer; model: the whole ZINC model (embeddings, hop stacking, layers, final norm, "
                         "masked mean pooling, MLP head, MAE loss), SURVEY 8(f)-2 -- NOT the headline workload")
    ap.add_argument("--ffn-matmul", default="f32", choices=["f32", "bf16x3", "bf16"],
                    help="layers / model scopes: how the channel FFN evaluates its matrix products (egt_ffn_desc.matmul): "
                         "f32 exact (default); bf16x3 3-term bfloat16 split on the bf16 matrix pipe, fp32 accumulate, "
                         "per-product error 2^-16 (holds the fp32 parity tolerances); bf16 plain bfloat16 products")
    ap.add_argument("--layers", type=int, default=0, help="override the workload's layer count (1 = single-block scope)")
    ap.add_argument("--fused", default="auto", choices=["auto", "on", "off"])
    ap.add_argument("--cpu-seconds", type=float, default=18.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prof", action="store_true")
    ap.add_argument("--dominant", default="k_block_bwd",
                    help="kernel timed with hipEvents inside the timed region")
    ap.add_argument("--graph", default="auto", choices=["auto", "off", "on"],
                    help="auto (default): for the attention-block stack (the headline scope) a few steps of BOTH step modes are timed after "
                         "the warm-up and the faster one runs the timed region (config.step_mode; the other mode's figure is the line's "
                         "secondary leg); off for the wider scopes.  "
                         "on: forward + backward of the step replayed from ONE captured hipGraph (egt_amd.graph.GraphedStep; the "
                         "random-mask seeds live in device memory, EGT_BF_SEED_DEVICE, and advance inside the graph, so every "
                         "replay draws a fresh sample); the gradient collective stays an eager call after the replay.  The "
                         "dominant kernel is timed by hipEvent-record nodes inside the captured graph (egt_prof_collect_graph)")
    ap.add_argument("--graph-collective", action="store_true",
                    help="with --graph on under a launcher: the flat gradient all-reduce is captured INTO the step's hipGraph (RCCL "
                         "supports stream capture), so a DP step is one host call; default: the collective is an eager call after the replay")
    ap.add_argument("--bind-grads", default="on", choices=["on", "off"],
                    help="stack scope: EGTStack.bind_flat_gradients() -- the stack backward writes every parameter gradient into one "
                         "persistent flat buffer whose views ARE the parameters' .grad (no per-parameter autograd work on the host)")
    ap.add_argument("--no-graph-leg", action="store_true", help="skip the secondary hipGraph-replay figure (hipgraph_replay in the line)")
    ap.add_argument("--overlap-ffn", action="store_true",
                    help="layers / model scopes: the node FFN of a layer runs on a side stream beside the edge FFN (EGTLayerStack.overlap_ffn)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: the workload's B graphs per GPU; strong: B graphs split over the ranks")
    ap.add_argument("--dp-backend", default="torch", choices=["torch", "capi"],
                    help="gradient all-reduce through torch.distributed (RCCL) or the library's own egt_dp_* entry points (RCCL)")
    args = ap.parse_args()
    if args.scope == "layers":
        args.with_ffn = True
    if args.graph == "auto":     # the stack step is ~25 short launches: replayed as one hipGraph (bit-identical to the eager calls,
        args.graph = "calibrate" if (args.scope in ("", "stack") and not args.with_ffn) else "off"   # tests/test_graph_gpu.py); "calibrate": a few steps of
        # both modes are timed after the warm-up and the faster one runs the timed region (`config.step_mode`); the other mode's figure is the
        # line's secondary leg.  Launch-bound small batches favour the replay, the full-size headline batch the eager stream.
    if args.scope == "model" and args.dominant == "k_block_bwd":
        args.dominant = "k_ffn_bwd"

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not launched:
        self_launch(args.gpus)           # does not return
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if env_world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; "
                         "pass matching values (the line must report the GPUs that really ran)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = launched                  # under torch.distributed.run even a 1-rank job goes through RCCL init
    world = 1
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=env_world)  # "nccl" is RCCL on ROCm
        world = dist.get_world_size()    # what RCCL really connected -- this is what the line reports
        if world != args.gpus:
            raise SystemExit(f"bench.py: RCCL world size {world} != --gpus {args.gpus}")
        # fail FAST, with RCCL's own error text, if the communicator does not really span N ranks: one tiny all-reduce of ones
        # (this is also where RCCL builds its rings; without it the first failure would surface inside the timed region)
        try:
            probe = torch.ones(1, device=dev)
            dist.all_reduce(probe)
            torch.cuda.synchronize()
            nranks = int(round(float(probe.item())))
        except Exception as ex:   # torch wraps ncclGetErrorString / ncclGetLastError in the exception text
            raise SystemExit(f"bench.py: RCCL all-reduce probe failed on rank {rank} of {env_world}: {ex}")
        if nranks != args.gpus:
            raise SystemExit(f"bench.py: the RCCL communicator reduced over {nranks} ranks, --gpus says {args.gpus}")

    from egt_amd import build as _build
    if local_rank == 0:
        _build.build()           # no-op when egt_amd/lib/libegt_amd.so is current (hipcc otherwise)
    if use_dist:
        dist.barrier()
    from egt_amd import EGTStack, _lib
    from egt_amd.dp import FlatGradAllReduce, flat_grad_view, all_reduce_flat
    lib = _lib.load()
    comm = None
    if use_dist and args.dp_backend == "capi":
        from egt_amd.dp import CapiComm
        comm = CapiComm()                # ncclCommInitRank behind the C-ABI; the id travels over the process group once
        if comm.world != world:
            raise SystemExit(f"bench.py: egt_dp world {comm.world} != {world}")
        probe = torch.ones(1, device=dev)
        comm.all_reduce_flat(probe, average=False)   # raises RuntimeError carrying ncclGetErrorString (egt_last_error) on failure
        torch.cuda.synchronize()
        if int(round(float(probe.item()))) != args.gpus:
            raise SystemExit(f"bench.py: the egt_dp communicator reduced over {probe.item():.0f} ranks, --gpus says {args.gpus}")

    w = dict(WORKLOADS[args.workload])
    if w.get("scope") in ("core", "block"):
        (run_core if w["scope"] == "core" else run_block)(args, w, dev, lib, rank, world, use_dist)
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return
    if os.environ.get("EGT_BENCH_B"):   # experiments only (batch sweeps): the line's config carries the overridden B
        w["B"] = int(os.environ["EGT_BENCH_B"])
    if args.layers > 0:
        w["Ly"] = args.layers
    if args.edge_dtype:
        w["edge_dtype"] = args.edge_dtype
    bf16 = w.get("edge_dtype", "f32") == "bf16"
    global_B = w["B"] * world
    if args.scaling == "strong":
        from egt_amd.dp import shard_batch
        global_B = w["B"]
        lo, hi = shard_batch(global_B, world, rank)
        w["B"] = hi - lo                 # this rank's contiguous slice of the global batch
        if w["B"] < 1:
            raise SystemExit("bench.py: strong scaling needs at least one graph per rank")
    if w["N"] is None:                   # padded to the per-batch max (each rank pads to ITS shard's largest graph: SURVEY 8(e))
        gN = torch.Generator().manual_seed(1234 + rank)     # the same first draw make_inputs() makes
        w["N"] = int(torch.randint(w["nodes"][0], w["nodes"][1] + 1, (w["B"],), generator=gN).max())
    torch.manual_seed(1234)  # same weights on every rank (replicated parameters)
    mask_seed = 1 * world + rank  # the replicas draw independent random attention masks
    fused = {"auto": "auto", "on": True, "off": False}[args.fused]
    zinc = None
    pattern = args.workload.startswith("pattern")
    cifar = args.workload.startswith("cifar")
    if args.scope == "model":
        from egt_amd import (ZincDCTransformer, PatternDCTransformer, Cifar10DCTransformer, mae_loss, weighted_sparse_xent_loss,
                             class_weights_from_sizes, sparse_xent_loss)
        from types import SimpleNamespace
        # PATTERN: lib/models/sbm_pattern/dc.py (BASELINE config 4); CIFAR10: lib/models/cifar10/dc.py (config 3)
        cls = PatternDCTransformer if pattern else (Cifar10DCTransformer if cifar else ZincDCTransformer)
        model = cls(model_width=w["Dh"], edge_width=w["De"], num_heads=w["H"], model_height=w["Ly"],
                    upto_hop=16, random_mask_prob=w["rand_p"], seed=mask_seed, ffn_matmul=args.ffn_matmul).to(dev).train()
        model.fused_parameters = model.trainable_parameters
        model.grad_holder = SimpleNamespace(flat=None)
        zinc = make_zinc_inputs(w, dev, seed=1234 + rank)
        if pattern:
            gy = torch.Generator().manual_seed(99 + rank)
            zinc.append(torch.randint(0, 2, (w["B"], w["N"]), generator=gy).to(dev))          # node class targets
            zinc.append(class_weights_from_sizes([979220, 209900], device=dev))
            zinc[0] = torch.where(zinc[0] >= 0, zinc[0] % 3, zinc[0])                           # 3 node feature values
        if cifar:     # real-valued superpixel features in the reference's format (padding / non-edges = -1)
            gy = torch.Generator().manual_seed(99 + rank)
            nfi, fmi, adj_, _ = zinc
            real = (nfi >= 0)
            nff = torch.where(real[..., None], torch.rand(w["B"], w["N"], 5, generator=gy).to(dev), torch.tensor(-1.0, device=dev))
            fmf = torch.where(adj_ > 0, torch.rand(w["B"], w["N"], w["N"], generator=gy).to(dev), torch.tensor(-1.0, device=dev))[..., None]
            zinc = [nff, fmf.contiguous(), adj_, torch.randint(0, 10, (w["B"],), generator=gy).to(dev)]
    elif args.with_ffn:
        from egt_amd import EGTLayerStack
        model = EGTLayerStack(model_height=w["Ly"], model_width=w["Dh"], edge_width=w["De"], num_heads=w["H"],
                              random_mask_prob=w["rand_p"], seed=mask_seed, fused=fused,
                              ffn_matmul=args.ffn_matmul).to(dev).train()
        model.fused_parameters = lambda: list(model.parameters())
        from types import SimpleNamespace
        model.grad_holder = SimpleNamespace(flat=None)   # per-block calls: classic flat buffer bound to .grad
    else:
        model = EGTStack(model_height=w["Ly"], model_width=w["Dh"], edge_width=w["De"], num_heads=w["H"],
                         random_mask_prob=w["rand_p"], seed=mask_seed, fused=fused).to(dev).train()
    if args.overlap_ffn:
        (model.layers if args.scope == "model" else model).overlap_ffn = True
    h, e, mask, dh, de = make_inputs(w, dev, seed=1234 + rank)  # each rank its own graphs
    if bf16:
        e, de = e.bfloat16(), de.bfloat16()
    h.requires_grad_(); e.requires_grad_()
    params = model.fused_parameters()
    nbytes = sum(p.numel() for p in params) * 4
    state = {"flat_ok": None, "fa": None, "ar_events": None}
    # uneven strong-scaling shards: weight the local-mean gradients by local/global graphs
    ar_kw = dict(local_count=w["B"], global_count=global_B) if args.scaling == "strong" else {}
    ar_kw["force"] = use_dist            # a launched 1-rank job still issues the RCCL collective

    def compute():
        # fused stack: the backward writes every parameter gradient into one flat buffer whose
        # views autograd adopts as .grad (no per-parameter kernels); the DP collective runs on it.
        # composed path: classic flat buffer pre-bound to .grad (decided on the first warm-up step).
        fa = state["fa"]
        if fa is not None:
            fa.zero(); fa.rebind()
        elif not state.get("bound"):
            for p in params:
                p.grad = None
        if zinc is not None:             # whole model: prediction -> MAE -> backward
            if cifar:
                nf, fm, adj, ycls = zinc
                sparse_xent_loss(model(nf, fm, adj), ycls).backward()
            elif pattern:
                nf, fm, adj, tgt, ycls, cw = zinc
                logits, nmask = model(nf, adj, return_mask=True)
                weighted_sparse_xent_loss(logits, ycls, nmask, cw).backward()
            else:
                nf, fm, adj, tgt = zinc
                mae_loss(model(nf, fm, adj), tgt).backward()
        else:
            h.grad = None; e.grad = None
            h2, e2 = model(h, e, mask)
            torch.autograd.backward([h2, e2], [dh, de])

    def reduce():
        if state["flat_ok"] is None:
            state["flat_ok"] = flat_grad_view(params, model.grad_holder.flat)
            if not state["flat_ok"]:
                state["fa"] = FlatGradAllReduce(params, direct=True)   # takes effect from the next step; fused backwards write their gradients straight into it
                return
        fa = state["fa"]
        evs = state["ar_events"]
        if evs is not None:              # hipEvents around the collective (untimed pass only)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if comm is not None:
            comm.all_reduce_flat(model.grad_holder.flat if state["flat_ok"] else fa.flat, True,
                                 ar_kw.get("local_count"), ar_kw.get("global_count"))
        elif state["flat_ok"]:
            all_reduce_flat(model.grad_holder.flat, average=True, **ar_kw)
        else:
            fa.all_reduce(average=True, **ar_kw)
        if evs is not None:
            e1.record()
            evs.append((e0, e1))

    seeds = None
    if args.graph == "on":               # from here on every EGT module reads its mask seed from HBM (eager steps too)
        from egt_amd import DeviceSeeds
        seeds = DeviceSeeds.attach(model, dev)
    step_mode = None

    def eager_step():
        if seeds is not None:
            seeds.advance()
        compute()
        reduce()

    step = eager_step

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    if args.bind_grads == "on" and zinc is None and not args.with_ffn and state["flat_ok"] and getattr(model, "last_path", "") == "fused-stack":
        model.bind_flat_gradients()      # from here on .grad is never reset: the backward overwrites the bound buffer
        state["bound"] = True
        for _ in range(2):
            step()
        fence()
        assert flat_grad_view(params, model.grad_holder.flat)
    graphed = None

    def time_steps(fn, n):               # calibration: n fenced steps, MAX over ranks (every rank must pick the same mode)
        fence()
        t_ = time.perf_counter()
        for _ in range(n):
            fn()
        fence()
        dt = time.perf_counter() - t_
        if use_dist:
            tt = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt / n * 1e3

    eager_ms = None
    if args.graph == "calibrate":
        ncal = max(5, min(20, args.steps))
        eager_ms = time_steps(eager_step, ncal)      # host-side mask seeds, one host call per kernel
        from egt_amd import DeviceSeeds
        seeds = DeviceSeeds.attach(model, dev)
        for _ in range(2):
            eager_step()
        fence()
    # hipEvents around the dominant kernel INSIDE the captured graph: launches captured while the profile is enabled carry their
    # events as external event-record nodes (egt_prof_collect_graph), so the dominant kernel is timed inside the timed region in the
    # replay mode too.  Every GRAPH_STRIDE-th launch of it (of the 10 per step: launches 0 and 7 -- an event-record node costs the
    # replay ~5 us, four pairs per step were 2 % of the headline step).
    GRAPH_STRIDE = 7
    if args.graph in ("on", "calibrate"):
        from egt_amd import GraphedStep
        if not args.no_prof:
            lib.egt_prof_forget_graphs()
            lib.egt_prof_filter(args.dominant.encode())
            lib.egt_prof_stride(GRAPH_STRIDE)
            lib.egt_prof_enable(2)
        in_graph = bool(args.graph_collective and use_dist and state["flat_ok"] is not None and (state["flat_ok"] or state["fa"] is not None))
        if in_graph:                     # forward + backward + the RCCL all-reduce of the flat gradient buffer: one graph, one host call

            def compute_and_reduce():
                compute()
                reduce()
            graphed = GraphedStep(compute_and_reduce, seeds, warmup=1)
        else:
            graphed = GraphedStep(compute, seeds, warmup=1)
        state["collective_in_graph"] = in_graph
        lib.egt_prof_enable(0)           # (the event nodes stay inside the graph)
        lib.egt_prof_stride(1)

        def step():                      # ONE host call for forward + backward (and the collective when captured), else the eager collective
            graphed.replay()
            if not in_graph:
                reduce()
        for _ in range(3):
            step()
        fence()
        if args.graph == "calibrate":
            graph_ms = time_steps(step, ncal)
            # both modes a second time, alternating, the faster figure of each counts: one slow calibration window (seen: a replay
            # window at 1.6 x its usual time on a ZINC-100K run) must not pick the mode of the whole timed region
            eager_ms = min(eager_ms, time_steps(eager_step, ncal))
            graph_ms = min(graph_ms, time_steps(step, ncal))
            # the replay has to win by more than 1 %: on the eager stream the dominant kernel's hipEvents sit INSIDE the timed region
            # (roofline.timed_in_region), a replay has no per-launch host hooks and the kernel is timed in the untimed eager pass
            use_graph = graph_ms < 0.99 * eager_ms
            step_mode = dict(chosen="graph" if use_graph else "eager", eager_ms_per_step=eager_ms, graph_ms_per_step=graph_ms,
                             calibration_steps=ncal, rule="hipGraph replay if it is more than 1 % faster than the eager stream (the faster of two windows each), else eager; "
                                                          "outputs are bit-identical; the dominant kernel is timed inside the timed region in both modes")
            if not use_graph:                # back to host-side seeds and one host call per kernel
                lib.egt_prof_forget_graphs()
                seeds.detach()
                seeds = None
                graphed = None
                step = eager_step
                for _ in range(2):
                    step()
                fence()
    # Timed region.  hipEvents bracket ONLY the dominant kernel's launches here (an event pair
    # around every launch costs ~15% of the step), and of those every PROF_STRIDE-th launch: an event pair costs the stream
    # 4-5 us (ten of them per step were 2.5 % of the headline step), and a stride coprime with the launches per step walks
    # through the layers.  The per-kernel table comes from a second, untimed pass over the same steps below.
    PROF_STRIDE = 13
    COLLECT_EVERY = 5                    # replay mode: the graph's event pairs are read after every 5th replay (one stream sync each)
    if not args.no_prof:
        lib.egt_prof_filter(args.dominant.encode())
        lib.egt_prof_stride(PROF_STRIDE)
        lib.egt_prof_enable(2)           # (reset: counts and un-read eager pairs; event pairs inside captured graphs are kept)
    graph_prof = graphed is not None and not args.no_prof
    # per-step hipEvents on the launch stream: SURVEY 8(d) asks for the MEDIAN step time beside the wall-clock mean
    sev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    cur = torch.cuda.current_stream()
    t0 = time.perf_counter()
    for i in range(args.steps):
        sev[i].record(cur)
        step()
        if graph_prof and (i % COLLECT_EVERY == COLLECT_EVERY - 1 or i == args.steps - 1):
            cur.synchronize()
            lib.egt_prof_collect_graph(0)
    sev[args.steps].record(cur)
    fence()
    elapsed = time.perf_counter() - t0
    lib.egt_prof_enable(0)
    lib.egt_prof_stride(1)
    step_ms = sorted(sev[i].elapsed_time(sev[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if step_ms else None
    graphs_step = w["B"]
    if use_dist:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        gs = torch.tensor([w["B"]], device=dev, dtype=torch.int64)
        dist.all_reduce(gs)              # graphs all ranks processed per step
        graphs_step = int(gs.item())
    else:
        graphs_step = w["B"]
    dom_prof = prof_read_all(lib) if not args.no_prof else {}
    all_prof = {}
    table_mode = "untimed eager pass, hipEvents around every launch"
    if not args.no_prof:   # every rank runs it (the step contains the collective)
        lib.egt_prof_filter(b"")
        if graphed is not None:
            # replay mode: the per-kernel table comes from a SECOND captured graph of the same step with an event pair around EVERY
            # launch, replayed and read ten times -- kernel durations as they are inside a replay, not inside an eager stream
            from egt_amd import GraphedStep
            lib.egt_prof_forget_graphs()
            lib.egt_prof_enable(2)
            pg = GraphedStep(graphed.fn, seeds, warmup=1)
            lib.egt_prof_enable(2)       # drop the warm-up's eager pairs; the graph's pairs stay
            lib.egt_prof_enable(0)
            for _ in range(min(args.steps, 10)):
                pg.replay()
                if not state.get("collective_in_graph"):
                    reduce()
                torch.cuda.current_stream().synchronize()
                lib.egt_prof_collect_graph(0)
            fence()
            all_prof = prof_read_all(lib)
            lib.egt_prof_forget_graphs()
            del pg
            table_mode = "untimed replays of a second captured graph with an event-record pair around every launch"
        else:
            lib.egt_prof_enable(2)
            for _ in range(min(args.steps, 10)):
                eager_step()
            fence()
            lib.egt_prof_enable(0)
            all_prof = prof_read_all(lib)
    pair_ov = event_pair_overhead_us() if (all_prof and graphed is None) else None   # (eager table only: the replay table's pairs are graph nodes)
    # the one exchange step, timed on its own: hipEvents around the flat-buffer collective
    ar_us = None
    if use_dist:
        state["ar_events"] = []
        for _ in range(min(args.steps, 10)):
            eager_step()
        fence()
        ts = sorted(a.elapsed_time(b) * 1e3 for a, b in state["ar_events"])
        state["ar_events"] = None
        ar_us = ts[len(ts) // 2] if ts else None

    # Secondary figure (never `value`): the same workload with forward + backward replayed from ONE hipGraph per step
    # (egt_amd.graph: device-resident mask seeds, fresh sample per replay).  Its own process: nothing it does can cost
    # the contract line.  Single-process runs only.
    graph_leg = None
    if not args.no_graph_leg and not use_dist and rank == 0:
        try:
            import subprocess
            cmd = [sys.executable, os.path.abspath(__file__), "--graph", "off" if graphed is not None else "on", "--no-cpu-baseline", "--no-prof", "--no-graph-leg",
                   "--steps", str(args.steps), "--warmup", str(args.warmup), "--workload", args.workload, "--ffn-matmul", args.ffn_matmul,
                   "--fused", args.fused]
            if args.scope:
                cmd += ["--scope", args.scope]
            if args.with_ffn and not args.scope:
                cmd += ["--with-ffn"]
            if args.layers:
                cmd += ["--layers", str(args.layers)]
            if args.edge_dtype:
                cmd += ["--edge-dtype", args.edge_dtype]
            if args.overlap_ffn:
                cmd += ["--overlap-ffn"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
            sub = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and sub:
                d = json.loads(sub[-1])
                graph_leg = dict(value=d["value"], unit=d["unit"], ms_per_step=d["ms_per_step"], steps=d["steps"],
                                 what="same workload in a second process: " + (d["config"]["hipgraph"] or "eager launches (one host call per kernel)")
                                      + "; replays are bit-identical to the eager calls (tests/test_graph_gpu.py)")
            else:
                graph_leg = dict(error=f"rc={r.returncode}: {r.stderr[-200:]}")
        except Exception as ex:  # noqa: BLE001  (a secondary figure must never cost the contract line)
            graph_leg = dict(error=f"{type(ex).__name__}: {ex}"[:300])

    prof = all_prof
    if rank == 0:
        roof = None
        if prof:
            dom = max(prof, key=lambda k: prof[k][1])
            if dom in dom_prof:      # the figure measured inside the timed region
                cnt, ms = dom_prof[dom]
            else:
                cnt, ms = prof[dom]
            avg_s = ms / cnt / 1e3
            ab = algorithmic_bytes(dom, w)
            ach = ab / avg_s / 1e9 if avg_s > 0 and ab > 0 else None
            traffic = None   # PMC HBM bytes per launch of this kernel, from the committed rocprofv3 passes
            try:
                pt = json.load(open(os.path.join(REPO, "profiles", "pmc_traffic.json")))
                wl = args.workload + ("" if not args.edge_dtype or args.edge_dtype == WORKLOADS[args.workload].get("edge_dtype", "f32") else "@" + args.edge_dtype)
                traffic = (pt.get(wl) or {}).get(dom) if args.scope in ("", "stack") and args.layers == 0 else None
            except Exception:
                traffic = None
            # the OTHER roof of an exact-fp32 pair kernel: instruction issue.  MFMA cycles + VALU cycles per launch over
            # SIMDs x active cycles, and the matrix-pipe busy fraction, from the committed counter pass of this workload (static)
            issue = mfma_busy = None
            try:
                pm = (json.load(open(os.path.join(REPO, "profiles", "pmc_mfma.json"))).get(args.workload) or {}).get(dom) or {}
                if args.scope in ("", "stack") and args.layers == 0 and not args.edge_dtype:
                    issue, mfma_busy = pm.get("issue"), pm.get("mfma_busy")
            except Exception:  # noqa: BLE001
                pass
            nprof = min(args.steps, 10)
            fl = algorithmic_flops(dom, w, args.scope)
            if fl > 0:    # MFMA-bound dominant kernel (whole-layer / whole-model scopes): fp32 matrix peak
                tot_s = prof[dom][1] / 1e3 / nprof        # this kernel's time per step (all its launches)
                roof = dict(bound="mfma", kernel=dom, timed_in_region=False, achieved=fl / tot_s / 1e12, peak=157.3,
                            unit="TFLOP/s", frac=fl / tot_s / 1e12 / 157.3, traffic=None,
                            avg_launch_us=avg_s * 1e6, launches=cnt, algorithmic_flops_per_step=fl,
                            kernels={k: dict(launches=v[0], avg_us=v[1] / v[0] * 1e3,
                                             share=v[1] / sum(x[1] for x in prof.values()))
                                     for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])})
            else:
              roof = dict(bound="hbm", kernel=dom, timed_in_region=dom in dom_prof, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s",
                        kernels_table=table_mode,
                        # sum of (launches x mean) over the table: what the step's kernels AND their launch gaps add up to, plus one event
                        # pair's own cost per launch (an empty pair measures it: event_pair_overhead_us); `net` has that cost taken out
                        kernels_sum_ms_per_step=sum(v[1] for v in prof.values()) / nprof,
                        event_pair_overhead_us=pair_ov,
                        kernels_sum_net_ms_per_step=(sum(v[1] - v[0] * pair_ov * 1e-3 for v in prof.values()) / nprof) if pair_ov is not None else None,
                        frac=(ach / HBM_PEAK_GBS) if ach else None,
                        achievable_peak=HBM_ACHIEVABLE_GBS, frac_of_achievable=(ach / HBM_ACHIEVABLE_GBS) if ach else None,   # the 6.3 TB/s a streaming copy sustains (MI355X_MICROARCH.md, HBM)
                        traffic=traffic, issue=issue, mfma_busy=mfma_busy,
                        traffic_source=("static: profiles/pmc_traffic.json = HBM bytes per launch from the committed rocprofv3 --pmc "
                                        "FETCH_SIZE / WRITE_SIZE passes of this workload (FETCH doubled per the gfx950 note); not re-measured in this run"
                                        if traffic is not None else None),
                        avg_launch_us=avg_s * 1e6, launches=cnt, algorithmic_bytes_per_launch=ab,
                        launches_sampled=((f"hipEvent-record nodes around every {GRAPH_STRIDE}th launch of the kernel inside the captured graph, read "
                                           f"after every {COLLECT_EVERY}th replay of the timed region") if (dom in dom_prof and graphed is not None) else
                                          f"hipEvents around every {PROF_STRIDE}th launch of the kernel inside the timed region" if dom in dom_prof
                                          else "every launch of the untimed pass"),
                        kernels={k: dict(launches=v[0], avg_us=v[1] / v[0] * 1e3,
                                         share=v[1] / sum(x[1] for x in prof.values()))
                                 for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])})
        cpu = None
        if not args.no_cpu_baseline and world == 1:   # the CPU leg is timed at N = 1 only (the other ranks would idle at the last barrier)
            if args.scope == "model" and cifar:
                cpu = None      # (no CPU leg for the CIFAR10 model scope: not a bench line of BASELINE.json's metric)
            else:
                cpu = cpu_baseline_model(w, args.cpu_seconds, pattern=args.workload.startswith("pattern")) if args.scope == "model" else cpu_baseline(w, args.cpu_seconds)
        graphs = graphs_step * args.steps
        path = "fused-stack" if state["flat_ok"] else ("fused" if any(k.startswith("k_block") for k in prof) else "composed")
        if args.with_ffn:
            path += "+ffn"
        if args.scope == "model":
            path = ("pattern" if pattern else "cifar10" if cifar else "zinc") + "-model (HIP edge embedding + fused blocks + fused FFNs, torch node-side head)"
        line = {
            "metric": "graphs/sec EGT fwd+bwd, " + METRIC_OF[args.workload]
                      + ("" if (args.scope or "stack") == "stack" and not args.with_ffn else f" ({args.scope or 'layers'} scope)"),
            "value": graphs / elapsed, "unit": "graphs/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "median_ms_per_step": median_ms,     # per-step hipEvents on the launch stream; `value` stays the wall-clock figure of the contract
            "value_at_median": (graphs_step / (median_ms * 1e-3)) if median_ms else None,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f32 (edge tensors stored bf16)" if bf16 else "f32",
            "data": "synthetic",
            "config": {"workload": (f"{args.workload}: " + {"model": ("WHOLE PATTERN MODEL (node embedding, 16-hop adjacency embedding, Ly x [attention block + node/edge FFN], "
                                                                       "final norm, per-node MLP head, class-weighted x-ent) fwd+bwd" if args.workload.startswith("pattern") else
                                                                       "WHOLE CIFAR10 MODEL (Masking+Dense node / edge embeddings, 16-hop adjacency embedding, Ly x [attention block + "
                                                                       "node/edge FFN], final norm, masked mean pool, MLP head, sparse x-ent) fwd+bwd" if args.workload.startswith("cifar") else
                                                                       "WHOLE ZINC MODEL (embeddings, 16-hop stacking, Ly x [attention block + node/edge FFN], "
                                                                       "final norm, masked mean pool, MLP head, MAE) fwd+bwd"),
                                                             "layers": "Ly x [attention block + node/edge FFN] fwd+bwd"}.get(
                                        args.scope or ("layers" if args.with_ffn else "stack"),
                                        "attention-block stack (h,e,mask)->(h',e') x Ly, fwd+bwd")
                                    + " + param grads" + (" + flat RCCL grad all-reduce" if world > 1 else "")),
                       "scope": args.scope or ("layers" if args.with_ffn else "stack"),
                       "ffn_matmul": args.ffn_matmul if (args.with_ffn or args.scope == "model") else None,
                       "graphs_per_gpu": w["B"], "global_batch": graphs_step, "N": w["N"],
                       "Dh": w["Dh"], "De": w["De"], "H": w["H"], "d": w["Dh"] // w["H"], "Ly": w["Ly"],
                       "random_mask_prob": w["rand_p"], "nodes": list(w["nodes"]), "path": path,
                       "parallelism": f"dp{world}", "grad_allreduce_bytes": nbytes,
                       "grad_allreduce_us": ar_us, "backend": ("rccl (egt_dp_* C-ABI)" if comm is not None else "rccl") if use_dist else "none (single process)",
                       "rccl_nranks": (comm.world if comm is not None else dist.get_world_size()) if use_dist else None,   # what the communicator reports
                       "flat_grad_adopted": bool(state["flat_ok"]), "flat_grad_bound": bool(state.get("bound")),
                       "hipgraph": (f"forward + backward{' + the gradient all-reduce' if state.get('collective_in_graph') else ''} replayed from one "
                                    f"captured hipGraph ({graphed.replays} replays), device-resident "
                                    "mask seeds; dominant kernel timed by event-record nodes inside the graph") if graphed is not None else None,
                       "step_mode": step_mode,
                       "collective_in_graph": bool(state.get("collective_in_graph"))},
            "roofline": roof, "cpu_baseline": cpu, ("eager_step" if graphed is not None else "hipgraph_replay"): graph_leg,
        }
        print(json.dumps(line))
    if use_dist:
        dist.barrier()
        if comm is not None:
            comm.close()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
