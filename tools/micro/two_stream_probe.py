#!/usr/bin/env python3
"""Probe (round 5): does the headline step run faster as TWO half-batches on two HIP streams (their pre-loop / epilogue phases, during
which nothing streams, overlapping the other half's row loops) than as one batch whose 512 workgroups run in lock step?
    python tools/micro/two_stream_probe.py [B] [steps]
Prints graphs/s of: one stream B graphs; two streams B/2 + B/2; two streams with an uneven split (so the phases drift)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from egt_amd import EGTStack  # noqa: E402


def make(B, N=64, Ly=10, seed=0, dev="cuda:0"):
    torch.manual_seed(7)
    st = EGTStack(model_height=Ly, model_width=64, edge_width=64, num_heads=8, random_mask_prob=0.1, seed=5, fused=True).to(dev).train()
    g = torch.Generator().manual_seed(seed)
    h = torch.randn(B, N, 64, generator=g).to(dev).requires_grad_()
    e = torch.randn(B, N, N, 64, generator=g).to(dev).requires_grad_()
    nodes = torch.randint(9, 38, (B,), generator=g)
    mask = (torch.arange(N)[None, :] < nodes[:, None]).to(dev)
    dh = torch.randn(B, N, 64, generator=g).to(dev); de = torch.randn(B, N, N, 64, generator=g).to(dev)
    return st, h, e, mask, dh, de


def step(p):
    st, h, e, mask, dh, de = p
    h.grad = None; e.grad = None
    for q in st.parameters():
        q.grad = None
    h2, e2 = st(h, e, mask)
    torch.autograd.backward([h2, e2], [dh, de])


def run_eager(parts, streams, steps):
    for _ in range(5):
        for p, s in zip(parts, streams):
            with torch.cuda.stream(s):
                step(p)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for p, s in zip(parts, streams):
            with torch.cuda.stream(s):
                step(p)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return sum(p[1].shape[0] for p in parts) * steps / dt


def run(parts, streams, steps):
    """every part's step captured into its own hipGraph (the eager half-batch steps are host-bound), replayed on the given streams"""
    from egt_amd import DeviceSeeds, GraphedStep
    gs = []
    for p in parts:
        seeds = DeviceSeeds.attach(p[0], "cuda:0")
        gs.append(GraphedStep(lambda p=p: step(p), seeds, warmup=2))
    for _ in range(5):
        for g, s in zip(gs, streams):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for g, s in zip(gs, streams):
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for p in parts:
        for m in p[0].modules():
            if hasattr(m, "seed_device"):
                m.seed_device = None
    return sum(p[1].shape[0] for p in parts) * steps / dt


def run_forked(parts, steps):
    """both halves inside ONE hipGraph as two parallel branches (fork / join on a second stream during capture)"""
    from egt_amd import DeviceSeeds, GraphedStep
    mods = torch.nn.ModuleList([p[0] for p in parts])
    seeds = DeviceSeeds.attach(mods, "cuda:0")
    side = torch.cuda.Stream()

    def both():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            step(parts[1])
        step(parts[0])
        cur.wait_stream(side)
    g = GraphedStep(both, seeds, warmup=2)
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for m in mods.modules():
        if hasattr(m, "seed_device"):
            m.seed_device = None
    return sum(p[1].shape[0] for p in parts) * steps / dt


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    one = make(B)
    print(f"one stream, B = {B}: {run([one], [s0], steps):.0f} graphs/s")
    del one
    a, b = make(B // 2, seed=1), make(B // 2, seed=2)
    print(f"two streams, {B // 2} + {B // 2}: {run([a, b], [s0, s1], steps):.0f} graphs/s")
    print(f"   (the same two halves on ONE stream: {run([a, b], [s0, s0], steps):.0f} graphs/s)")
    try:
        print(f"   both halves as two parallel branches of ONE hipGraph: {run_forked([a, b], steps):.0f} graphs/s")
    except Exception as ex:  # noqa: BLE001
        print("   forked capture failed:", type(ex).__name__, str(ex)[:200])
    del a, b
    na = (B * 5 // 8)
    a, b = make(na, seed=1), make(B - na, seed=2)
    print(f"two streams, {na} + {B - na}: {run([a, b], [s0, s1], steps):.0f} graphs/s")
    print(f"one stream again: {run([make(B)], [s0], steps):.0f} graphs/s")
    del a, b
    # eager launches (GPU-bound only when a part has >= 128 graphs)
    one = make(B)
    print(f"eager, one stream, B = {B}: {run_eager([one], [s0], steps):.0f} graphs/s")
    del one
    a, b = make(B // 2, seed=1), make(B // 2, seed=2)
    print(f"eager, two streams, {B // 2} + {B // 2}: {run_eager([a, b], [s0, s1], steps):.0f} graphs/s")
    print(f"eager, the two halves on one stream: {run_eager([a, b], [s0, s0], steps):.0f} graphs/s")


if __name__ == "__main__":
    main()
