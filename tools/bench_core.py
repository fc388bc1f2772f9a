#!/usr/bin/env python
"""Core-op scope of SURVEY §8(d): ([QKV,E,G],mask) -> (V_att,H_hat) forward+backward through
egt_attn_fwd/bwd, per-kernel hipEvent times.  Usage: python tools/bench_core.py [config]"""
import ctypes as C, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egt_amd import egt_attention, AttnConfig, _lib

CONFIGS = {"cfg2": dict(B=128, N=64, H=8, d=8), "cfg5": dict(B=8, N=512, H=8, d=64),
           "cfg4": dict(B=16, N=120, H=8, d=8), "cfg5_b32": dict(B=32, N=512, H=8, d=64)}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
    c = CONFIGS[name]
    B, N, H, d = c["B"], c["N"], c["H"], c["d"]
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1234)
    qkv = torch.randn(B, N, 3 * d * H, generator=g).to(dev).requires_grad_()
    E = torch.randn(B, N, N, H, generator=g).to(dev).requires_grad_()
    G = torch.randn(B, N, N, H, generator=g).to(dev).requires_grad_()
    mask = torch.ones(B, N, dtype=torch.bool, device=dev)
    dV = torch.randn(B, N, d * H, generator=g).to(dev)
    dH = torch.randn(B, N, N, H, generator=g).to(dev)
    cfg = AttnConfig(num_heads=H)
    lib = _lib.load()

    def step():
        qkv.grad = E.grad = G.grad = None
        V, Hh, _ = egt_attention(qkv, E, G, None, mask, cfg=cfg)
        torch.autograd.backward([V, Hh], [dV, dH])

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    lib.egt_prof_filter(b""); lib.egt_prof_enable(2)
    K = 10
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    lib.egt_prof_enable(0)
    buf = C.create_string_buffer(4096); lib.egt_prof_names(buf, 4096)
    ks = {}
    for nm in buf.value.decode().split():
        cnt, ms = C.c_int64(0), C.c_double(0.0)
        lib.egt_prof_read(nm.encode(), C.byref(cnt), C.byref(ms))
        if cnt.value:
            ks[nm] = round(ms.value / cnt.value * 1e3, 1)
    s = 4
    bytes_core = B * (8 * N * N * H * s + 8 * N * d * H * s)
    flops_core = B * 12 * N * N * d * H
    print(json.dumps(dict(config=name, shape=c, ms_per_step=dt * 1e3, graphs_per_s=B / dt,
                          algorithmic_GBps=bytes_core / dt / 1e9, TFLOPs=flops_core / dt / 1e12,
                          kernels_us=ks)))


if __name__ == "__main__":
    main()
