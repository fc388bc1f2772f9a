#!/usr/bin/env python
"""Block scope of SURVEY §8(d) at BASELINE config 5 (B=8, N=512, Dh=512, De=32, H=8, d=64, fp32):
(h, e, mask) -> (h', e') forward+backward of ONE attention block.  d=64 is outside the fused pair
kernels (built for d=8), so this is the composed path: HIP edge projections + MFMA inner op +
rocBLAS node-side Dense.  Usage: python tools/bench_block_cfg5.py [B]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egt_amd import EGTBlock


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    N, Dh, De, H = 512, 512, 32, 8
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1234)
    blk = EGTBlock(model_width=Dh, edge_width=De, num_heads=H, random_mask_prob=0.1).to(dev).train()
    h = torch.randn(B, N, Dh, generator=g).to(dev).requires_grad_()
    e = torch.randn(B, N, N, De, generator=g).to(dev).requires_grad_()
    mask = torch.ones(B, N, dtype=torch.bool, device=dev)
    dh = torch.randn(B, N, Dh, generator=g).to(dev)
    de = torch.randn(B, N, N, De, generator=g).to(dev)

    def step():
        h.grad = e.grad = None
        for p in blk.parameters():
            p.grad = None
        h2, e2 = blk(h, e, mask)
        torch.autograd.backward([h2, e2], [dh, de])

    for _ in range(6):   # rocBLAS picks its kernels on the first calls
        step()
    torch.cuda.synchronize()
    K = 20
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    s = 4
    bytes_blk = B * (5 * N * N * De * s + 6 * N * Dh * s)
    flops_blk = B * 3 * (6 * N * N * De * H + 8 * N * Dh * Dh + 4 * N * N * Dh)
    print(json.dumps(dict(scope="block cfg5", shape=dict(B=B, N=N, Dh=Dh, De=De, H=H, d=Dh // H), path="composed",
                          ms_per_step=dt * 1e3, graphs_per_s=B / dt, algorithmic_GBps=bytes_blk / dt / 1e9,
                          TFLOPs=flops_blk / dt / 1e12, frac_of_157_3=flops_blk / dt / 157.3e12)))


if __name__ == "__main__":
    main()
