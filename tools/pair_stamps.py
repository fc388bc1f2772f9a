#!/usr/bin/env python
"""Phase stamps of the fused pair kernels (build with EGT_ATTN_FLAGS=-DEGT_ATTN_STAMPS): per-phase s_memtime cycles of the eight
waves of workgroup 0 (waves 0-3 attention, 4-7 edge), summed over the trips of one launch.  Usage: EGT_ATTN_FLAGS=-DEGT_ATTN_STAMPS python tools/pair_stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egt_amd import EGTBlock, _lib
B, N, Dh, De = 8, 512, 512, 32
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(1234)
blk = EGTBlock(model_width=Dh, edge_width=De, num_heads=8, random_mask_prob=0.1).to(dev).train()
h = torch.randn(B, N, Dh, generator=g).to(dev).requires_grad_(); e = torch.randn(B, N, N, De, generator=g).to(dev).requires_grad_()
mask = torch.ones(B, N, dtype=torch.bool, device=dev)
dh = torch.randn(B, N, Dh, generator=g).to(dev); de = torch.randn(B, N, N, De, generator=g).to(dev)
for _ in range(3):
    h.grad = e.grad = None
    h2, e2 = blk(h, e, mask); torch.autograd.backward([h2, e2], [dh, de])
torch.cuda.synchronize()
assert blk.last_path == "fused-pair"
lib = C.CDLL(_lib.load()._name)
buf = (C.c_longlong * (3 * 8 * 16))()
assert lib.egt_attn_mfma_read_stamps(buf, 3 * 8 * 16) == 0
NAMES = {
    0: (["wait K + loads + S: 32 MFMA", "DMA K + softmax / gates / H_hat", "wait V + A.V: 32 MFMA", "DMA V + barrier", "-", "-", "-", "(setup)"],
        ["dense_edge_r of tile it-1", "e requests", "LN + projections", "barrier", "-", "-", "-", "(setup)", "last update"]),
    1: (["wait stage + loads + S, dP: 32 MFMA", "softmax bwd + planes + dA", "dV, dK: 32 MFMA", "DMA next (+ barrier)", "-", "-", "-", "-", "(setup)"],
        ["de' reload issue", "POST 4 rows + e, de' requests", "-", "-", "stat request", "PRE 4 rows", "stat put", "barrier", "(setup)", "last POST"]),
}
for k, kn in ((0, "k_pair_fwd"), (1, "k_pair_bwd")):
    print(kn)
    for role, ws in ((0, range(0, 4)), (1, range(4, 8))):
        print("  attention waves" if role == 0 else "  edge waves")
        tot = [0] * 4
        for i, nm in enumerate(NAMES[k][role]):
            row = [buf[(k * 8 + w) * 16 + i] for w in ws]
            tot = [a + b for a, b in zip(tot, row)]
            print(f"    {nm:34s} " + " ".join(f"{v:9d}" for v in row))
        print(f"    {'total':34s} " + " ".join(f"{v:9d}" for v in tot))
