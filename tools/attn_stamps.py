#!/usr/bin/env python
"""Phase stamps of the MFMA inner-op kernels (build with EGT_ATTN_FLAGS=-DEGT_ATTN_STAMPS): per-phase s_memtime
cycles of the eight waves of workgroup 0, one launch.  Usage: EGT_ATTN_FLAGS=-DEGT_ATTN_STAMPS python tools/attn_stamps.py [cfg]
Caveat (round 4): hipcc rotates the forward's loop -- the S MFMAs open the loop body and the P.V MFMAs follow the barrier -- so the
forward's per-phase figures are NOT phase times (the MFMAs are not where the source has them); use the ablation builds
(-DEGT_ATTN_ABL=<bits>) for attribution and the stamps for totals / the loader waves."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from egt_amd import egt_attention, AttnConfig, _lib
from tools.bench_core import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
c = CONFIGS[name]; B, N, H, d = c["B"], c["N"], c["H"], c["d"]
dev = torch.device("cuda:0"); g = torch.Generator().manual_seed(1234)
qkv = torch.randn(B, N, 3 * d * H, generator=g).to(dev).requires_grad_()
E = torch.randn(B, N, N, H, generator=g).to(dev).requires_grad_(); G = torch.randn(B, N, N, H, generator=g).to(dev).requires_grad_()
mask = torch.ones(B, N, dtype=torch.bool, device=dev)
dV = torch.randn(B, N, d * H, generator=g).to(dev); dH = torch.randn(B, N, N, H, generator=g).to(dev)
cfg = AttnConfig(num_heads=H)
for _ in range(3):
    qkv.grad = E.grad = G.grad = None
    V, Hh, _ = egt_attention(qkv, E, G, None, mask, cfg=cfg); torch.autograd.backward([V, Hh], [dV, dH])
torch.cuda.synchronize()
lib = C.CDLL(_lib.load()._name)
buf = (C.c_longlong * (3 * 8 * 16))()
assert lib.egt_attn_mfma_read_stamps(buf, 3 * 8 * 16) == 0
names = ["setup", "top: pair loads, out stores, LDS reads", "MFMA 1 + operand reloads", "elementwise", "MFMA 2 + reloads", "scatter", "barrier", "epilogue"]
for k, kn in enumerate(["fwd", "bwd_kv"]):
    print(kn)
    for i in range(8):
        row = [buf[(k * 8 + w) * 16 + i] for w in range(8)]
        print(f"  {names[i]:42s} " + " ".join(f"{v:8d}" for v in row) + f"   mean {sum(row)/8:9.0f}")
    tot = [sum(buf[(k * 8 + w) * 16 + i] for i in range(8)) for w in range(8)]
    print(f"  {'total':42s} " + " ".join(f"{v:8d}" for v in tot))
