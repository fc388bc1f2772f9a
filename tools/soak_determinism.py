#!/usr/bin/env python3
"""Run the headline stack fwd+bwd many times on the same inputs and require bit-identical outputs and
gradients every time (a race between workgroups -- e.g. on the layer-parity partial buffers of the
backward prologue -- would show up as run-to-run differences)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egt_amd import EGTStack  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    # optional: workload = "zinc" (headline, default) | "cifar" (config 3: N = 150, De = 8, bf16 -- the De = 8 kernels with the
    # balanced ranges and their shared-tile hand-off through the dkvp slot) | "cifar32" (the same in fp32) |
    # "pattern16" (config 4 as specified: B = 16, N = 120 -- 8-row backward workgroups, 8-wave forward workgroups)
    wl = sys.argv[2] if len(sys.argv) > 2 else "zinc"
    torch.manual_seed(0)
    B, N, Ly, De, lo, hi = (128, 64, 10, 64, 9, 38) if wl == "zinc" else (16, 120, 4, 8, 44, 121) if wl == "pattern16" else (64, 150, 4, 8, 85, 151)
    st = EGTStack(model_height=Ly, model_width=64, edge_width=De, num_heads=8, random_mask_prob=0.1, seed=3,
                  fused=True).to(dev).train()
    g = torch.Generator().manual_seed(1)
    h = torch.randn(B, N, 64, generator=g).to(dev).requires_grad_()
    e = torch.randn(B, N, N, De, generator=g)
    de = torch.randn(B, N, N, De, generator=g)
    if wl == "cifar":
        e, de = e.bfloat16(), de.bfloat16()
    e = e.to(dev).requires_grad_(); de = de.to(dev)
    n = torch.randint(lo, hi, (B,), generator=g)
    mask = (torch.arange(N)[None] < n[:, None]).to(dev)
    dh = torch.randn(B, N, 64, generator=g).to(dev)
    ref = None
    bad = 0
    for it in range(reps):
        for p in st.parameters():
            p.grad = None
        h.grad = e.grad = None
        for blk in st.blocks:          # same random-mask stream every repetition
            blk.mha._calls = 0
        h2, e2 = st(h, e, mask)
        torch.autograd.backward([h2, e2], [dh, de])
        cur = [h2.detach().clone(), e2.detach().clone(), h.grad.clone(), e.grad.clone()] + [p.grad.clone() for p in st.parameters()]
        if ref is None:
            ref = cur
        else:
            for i, (a, b) in enumerate(zip(ref, cur)):
                if not torch.equal(a, b):
                    bad += 1
                    print(f"rep {it}: tensor #{i} differs, max |diff| {float((a - b).abs().max()):.3e}")
                    break
    print(f"soak: {reps} repetitions, {bad} with differences")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
