#!/usr/bin/env python3
"""Randomised parity sweep of the MFMA inner-op kernels (egt_attn_mfma.hip: d in {16,32,64}) against the fp64 oracle: ragged N,
odd tile counts, every feature mix the kernels cover (edge / gate / attention-mask tensor / key padding / injected random-mask
bytes / clip), the straight-line and the generic instances.  Test infrastructure (imports oracle/); run on the GPU box:
    python tools/sweep_mfma.py [seed] [count]"""
import os
import random
import sys
import traceback

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))
sys.path.insert(0, os.path.join(REPO, "tools"))
from sweep_parity import one_attn  # noqa: E402


def main(seed=None, count=None):
    if seed is None:
        seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    if count is None:
        count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rnd = random.Random(seed)
    dev = torch.device("cuda", 0)
    bad = 0
    for i in range(count):
        N = rnd.choice([1, 2, 5, 15, 16, 17, 31, 32, 33, 47, 48, 49, 63, 64, 65, 80, 97, 128, 130, 160])
        d = rnd.choice([16, 32, 64])
        main_cfg = rnd.random() < 0.4      # the straight-line instances: edge + gate + key padding + clip, no mask tensors
        if main_cfg:
            opts = dict(edge=True, gate=True, attn_mask=False, pad=True, rand=False, drop=False, clip=True, deg=False, scaler="log", atild=False)
        else:
            opts = dict(edge=rnd.random() < 0.8, gate=rnd.random() < 0.7, attn_mask=rnd.random() < 0.4, pad=rnd.random() < 0.7,
                        rand=rnd.random() < 0.5, drop=False, clip=rnd.random() < 0.7, deg=False, scaler="log", atild=False)
        B = rnd.choice([1, 2, 3])
        try:
            one_attn(N, d, opts, B, dev, seed=9000 + 131 * seed + i)
        except Exception as ex:  # noqa: BLE001
            bad += 1
            print("FAIL", dict(N=N, d=d, B=B, **opts), type(ex).__name__, str(ex)[:300])
            if not isinstance(ex, AssertionError):
                traceback.print_exc()
    print(f"sweep_mfma seed {seed}: {count - bad}/{count} configurations ok")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
