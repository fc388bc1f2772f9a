#!/usr/bin/env python
"""Randomised parity sweep of the fused pair operator (k_pair_fwd / k_pair_bwd, d = 64, De = 32) against the fp64 oracle:
random batch / N (ragged tiles, single tiles, many tiles) / node counts (incl. graphs without nodes) / clip on-off / eval and
training mode (in-kernel random mask == oracle fed the rng_ref replica), the tolerances of tests/test_pair_gpu.py.
    python tools/sweep_pair.py [cases = 24]      (EGT_SWEEP_SEED picks another stream)
Prints one line per case and `failures: K`; exit code 1 if K > 0."""
import os
import random
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
import torch

import cases as CS
from test_block_gpu import PMAP
from test_pair_gpu import _case, _run, _compare, ATTRS
from egt_amd import EGTBlock
from oracle import rng_ref


def main():
    ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rnd = random.Random(int(os.environ.get("EGT_SWEEP_SEED", "606")))
    gpu = torch.device("cuda:0")
    fails = 0
    for i in range(ncase):
        B = rnd.choice([1, 1, 2, 3, 4])
        N = rnd.choice([rnd.randint(1, 15), 16, rnd.randint(17, 47), 48, rnd.randint(49, 130), rnd.randint(17, 96)])
        nodes = [rnd.choice([N, N, rnd.randint(0, N), rnd.randint(max(0, N - 17), N)]) for _ in range(B)]
        clip = rnd.random() < 0.7
        train = rnd.random() < 0.4
        p = rnd.choice([0.1, 0.3]) if train else None
        inp, params, c = _case(B, N, nodes, seed=rnd.randrange(1 << 30), rand_p=p)
        kw = {} if clip else dict(clip_logits_value=None)   # (default: the reference's [-5, 5])
        blk = EGTBlock(model_width=512, edge_width=32, num_heads=8, fused="auto", **kw).to(gpu)
        with torch.no_grad():
            for k, (m, a) in PMAP.items():
                getattr(getattr(blk, m), a).copy_(params[k].to(gpu))
        tag = f"case {i:3d}: B {B} N {N:3d} nodes {nodes} clip {int(clip)} train p {p}"
        try:
            if train:
                blk.mha.random_mask_prob = p
                blk.train()
            else:
                blk.eval()
            out = _run(blk, inp, gpu)
            assert blk.last_path == "fused-pair", blk.last_path
            inp2 = inp
            if train:
                m = blk.mha
                seed = (m.seed * 0x9E3779B97F4A7C15 + m._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
                inp2 = dict(inp, rand_mask=torch.from_numpy(rng_ref.random_mask(seed, B, N, 8, p)))
            ref = CS.block_oracle(inp2, params, dict(num_heads=8, **ATTRS, **kw))
            _compare(blk, out, ref)
            print(tag, "ok", flush=True)
        except AssertionError as ex:
            fails += 1
            print(tag, "FAIL", str(ex)[:300], flush=True)
    print(f"sweep: fused pair operator, {ncase} random geometries, failures: {fails}")
    sys.exit(1 if fails else 0)


if __name__ == "__main__":
    main()
