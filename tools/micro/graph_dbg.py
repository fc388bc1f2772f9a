"""Bisection harness for hipGraph capture of the stack step (one variant per process)."""
import argparse, sys, os, faulthandler
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
faulthandler.enable()
import torch
ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=2); ap.add_argument("--N", type=int, default=32); ap.add_argument("--De", type=int, default=64)
ap.add_argument("--Ly", type=int, default=3); ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--no-eager-first", action="store_true"); ap.add_argument("--ret-none", action="store_true")
ap.add_argument("--no-seeds", action="store_true"); ap.add_argument("--fwd-only", action="store_true")
ap.add_argument("--keep-grads", action="store_true")
a = ap.parse_args()
from test_graph_gpu import _stack, _inputs, _run
from egt_amd import DeviceSeeds, GraphedStep
gpu = torch.device("cuda:0")
st = _stack(gpu, a.N, a.De, a.Ly, seed=7)
h, e, mask, dh, de = _inputs(gpu, a.B, a.N, a.De)
seeds = None if a.no_seeds else DeviceSeeds.attach(st, gpu)
def run():
    if a.fwd_only:
        with torch.no_grad():
            return st(h, e, mask)
    if a.keep_grads:
        h2, e2 = st(h, e, mask)
        torch.autograd.backward([h2, e2], [dh, de])
        return h2, e2
    return _run(st, h, e, mask, dh, de)
if not a.no_eager_first:
    if seeds: seeds.advance()
    run()
def fn():
    o = run()
    return None if a.ret_none else o
g = GraphedStep(fn, seeds, warmup=a.warmup)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
print("OK", vars(a))
