import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cases as CS
from test_block_gpu import build_block
from util import assert_close, FWD, BWD
name = sys.argv[1] if len(sys.argv) > 1 else "residual_zinc500k"
dev = torch.device("cuda:0")
inp, params, attrs, c = CS.make_block_case(name)
blk = build_block(c, attrs, params, dev, True)
blk.train(c.get("rand_p") is not None)
cu = lambda t: None if t is None else t.to(dev)
h = cu(inp["h"]).requires_grad_(); e = cu(inp["e"]).requires_grad_()
print("fwd...", flush=True)
h2, e2 = blk(h, e, cu(inp["mask"]), cu(inp["attn_mask"]), rand_mask=cu(inp["rand_mask"]))
torch.cuda.synchronize(); print("fwd ok", flush=True)
ref = CS.block_oracle(inp, params, attrs)
for n, a, b in (("h_out", h2, ref["h_out"]), ("e_out", e2, ref["e_out"])):
    try:
        assert_close(a, b, name=n, **FWD); print(n, "OK")
    except AssertionError as ex:
        print("MISMATCH", ex)
print("bwd...", flush=True)
loss = (h2 * cu(inp["dh"])).sum() + (e2 * cu(inp["de"])).sum()
loss.backward()
torch.cuda.synchronize(); print("bwd ok", flush=True)
from test_block_gpu import PMAP
for n, a, b in (("dh", h.grad, ref["dh"]), ("de", e.grad, ref["de"])):
    try:
        assert_close(a, b, name=n, **BWD); print(n, "OK")
    except AssertionError as ex:
        print("MISMATCH", ex)
for k, (m, at) in PMAP.items():
    if hasattr(blk, m) and ref["dparams"][k] is not None:
        try:
            assert_close(getattr(getattr(blk, m), at).grad, ref["dparams"][k], name=k, **BWD); print(k, "OK")
        except AssertionError as ex:
            print("MISMATCH", ex)
