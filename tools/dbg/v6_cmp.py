import os, sys, subprocess, json
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
if len(sys.argv) > 1:
    import torch
    from test_block_gpu import run_block
    out, dparams, _ = run_block(sys.argv[1], torch.device("cuda:0"), fused=True)
    torch.save({k: v.cpu() for k, v in {**out, **{("dp/" + k): v for k, v in dparams.items() if v is not None}}.items()}, sys.argv[2])
    sys.exit(0)
import torch
for case in ("residual_n64", "residual_randmask"):
    for v in ("0", "1"):
        subprocess.run([sys.executable, __file__, case, f"/tmp/v6cmp_{v}.pt"], env=dict(os.environ, EGT_BWD_V6=v), check=True)
    a = torch.load("/tmp/v6cmp_0.pt"); b = torch.load("/tmp/v6cmp_1.pt")
    for k in a:
        d = (a[k] - b[k]).abs().max().item(); s = a[k].abs().max().item()
        print(case, k, "max|v5|", f"{s:.3e}", "max diff", f"{d:.3e}", "ratio-ish", (b[k].flatten()[:4] / a[k].flatten()[:4]).tolist() if d > 1e-3 * s else "")
