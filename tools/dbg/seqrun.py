"""One process runs a SEQUENCE of stack geometries (the order of tests/test_block_gpu.py::test_stack_call_vs_oracle); every geometry's
outputs are compared bit for bit with those of the first process that ran (REF_DIR).  python tools/dbg/seqrun.py REF_DIR"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from egt_amd import EGTStack
SEQ = [(24, 64, 64, False), (32, 64, 64, True), (11, 48, 48, False), (20, 8, 64, True), (80, 16, 64, True), (32, 32, 64, False), (48, 48, 64, True),
       (32, 8, 64, True), (128, 8, 64, False), (150, 8, 64, True), (144, 16, 64, False), (37, 48, 48, True), (40, 8, 32, True), (24, 16, 40, False), (20, 64, 8, False)]
ref_dir = sys.argv[1]; os.makedirs(ref_dir, exist_ok=True)
gpu = torch.device("cuda:0"); B, Ly, p = 2, 3, 0.2
for (N, De, Dh, train) in SEQ:
    torch.manual_seed(11)
    st = EGTStack(model_height=Ly, model_width=Dh, edge_width=De, num_heads=8, random_mask_prob=p if train else 0.0, seed=5, fused=True).to(gpu).train(train)
    with torch.no_grad():
        for prm in st.parameters():
            if prm.dim() == 1: prm.add_(0.2 * torch.randn_like(prm))
    g = torch.Generator().manual_seed(N * 7 + De)
    h = torch.randn(B, N, Dh, generator=g); e = torch.randn(B, N, N, De, generator=g) * 1.3
    mask = torch.ones(B, N, dtype=torch.bool); mask[1, N - 3:] = False
    dh = torch.randn(B, N, Dh, generator=g); de = torch.randn(B, N, N, De, generator=g)
    hg = h.to(gpu).requires_grad_(); eg = e.to(gpu).requires_grad_()
    h2, e2 = st(hg, eg, mask.to(gpu))
    torch.autograd.backward([h2, e2], [dh.to(gpu), de.to(gpu)])
    torch.cuda.synchronize()
    out = {"h_out": h2.detach().cpu(), "e_out": e2.detach().cpu(), "dh": hg.grad.cpu(), "de": eg.grad.cpu()}
    for n_, prm in st.named_parameters(): out["g:" + n_] = prm.grad.cpu()
    f = os.path.join(ref_dir, f"{N}_{De}_{Dh}.pt")
    if not os.path.exists(f): torch.save(out, f); continue
    old = torch.load(f)
    for k, v in out.items():
        if not torch.equal(v, old[k]):
            d = (v != old[k]); idx = d.nonzero()
            info = f"{int(d.sum())}/{d.numel()} differ, max |diff| {float((v - old[k]).abs().max()):.3e}"
            if v.dim() == 3: info += f"; batch {sorted(set(idx[:,0].tolist()))} rows {sorted(set(idx[:,1].tolist()))} cols {sorted(set(idx[:,2].tolist()))}"
            elif v.dim() <= 2: info += f"; idx0 {sorted(set(idx[:,0].tolist()))[:48]}"
            print(f"[{N},{De},{Dh}] {k}: {info}", flush=True)
