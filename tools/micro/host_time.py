"""Host time per eager stack step (how far the Python side runs ahead of the GPU): python tools/micro/host_time.py [B]"""
import os, sys, time, cProfile, pstats, io
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from egt_amd import EGTStack
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
torch.manual_seed(0)
st = EGTStack(model_height=10, model_width=64, edge_width=64, num_heads=8, random_mask_prob=0.1, seed=1, fused=True).to(dev).train()
h = torch.randn(B, 64, 64, device=dev, requires_grad=True); e = torch.randn(B, 64, 64, 64, device=dev, requires_grad=True)
mask = torch.ones(B, 64, dtype=torch.bool, device=dev); dh = torch.randn_like(h); de = torch.randn_like(e)
params = list(st.parameters())
def step():
    for p in params: p.grad = None
    h.grad = None; e.grad = None
    h2, e2 = st(h, e, mask)
    torch.autograd.backward([h2, e2], [dh, de])
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host time per step (B={B}: tiny kernels, the host is the bound): {(t1 - t0) / 200 * 1e3:.3f} ms")
tf = tb = 0.0
for _ in range(200):
    for p in params: p.grad = None
    h.grad = None; e.grad = None
    a = time.perf_counter(); h2, e2 = st(h, e, mask); b = time.perf_counter()
    torch.autograd.backward([h2, e2], [dh, de]); c = time.perf_counter()
    tf += b - a; tb += c - b
torch.cuda.synchronize()
print(f"  of which forward call {tf / 200 * 1e3:.3f} ms, autograd.backward {tb / 200 * 1e3:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
