import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B
from egt_amd import EGTStack, _lib
from egt_amd.dp import FlatGradAllReduce
lib = _lib.load()
dev = torch.device("cuda:0")
w = B.WORKLOADS["zinc500k_n64"]
torch.manual_seed(1234)
model = EGTStack(model_height=w["Ly"], model_width=w["Dh"], edge_width=w["De"], num_heads=w["H"],
                 random_mask_prob=w["rand_p"], seed=1, fused="auto").to(dev).train()
h, e, mask, dh, de = B.make_inputs(w, dev)
h.requires_grad_(); e.requires_grad_()
fa = FlatGradAllReduce(model.parameters())
def step():
    fa.zero(); h.grad = None; e.grad = None
    h2, e2 = model(h, e, mask)
    torch.autograd.backward([h2, e2], [dh, de])
for prof in (0, 1):
    lib.egt_prof_enable(2 if prof else 0)
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"prof={prof}: enqueue {1e3*(t1-t0)/20:.3f} ms/step, total {1e3*(t2-t0)/20:.3f} ms/step")
lib.egt_prof_enable(0)
# CUDA graph capture of one step
try:
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    print(f"graph replay: {1e3*(time.perf_counter()-t0)/20:.3f} ms/step")
except Exception as ex:
    print("graph capture failed:", repr(ex)[:500])
