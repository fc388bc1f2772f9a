import torch, time
dev = torch.device("cuda")
n = 128*64*64*64
a = torch.randn(n, device=dev); b = torch.randn(n, device=dev); c = torch.empty_like(a)
def t(fn, k=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3
us = t(lambda: torch.add(a, b, out=c)); print(f"add 2R+1W {3*n*4/1e6:.0f} MB: {us:.1f} us -> {3*n*4/us/1e6:.2f} TB/s")
us = t(lambda: c.copy_(a)); print(f"copy 1R+1W {2*n*4/1e6:.0f} MB: {us:.1f} us -> {2*n*4/us/1e6:.2f} TB/s")
us = t(lambda: a.sum()); print(f"sum 1R {n*4/1e6:.0f} MB: {us:.1f} us -> {n*4/us/1e6:.2f} TB/s")
us = t(lambda: c.fill_(1.0)); print(f"fill 1W {n*4/1e6:.0f} MB: {us:.1f} us -> {n*4/us/1e6:.2f} TB/s")
# in-place (like de' -> de): read 2, write over one of them
us = t(lambda: a.add_(b)); print(f"add_ inplace 2R+1W(in place): {us:.1f} us -> {3*n*4/us/1e6:.2f} TB/s")
