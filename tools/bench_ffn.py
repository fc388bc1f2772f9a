#!/usr/bin/env python3
"""FFN scope of SURVEY.md §8(f)-1: y = x + Dense2(elu(Dense1(LN(x)))) fwd+bwd on the edge channels
of a BASELINE config (default config 2: [128,64,64,64] fp32; `bench_ffn.py edge B N W` for another one).  Prints one JSON line."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from egt_amd import FFN, _lib as L  # noqa: E402


def main():
    B, N, W = 128, 64, 64
    if len(sys.argv) > 4:   # bench_ffn.py edge|node B N W
        B, N, W = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    if len(sys.argv) > 1 and sys.argv[1] == "node":
        shape = (B, N, W)
    else:
        shape = (B, N, N, W)
    mm = os.environ.get("EGT_FFN_MATMUL", "f32")
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    m = FFN(W, matmul=mm).to(dev)
    x = torch.randn(*shape, device=dev, requires_grad=True)
    dy = torch.randn(*shape, device=dev)
    lib = L.load()

    def step():
        for p in m.parameters():
            p.grad = None
        x.grad = None
        y = m(x)
        y.backward(dy)

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20
    e0.record()
    for _ in range(K):
        step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    lib.egt_prof_enable(2)
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    lib.egt_prof_enable(0)
    buf = C.create_string_buffer(4096)
    lib.egt_prof_names(buf, 4096)
    ks = {}
    for name in buf.value.decode().split():
        cnt, t = C.c_int64(0), C.c_double(0.0)
        lib.egt_prof_read(name.encode(), C.byref(cnt), C.byref(t))
        if cnt.value:
            ks[name] = round(t.value / cnt.value * 1e3, 1)
    rows = x.numel() // W
    flop_alg = 24 * W * W * rows            # fwd+bwd without recompute
    mfma_issued = (256 + 640) * 1024 * 2 * (rows / 16)   # MFMA flops actually issued (incl. recompute)
    bytes_alg = rows * W * 4 * 5            # fwd: r x, w y; bwd: r x, r dy, w dx
    print(json.dumps({"scope": "ffn", "matmul": mm, "shape": list(shape), "ms_per_step": ms, "graphs_per_s": B / ms * 1e3,
                      "TFLOPs_algorithmic": flop_alg / ms / 1e9, "TFLOPs_issued": mfma_issued / ms / 1e9,
                      "frac_of_157.3": flop_alg / ms / 1e9 / 157.3, "algorithmic_GBps": bytes_alg / ms / 1e6,
                      "kernels_us": ks}))


if __name__ == "__main__":
    main()
