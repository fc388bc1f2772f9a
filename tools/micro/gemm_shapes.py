"""The node-side Dense shapes of BASELINE config 5 (B*N = 4096 rows, Dh = 512) through torch's BLAS back ends:
    python tools/micro/gemm_shapes.py cublas|cublaslt      (= rocBLAS | hipBLASLt on ROCm)
MI355X, round 4: rocBLAS 87-101 TF on the five large products (0.55-0.64 of the fp32 matrix peak), 52 TF on dWo; 289 us per block
step in total (hipBLASLt 264 us): the library GEMMs are not where the block scope loses its time."""
import torch, time, sys
lib = sys.argv[1]
torch.backends.cuda.preferred_blas_library(lib)
dev = torch.device("cuda:0")
M, K, N1, N2 = 4096, 512, 1536, 512
x = torch.randn(M, K, device=dev); w1 = torch.randn(K, N1, device=dev); dy = torch.randn(M, N1, device=dev)
w2 = torch.randn(K, N2, device=dev); dy2 = torch.randn(M, N2, device=dev)
def t(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
res = {}
res["fwd x@W1 (4096x512x1536)"] = (t(lambda: x @ w1), 2*M*K*N1)
res["dX dy@W1^T (4096x1536x512)"] = (t(lambda: dy @ w1.t()), 2*M*K*N1)
res["dW x^T@dy (512x4096x1536)"] = (t(lambda: x.t() @ dy), 2*M*K*N1)
res["fwd x@W2 (4096x512x512)"] = (t(lambda: x @ w2), 2*M*K*N2)
res["dX2 (4096x512x512)"] = (t(lambda: dy2 @ w2.t()), 2*M*K*N2)
res["dW2 (512x4096x512)"] = (t(lambda: x.t() @ dy2), 2*M*K*N2)
tot = 0
for k, (us, fl) in res.items():
    print(f"{lib:10s} {k:32s} {us:7.1f} us  {fl/us/1e6:6.1f} TF"); tot += us
print(lib, "total", round(tot, 1), "us")
