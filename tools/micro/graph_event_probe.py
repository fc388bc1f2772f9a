"""Which HIP event calls are legal under torch's stream capture on this ROCm (for the profile's event-record nodes)?
hipEventRecordWithFlags(..., hipEventRecordExternal) is refused (invalid argument); explicit event-record nodes spliced into the
capturing stream's dependency chain (hipStreamGetCaptureInfo_v2 + hipGraphAddEventRecordNode + hipStreamUpdateCaptureDependencies)?"""
import ctypes as C
import torch

hip = C.CDLL("libamdhip64.so")
hip.hipGetErrorString.restype = C.c_char_p
def rc(name, r):
    print(f"  {name}: {r} {hip.hipGetErrorString(r).decode()}")
    return r

def record_node(st, event):
    status, cid, graph, deps, ndeps = C.c_int(0), C.c_ulonglong(0), C.c_void_p(), C.POINTER(C.c_void_p)(), C.c_size_t(0)
    rc("getCaptureInfo_v2", hip.hipStreamGetCaptureInfo_v2(st, C.byref(status), C.byref(cid), C.byref(graph), C.byref(deps), C.byref(ndeps)))
    print("   status", status.value, "ndeps", ndeps.value)
    node = C.c_void_p()
    rc("addEventRecordNode", hip.hipGraphAddEventRecordNode(C.byref(node), graph, deps, ndeps, event))
    rc("updateCaptureDeps", hip.hipStreamUpdateCaptureDependencies(st, C.byref(node), C.c_size_t(1), 1))   # 1 = hipStreamSetCaptureDependencies

x = torch.zeros(1 << 22, device="cuda")
a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
for e in (a, b, c):
    rc("create", hip.hipEventCreate(C.byref(e)))
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    x.add_(1.0)
    record_node(st, a)
    x.mul_(2.0); x.add_(3.0)
    record_node(st, b)
    for _ in range(20):
        x.add_(1.0)
    record_node(st, c)
    rc("getLastError", hip.hipGetLastError())
torch.cuda.synchronize()
for k in range(3):
    g.replay(); torch.cuda.synchronize()
    ms = C.c_float(0)
    r = hip.hipEventElapsedTime(C.byref(ms), a, b); print(f"  replay {k}: a->b rc {r} {ms.value * 1e3:.1f} us")
    r = hip.hipEventElapsedTime(C.byref(ms), b, c); print(f"  replay {k}: b->c rc {r} {ms.value * 1e3:.1f} us")
print("x[0]", float(x[0]))
