import numpy as np
import torch

# fp32 tolerances vs the fp64 oracle: forward rtol 1e-4, grads rtol 1e-3;
# the absolute term is relative to the tensor's max magnitude (sums over up to
# B*N*N rows for weight grads).
# `floor`: lower bound on the magnitude the absolute term is taken relative to — gradients
# that are identically 0 in exact arithmetic (e.g. d(dense_edge_b.bias) in the 'bias'
# variant: a sum of softmax grads) come out as fp32 rounding noise of O(1e-6).
FWD = dict(rtol=1e-4, arel=2e-5, floor=1e-30)
BWD = dict(rtol=1e-3, arel=1e-4, floor=0.1)


def assert_close(actual, ref, *, rtol, arel, name="", floor=1e-30):
    a = actual.detach().double().cpu() if isinstance(actual, torch.Tensor) else torch.as_tensor(np.asarray(actual)).double()
    r = ref.detach().double().cpu() if isinstance(ref, torch.Tensor) else torch.as_tensor(np.asarray(ref)).double()
    assert a.shape == r.shape, f"{name}: shape {tuple(a.shape)} vs {tuple(r.shape)}"
    assert torch.isfinite(a).all(), f"{name}: non-finite values"
    scale = float(r.abs().max()) if r.numel() else 0.0
    tol = arel * max(scale, floor) + rtol * r.abs()
    err = (a - r).abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        raise AssertionError(
            f"{name}: {int(bad.sum())}/{a.numel()} elements out of tolerance; worst err "
            f"{float(err.flatten()[i]):.3e} (ref {float(r.flatten()[i]):.6e}, got {float(a.flatten()[i]):.6e}, "
            f"max|ref| {scale:.3e})")


def load_golden(path):
    z = np.load(path)
    tree = {}
    for k in z.files:
        if "/" in k:
            a, b = k.split("/", 1)
            tree.setdefault(a, {})[b] = z[k]
        else:
            tree[k] = z[k]
    return tree
