"""hipGraph capture of the step + device-resident mask seeds (egt_amd/graph.py, EGT_BF_SEED_DEVICE).

The random attention mask (egt_layers.py:97-103) is redrawn on every call; a captured launch freezes its
kernel arguments, so the seed is completed from HBM inside the kernels.  Bar: the device-seed path and
every hipGraph replay are BIT-identical to the eager host-seed call with the same call index — outputs,
input gradients and every parameter gradient — across the kernel selections (De = 64 MFMA-tile kernels,
De = 8 VALU kernels fp32 / bf16, ragged N)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


def _stack(gpu, N, De, Ly, seed, p=0.2, Dh=64):
    from egt_amd import EGTStack
    torch.manual_seed(3)
    st = EGTStack(model_height=Ly, model_width=Dh, edge_width=De, num_heads=8, random_mask_prob=p, seed=seed,
                  fused=True).to(gpu).train()
    with torch.no_grad():
        for prm in st.parameters():
            if prm.dim() == 1:
                prm.add_(0.2 * torch.randn_like(prm))
    return st


def _inputs(gpu, B, N, De, Dh=64, bf16=False):
    g = torch.Generator().manual_seed(N * 3 + De)
    h = torch.randn(B, N, Dh, generator=g).to(gpu)
    e = (torch.randn(B, N, N, De, generator=g) * 1.3).to(gpu)
    dh = torch.randn(B, N, Dh, generator=g).to(gpu)
    de = torch.randn(B, N, N, De, generator=g).to(gpu)
    mask = torch.ones(B, N, dtype=torch.bool); mask[1, N - 3:] = False
    if bf16:
        e, de = e.bfloat16(), de.bfloat16()
    return h.requires_grad_(), e.requires_grad_(), mask.to(gpu), dh, de


def _run(st, h, e, mask, dh, de):
    h.grad = None; e.grad = None
    for p in st.parameters():
        p.grad = None
    h2, e2 = st(h, e, mask)
    torch.autograd.backward([h2, e2], [dh, de])
    return h2, e2


def _snapshot(st, h, e, out):
    # detached copies: a snapshot that kept the step's autograd graph alive would also keep the leaves' AccumulateGrad
    # nodes (bound to the stream they were created on) alive into the capture
    return [t.detach().clone() for t in (out[0], out[1], h.grad, e.grad)] + [p.grad.detach().clone() for p in st.parameters()]


def _same(a, b, what):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.dtype == y.dtype and torch.equal(x, y), f"{what}: tensor #{i} differs (max |d| = {(x.float() - y.float()).abs().max().item():.3e})"


@pytest.mark.parametrize("N,De,bf16", [(32, 64, False), (37, 48, False), (40, 8, False), (40, 8, True), (150, 8, True)])
def test_device_seed_and_graph_replay_bit_exact(N, De, bf16, gpu, egt_lib):
    from egt_amd import DeviceSeeds, GraphedStep
    B, Ly, calls = 2, 3, 6
    ref = _stack(gpu, N, De, Ly, seed=7)
    h, e, mask, dh, de = _inputs(gpu, B, N, De, bf16=bf16)
    want = []
    for _ in range(calls):                                   # eager, host-side seeds: calls 1..6
        want.append(_snapshot(ref, h, e, _run(ref, h, e, mask, dh, de)))
    assert ref.last_path == "fused-stack"
    for k in range(1, calls):                                # the sample really changes from call to call
        assert not torch.equal(want[k][0], want[0][0])

    st = _stack(gpu, N, De, Ly, seed=7)                      # same weights, same seed, device-resident seeds
    seeds = DeviceSeeds.attach(st, gpu)
    assert len(seeds.modules) == Ly
    seeds.advance()                                          # eager call 1 on the device-seed path
    _same(_snapshot(st, h, e, _run(st, h, e, mask, dh, de)), want[0], "device seed, eager call 1")
    g = GraphedStep(lambda: _run(st, h, e, mask, dh, de), seeds, warmup=2)   # warm-up = calls 2, 3 (eager, side stream)
    for k in range(3, calls):                                # replays = calls 4, 5, 6
        o = g.replay()
        torch.cuda.synchronize()
        _same(_snapshot(st, h, e, o), want[k], f"hipGraph replay, call {k + 1}")
    del o, g
    seeds.detach()                                           # the host-side stream continues where the device words stand
    assert st.blocks[0].mha._calls == calls and st.blocks[0].mha.seed_device is None
    more = _snapshot(st, h, e, _run(st, h, e, mask, dh, de))
    _same(more, _snapshot(ref, h, e, _run(ref, h, e, mask, dh, de)), "after detach, call 7")


def test_graphed_whole_layers_match_eager(gpu, egt_lib):
    """Per-block calls (attention block + node/edge FFN per layer: every block owns its EGT module and its word)."""
    from egt_amd import EGTLayerStack, DeviceSeeds, GraphedStep
    B, N, De, Ly = 2, 24, 32, 2

    def make():
        torch.manual_seed(5)
        return EGTLayerStack(model_height=Ly, model_width=64, edge_width=De, num_heads=8, random_mask_prob=0.25, seed=3,
                             fused=True).to(gpu).train()
    ref, st = make(), make()
    h, e, mask, dh, de = _inputs(gpu, B, N, De)
    want = [_snapshot(ref, h, e, _run(ref, h, e, mask, dh, de)) for _ in range(4)]
    seeds = DeviceSeeds.attach(st, gpu)
    assert len(seeds.modules) == Ly
    g = GraphedStep(lambda: _run(st, h, e, mask, dh, de), seeds, warmup=2)
    for k in (2, 3):
        o = g.replay()
        torch.cuda.synchronize()
        _same(_snapshot(st, h, e, o), want[k], f"layers, replay = call {k + 1}")


def test_seed_device_flag_needs_pointer(gpu, egt_lib):
    import ctypes as C
    from egt_amd import _lib as L
    d = L.BlockDesc(B=1, N=16, H=8, d=8, De=64, dtype=L.EGT_F32, flags=L.BF_TRAINING | L.BF_SEED_DEVICE, clip_lo=0, clip_hi=0,
                    random_mask_prob=0.1, ln_eps=1e-3, reserved=0, seed=0, seed_device=None)
    assert egt_lib.egt_block_supported(C.byref(d)) == 0


@pytest.mark.parametrize("graph", [False, True])
def test_overlapped_node_ffn_is_bit_identical(graph, gpu, egt_lib):
    """EGTLayerStack.overlap_ffn: the node FFN on a side stream beside the edge FFN — same kernels, same results, eager
    and as a fork / join inside a captured hipGraph."""
    from egt_amd import EGTLayerStack, DeviceSeeds, GraphedStep
    B, N, De, Ly = 4, 40, 8, 3

    def make():
        torch.manual_seed(5)
        return EGTLayerStack(model_height=Ly, model_width=64, edge_width=De, num_heads=8, random_mask_prob=0.25, seed=3,
                             fused=True).to(gpu).train()
    ref, st = make(), make()
    st.overlap_ffn = True
    h, e, mask, dh, de = _inputs(gpu, B, N, De)
    want = [_snapshot(ref, h, e, _run(ref, h, e, mask, dh, de)) for _ in range(4)]
    if not graph:
        for k in range(4):
            _same(_snapshot(st, h, e, _run(st, h, e, mask, dh, de)), want[k], f"overlap, eager call {k + 1}")
        return
    seeds = DeviceSeeds.attach(st, gpu)
    g = GraphedStep(lambda: _run(st, h, e, mask, dh, de), seeds, warmup=2)
    for k in (2, 3):
        o = g.replay()
        torch.cuda.synchronize()
        _same(_snapshot(st, h, e, o), want[k], f"overlap, replay = call {k + 1}")


def test_direct_gradient_sinks_match_autograd_accumulation(gpu, egt_lib):
    """FlatGradAllReduce(direct=True): the fused block / FFN backward writes each parameter gradient straight into the
    parameter's view of the flat buffer (no `grad += g` launch per parameter).  Same numbers as autograd's own accumulation
    into the zeroed buffer and as plain .grad tensors."""
    from egt_amd import EGTLayerStack
    from egt_amd.dp import FlatGradAllReduce
    from egt_amd import fused
    B, N, De, Ly = 2, 24, 8, 2

    def make():
        torch.manual_seed(5)
        return EGTLayerStack(model_height=Ly, model_width=64, edge_width=De, num_heads=8, random_mask_prob=0.25, seed=3,
                             fused=True).to(gpu).train()
    h, e, mask, dh, de = _inputs(gpu, B, N, De)
    plain = make()
    want = _snapshot(plain, h, e, _run(plain, h, e, mask, dh, de))
    for direct in (False, True):
        st = make()
        fa = FlatGradAllReduce(list(st.parameters()), direct=direct)
        bufs, rets = fused.grad_sinks(list(st.parameters())[:3])
        assert all((r is None) == direct for r in rets)
        for _ in range(2):                                  # the second step starts from a dirty buffer
            fa.zero(); fa.rebind()
            h.grad = None; e.grad = None
            h2, e2 = st(h, e, mask)
            torch.autograd.backward([h2, e2], [dh, de])
            for m in st.modules():                          # (same call index as `plain` for the comparison below)
                if hasattr(m, "_calls"):
                    m._calls = 0
        for m in st.modules():
            if hasattr(m, "_calls"):
                m._calls = 0
        fa.zero(); fa.rebind()
        got = _snapshot(st, h, e, _run_keep(st, h, e, mask, dh, de))
        _same(got, want, f"direct={direct}")
        off = 0
        for p in st.parameters():                           # .grad still aliases the flat buffer
            assert p.grad.data_ptr() == fa.flat[off:].data_ptr()
            off += p.numel()


def _run_keep(st, h, e, mask, dh, de):
    """_run without dropping the pre-bound .grad views"""
    h.grad = None; e.grad = None
    h2, e2 = st(h, e, mask)
    torch.autograd.backward([h2, e2], [dh, de])
    return h2, e2


@pytest.mark.parametrize("N,De", [(32, 64), (40, 8)])
def test_bound_flat_gradients_match_the_adopted_views(N, De, gpu, egt_lib):
    """EGTStack.bind_flat_gradients(): the stack backward writes into one persistent flat buffer whose views are the
    parameters' .grad (no per-parameter autograd work); same numbers as the default path, step after step, eager and replayed."""
    from egt_amd import DeviceSeeds, GraphedStep
    ref, st = _stack(gpu, N, De, 3, seed=7), _stack(gpu, N, De, 3, seed=7)
    h, e, mask, dh, de = _inputs(gpu, 2, N, De)
    want = [_snapshot(ref, h, e, _run(ref, h, e, mask, dh, de)) for _ in range(5)]
    flat = st.bind_flat_gradients()
    for k in range(2):
        flat.fill_(float("nan"))                            # the backward must overwrite every element
        got = _snapshot(st, h, e, _run_keep(st, h, e, mask, dh, de))
        _same(got, want[k], f"bound, eager call {k + 1}")
        assert st.grad_holder.flat is flat and all(p.grad.data_ptr() >= flat.data_ptr() for p in st.parameters())
    seeds = DeviceSeeds.attach(st, gpu)
    g = GraphedStep(lambda: _run_keep(st, h, e, mask, dh, de), seeds, warmup=1)     # warm-up = call 3
    for k in (3, 4):
        flat.fill_(float("nan"))
        o = g.replay()
        torch.cuda.synchronize()
        _same(_snapshot(st, h, e, o), want[k], f"bound, replay = call {k + 1}")
    del o, g
    seeds.detach()
    st.unbind_flat_gradients()
    assert all(p.grad is None for p in st.parameters())
