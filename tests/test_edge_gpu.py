"""GPU parity of the edge-channel projection kernels vs the fp64 oracle."""
import pytest
import torch

from oracle import egt_oracle as O
from util import assert_close, FWD, BWD

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("De,rows_shape,use_ln,gates,act", [
    (64, (2, 9, 9), True, True, None),
    (64, (3, 37, 37), True, True, None),      # 4107 rows: multiple tiles + ragged tail
    (48, (2, 11, 11), True, True, None),
    (32, (1, 8, 8), True, True, None),
    (8, (1, 20, 20), True, True, None),
    (16, (2, 7, 7), False, True, None),       # 'bias' variant: no LN
    (16, (2, 7, 7), False, True, "lrelu2"),
    (16, (1, 6, 6), True, False, "elu"),      # ungated
    (64, (1, 5, 5), True, True, "relu"),
])
def test_edge_proj(De, rows_shape, use_ln, gates, act, gpu, egt_lib):
    from egt_amd import edge_proj
    g = torch.Generator().manual_seed(De * 7 + rows_shape[1])
    r = lambda *s: torch.randn(*s, generator=g)
    e = r(*rows_shape, De) * 1.7 + 0.4
    gamma, beta = 1 + 0.2 * r(De), 0.2 * r(De)
    Wg, bg, We, be = 0.3 * r(De, 8), 0.1 * r(8), 0.3 * r(De, 8), 0.1 * r(8)
    dG, dE = r(*rows_shape, 8), r(*rows_shape, 8)

    def oracle():
        t = [x.double().requires_grad_() for x in (e, gamma, beta, Wg, bg, We, be)]
        e_, ga, bt, wg, bg_, we, be_ = t
        en = O.layer_norm(e_, ga, bt) if use_ln else e_
        G = O.dense(en, wg, bg_) if gates else None
        E = O.edge_activation_fn(O.dense(en, we, be_), act)
        loss = (E * dE.double()).sum() + ((G * dG.double()).sum() if gates else 0)
        grads = torch.autograd.grad(loss, t, allow_unused=True)
        return G, E, grads

    Gr, Er, gr = oracle()
    t = [x.to(gpu).requires_grad_() for x in (e, gamma, beta, Wg, bg, We, be)]
    G, E = edge_proj(t[0], t[1] if use_ln else None, t[2] if use_ln else None,
                     t[3] if gates else None, t[4] if gates else None, t[5], t[6],
                     use_ln=use_ln, edge_activation=act)
    assert_close(E, Er, name="E", **FWD)
    if gates:
        assert_close(G, Gr, name="G", **FWD)
    loss = (E * dE.to(gpu)).sum() + ((G * dG.to(gpu)).sum() if gates else 0)
    grads = torch.autograd.grad(loss, t, allow_unused=True)
    names = ["de", "dgamma", "dbeta", "dWg", "dbg", "dWe", "dbe"]
    for n, a, b in zip(names, grads, gr):
        if b is None:
            assert a is None
            continue
        assert_close(a, b, name=n, **BWD)


@pytest.mark.parametrize("De,rows_shape", [(32, (2, 13, 13)), (64, (1, 21, 21)), (8, (1, 10, 10))])
def test_edge_proj_passthrough_accumulates(De, rows_shape, gpu, egt_lib):
    """edge_proj(passthrough=True): the gradient arriving on the handed-on e is summed into d_e by
    the kernel (egt_edge_proj_bwd_acc) -- same result as autograd's separate add."""
    from egt_amd import edge_proj
    g = torch.Generator().manual_seed(De + 5)
    r = lambda *s: torch.randn(*s, generator=g)
    e = r(*rows_shape, De)
    gamma, beta = 1 + 0.2 * r(De), 0.2 * r(De)
    Wg, bg, We, be = 0.3 * r(De, 8), 0.1 * r(8), 0.3 * r(De, 8), 0.1 * r(8)
    dG, dE, dres = r(*rows_shape, 8), r(*rows_shape, 8), r(*rows_shape, De)
    t64 = [x.double().requires_grad_() for x in (e, gamma, beta, Wg, bg, We, be)]
    en = O.layer_norm(t64[0], t64[1], t64[2])
    loss = (O.dense(en, t64[3], t64[4]) * dG.double()).sum() + (O.dense(en, t64[5], t64[6]) * dE.double()).sum() \
        + (t64[0] * dres.double()).sum()
    gref = torch.autograd.grad(loss, t64)
    t = [x.to(gpu).requires_grad_() for x in (e, gamma, beta, Wg, bg, We, be)]
    G, E, thru = edge_proj(*t, use_ln=True, passthrough=True)
    assert thru.data_ptr() == t[0].data_ptr()
    loss = (G * dG.to(gpu)).sum() + (E * dE.to(gpu)).sum() + (thru * dres.to(gpu)).sum()
    grads = torch.autograd.grad(loss, t)
    for n, a, b in zip(["de", "dgamma", "dbeta", "dWg", "dbg", "dWe", "dbe"], grads, gref):
        assert_close(a, b, name=n, **BWD)


@pytest.mark.parametrize("De,rows_shape", [(64, (2, 9, 9)), (64, (3, 37, 37)), (48, (2, 11, 11)),
                                           (32, (1, 8, 8)), (16, (1, 7, 7)), (8, (1, 20, 20))])
def test_edge_update(De, rows_shape, gpu, egt_lib):
    from egt_amd import edge_update
    g = torch.Generator().manual_seed(De + rows_shape[1])
    r = lambda *s: torch.randn(*s, generator=g)
    e, hh = r(*rows_shape, De), r(*rows_shape, 8) * 2
    Wr, br, de = 0.4 * r(8, De), 0.1 * r(De), r(*rows_shape, De)
    t64 = [x.double().requires_grad_() for x in (e, hh, Wr, br)]
    ref = O.dense(t64[1], t64[2], t64[3]) + t64[0]
    gref = torch.autograd.grad((ref * de.double()).sum(), t64)
    t = [x.to(gpu).requires_grad_() for x in (e, hh, Wr, br)]
    out = edge_update(*t)
    assert_close(out, ref, name="e_out", **FWD)
    grads = torch.autograd.grad((out * de.to(gpu)).sum(), t)
    for n, a, b in zip(["de", "dh_hat", "dWr", "dbr"], grads, gref):
        assert_close(a, b, name=n, **BWD)


def test_unsupported_width_raises(gpu, egt_lib):
    from egt_amd import edge_update
    with pytest.raises(AssertionError):
        edge_update(torch.zeros(1, 2, 2, 24, device=gpu), torch.zeros(1, 2, 2, 8, device=gpu),
                    torch.zeros(8, 24, device=gpu), torch.zeros(24, device=gpu))


def test_edge_proj_bwd_acc_in_place(gpu, egt_lib):
    """egt_edge_proj_bwd_acc with d_e_base aliasing d_e (documented in include/egt_amd.h): the
    residual-branch gradient buffer is updated in place and equals base + egt_edge_proj_bwd's d_e."""
    import ctypes as C
    from egt_amd import _lib as L
    from egt_amd.functional import _edge_desc
    g = torch.Generator().manual_seed(3)
    r = lambda *s: torch.randn(*s, generator=g).to(gpu)
    De, rows = 32, 1000   # ragged last tile
    e, gamma, beta = r(rows, De), 1 + 0.1 * r(De), 0.1 * r(De)
    Wg, We, dG, dE, base = 0.3 * r(De, 8), 0.3 * r(De, 8), r(rows, 8), r(rows, 8), r(rows, De)
    desc = _edge_desc(e, True, True, L.ACT_NONE, 0.0, 1e-3)
    mk = lambda *s: torch.empty(*s, device=gpu)
    ws = torch.empty(egt_lib.egt_edge_proj_bwd_workspace_bytes(C.byref(desc)), device=gpu, dtype=torch.uint8)
    grads = [mk(De), mk(De), mk(De, 8), mk(8), mk(De, 8), mk(8)]
    d_ref = mk(rows, De)
    L.check(egt_lib.egt_edge_proj_bwd(C.byref(desc), L.ptr(e), L.ptr(gamma), L.ptr(beta), L.ptr(Wg), L.ptr(We), None,
                                      L.ptr(dG), L.ptr(dE), L.ptr(d_ref), *[L.ptr(t) for t in grads], L.ptr(ws),
                                      L.current_stream()))
    grads_ref = [t.clone() for t in grads]
    inout = base.clone()
    L.check(egt_lib.egt_edge_proj_bwd_acc(C.byref(desc), L.ptr(e), L.ptr(gamma), L.ptr(beta), L.ptr(Wg), L.ptr(We), None,
                                          L.ptr(dG), L.ptr(dE), L.ptr(inout), L.ptr(inout), *[L.ptr(t) for t in grads],
                                          L.ptr(ws), L.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(inout, base + d_ref)          # same arithmetic, one extra add per element
    for a, b in zip(grads, grads_ref):
        assert torch.equal(a, b)                     # parameter gradients: bit-identical (deterministic partials)
