"""Whole ZINC model: oracle vs committed fixtures and its own invariants on CPU; the
product model (HIP edge embedding + attention blocks + FFNs behind the C-ABI, torch node-side head)
against the fp64 oracle on the GPU: prediction, loss and every parameter gradient."""
import os

import numpy as np
import pytest
import torch

import cases as CS
from util import assert_close, load_golden, FWD, BWD


@pytest.mark.parametrize("name", list(CS.MODEL_CASES))
def test_model_oracle_matches_golden(name):
    g = load_golden(os.path.join(CS.GOLDEN_DIR, f"model_{name}.npz"))
    inp, params, c = CS.make_model_case(name)
    for k, v in g["in"].items():
        assert np.array_equal(inp[k].numpy(), v), k
    out, dparams = CS.model_oracle(inp, params, c)
    assert_close(out["y"].float(), g["out"]["y"], rtol=1e-6, arel=1e-6, name="y")
    assert_close(out["loss"].float(), g["out"]["loss"], rtol=1e-6, arel=1e-6, name="loss")
    for k, v in g["dparams"].items():
        assert_close(dparams[k].float(), v, rtol=1e-5, arel=1e-6, name=k)


def test_model_oracle_invariants():
    """padding invariance: scrambling padded node features' edges / adding padded nodes leaves the
    prediction unchanged; the dead last-layer edge parameters get no gradient (they are not part of the
    reference's Keras model); hop stacking is idempotent once every reachable pair is 1."""
    from oracle import egt_model_oracle as MO
    inp, params, c = CS.make_model_case("zinc_small")
    out, dparams = CS.model_oracle(inp, params, c)
    Ly = c["cfg"]["model_height"]
    for k, g in dparams.items():
        dead = k.startswith(f"layer{Ly - 1}.dense_edge_r") or k.startswith(f"layer{Ly - 1}.ffn_edge") or k.startswith("edge_norm_final")
        if dead:
            assert g is None or float(g.abs().max()) == 0.0, k
        else:
            assert g is not None and float(g.abs().max()) > 0.0, k
    # grow the padding: same graphs padded to N + 3
    B, N = inp["node_features"].shape
    pad = lambda t, v: torch.nn.functional.pad(t, (0, 3) if t.dim() == 2 else (0, 3, 0, 3), value=v)
    inp2 = dict(node_features=pad(inp["node_features"], -1), feature_matrix=pad(inp["feature_matrix"], -1),
                graph_matrix=pad(inp["graph_matrix"], 0.0), target=inp["target"])
    out2, _ = CS.model_oracle(inp2, params, c)
    assert_close(out2["y"], out["y"], rtol=1e-9, arel=1e-9, name="padding invariance")
    hops = MO.stack_hops(inp["graph_matrix"].double(), 12)
    assert set(hops.unique().tolist()) <= {0.0, 1.0}
    assert torch.equal(hops[..., 11], MO.stack_hops(inp["graph_matrix"].double(), 13)[..., 11])


# ------------------------------------------------------------------------------------ GPU -----
def _load_params(model, params, dev):
    named = model.keras_named_parameters()
    Ly = len(model.layers.blocks)
    with torch.no_grad():
        if isinstance(model.node_emb, torch.nn.Parameter):
            model.node_emb.copy_(params["node_emb.embeddings"])
        else:
            model.node_emb.kernel.copy_(params["node_emb.kernel"]); model.node_emb.bias.copy_(params["node_emb.bias"])
            model.edge_emb.kernel.copy_(params["edge_emb.kernel"]); model.edge_emb.bias.copy_(params["edge_emb.bias"])
        if isinstance(model.fm_emb, torch.nn.Parameter):
            model.fm_emb.copy_(params["fm_emb.embeddings"])
        model.adj_emb.kernel.copy_(params["adj_emb.kernel"]); model.adj_emb.bias.copy_(params["adj_emb.bias"])
        model.node_norm_final.gamma.copy_(params["node_norm_final.gamma"]); model.node_norm_final.beta.copy_(params["node_norm_final.beta"])
        for i, m in enumerate(model.mlp_out):
            m.kernel.copy_(params[f"mlp_out_{i}.kernel"]); m.bias.copy_(params[f"mlp_out_{i}.bias"])
        model.target.kernel.copy_(params["target.kernel"]); model.target.bias.copy_(params["target.bias"])
        for ii in range(Ly):
            blk = model.layers.blocks[ii]
            for k in ("norm_edge.gamma", "norm_edge.beta", "attention_gates.kernel", "attention_gates.bias",
                      "dense_edge_b.kernel", "dense_edge_b.bias", "norm_mha.gamma", "norm_mha.beta", "dense_qkv.kernel",
                      "dense_qkv.bias", "dense_mha.kernel", "dense_mha.bias", "dense_edge_r.kernel", "dense_edge_r.bias"):
                m, a = k.split(".")
                getattr(getattr(blk, m), a).copy_(params[f"layer{ii}.{k}"])
            for tag, lst in (("node", model.layers.ffn_node), ("edge", model.layers.ffn_edge)):
                for a in ("norm_gamma", "norm_beta", "lr1_kernel", "lr1_bias", "lr2_kernel", "lr2_bias"):
                    getattr(lst[ii], a).copy_(params[f"layer{ii}.ffn_{tag}.{a}"])
    return named


def _grad_of(model, key):
    """oracle parameter name -> the module parameter"""
    if key == "node_emb.embeddings":
        return model.node_emb if isinstance(model.node_emb, torch.nn.Parameter) else None
    if key == "fm_emb.embeddings":
        return model.fm_emb
    parts = key.split(".")
    if parts[0].startswith("layer"):
        ii = int(parts[0][5:])
        if parts[1].startswith("ffn_"):
            lst = model.layers.ffn_node if parts[1] == "ffn_node" else model.layers.ffn_edge
            return getattr(lst[ii], parts[2])
        return getattr(getattr(model.layers.blocks[ii], parts[1]), parts[2])
    if parts[0].startswith("mlp_out_"):
        return getattr(model.mlp_out[int(parts[0][8:])], parts[1])
    if parts[0] == "edge_norm_final":
        return None
    return getattr(getattr(model, parts[0]), parts[1])


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,De,K,V", [(2, 9, 16, 4, 5), (3, 37, 64, 16, 5), (2, 64, 64, 16, 5), (1, 21, 48, 1, 3), (2, 70, 8, 7, 5),
                                       (2, 150, 8, 16, 5), (64, 150, 8, 4, 5), (1, 200, 8, 3, 5)])
def test_edge_embed_vs_oracle(B, N, De, K, V, gpu, egt_lib):
    """hop planes: k_hop_chain (one launch, adjacency + column block in LDS; B = 64 / N = 150 runs two column tiles per
    workgroup, B = 2 one) and the per-hop kernels where the adjacency does not fit (N = 200)"""
    from egt_amd import edge_embed
    from oracle import egt_model_oracle as MO, egt_oracle as O
    g = torch.Generator().manual_seed(B * 100 + N)
    adj = (torch.rand(B, N, N, generator=g) > 0.8).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float()
    fm = torch.where(adj > 0, torch.randint(0, V - 1, (B, N, N), generator=g), torch.tensor(-1))
    table = torch.randn(V, De, generator=g); W = torch.randn(K, De, generator=g) * 0.3; b = torch.randn(De, generator=g) * 0.2
    de = torch.randn(B, N, N, De, generator=g)
    t64, W64, b64 = (x.double().requires_grad_() for x in (table, W, b))
    hops_o = MO.stack_hops(adj.double(), K)
    e_o = O.dense(hops_o, W64, b64) + MO.neg1_masked_embedding(fm, t64)
    gr = torch.autograd.grad(e_o, [t64, W64, b64], de.double())
    tg, Wg, bg = (x.to(gpu).requires_grad_() for x in (table, W, b))
    e, hops = edge_embed(fm.to(gpu), adj.to(gpu), tg, Wg, bg, return_hops=True)
    assert torch.equal(hops.permute(1, 2, 3, 0).cpu(), hops_o.float()), "hop matrices of a 0/1 adjacency are exact"
    assert_close(e, e_o, name="e0", **FWD)
    e.backward(de.to(gpu))
    assert_close(tg.grad, gr[0], name="d fm_emb", **BWD)
    assert_close(Wg.grad, gr[1], name="d adj_emb.kernel", **BWD)
    assert_close(bg.grad, gr[2], name="d adj_emb.bias", **BWD)
    # weighted (non-binary) adjacency without clipping: fp32 contraction within tolerance
    adjw = adj * torch.rand(B, N, N, generator=g)
    _, hw = edge_embed(fm.to(gpu), adjw.to(gpu), tg, Wg, bg, clip_hops=False, return_hops=True)
    assert_close(hw.permute(1, 2, 3, 0), MO.stack_hops(adjw.double(), K, clip_hops=False), name="weighted hops", rtol=1e-4, arel=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CS.MODEL_CASES))
def test_zinc_model_vs_oracle_and_golden(name, gpu, egt_lib):
    from egt_amd import ZincDCTransformer, mae_loss
    inp, params, c = CS.make_model_case(name)
    model = ZincDCTransformer(random_mask_prob=0.0, **c["cfg"]).to(gpu).eval()
    _load_params(model, params, gpu)
    out, dparams = CS.model_oracle(inp, params, c)
    y = model(inp["node_features"].to(gpu), inp["feature_matrix"].to(gpu), inp["graph_matrix"].to(gpu))
    loss = mae_loss(y, inp["target"].to(gpu))
    loss.backward()
    assert_close(y, out["y"], name="prediction", rtol=2e-4, arel=5e-5)
    assert_close(loss.reshape(1), out["loss"], name="MAE", rtol=2e-4, arel=5e-5)
    gold = load_golden(os.path.join(CS.GOLDEN_DIR, f"model_{name}.npz"))
    assert_close(y, gold["out"]["y"], name="prediction vs golden", rtol=2e-4, arel=5e-5)
    dead = {id(p) for p in model._dead_edge_params()}
    for k, gref in dparams.items():
        prm = _grad_of(model, k)
        if prm is None:
            continue
        if gref is None or id(prm) in dead:
            assert prm.grad is None or float(prm.grad.abs().max()) == 0.0, k
            continue
        assert_close(prm.grad, gref, name=k, **BWD)
    for k, v in gold["dparams"].items():
        prm = _grad_of(model, k)
        if prm is not None and id(prm) not in dead:
            assert_close(prm.grad, v, name=f"{k} vs golden", **BWD)


@pytest.mark.gpu
def test_zinc_model_training_masks_and_names(gpu, egt_lib):
    """training mode: every block draws its random attention mask from the counter hash; the oracle gets the
    same sample (oracle/rng_ref.py).  Also: parameter names are the reference's Keras variable names."""
    from egt_amd import ZincDCTransformer, mae_loss
    from oracle import rng_ref
    inp, params, c = CS.make_model_case("zinc_small")
    p_rm = 0.3
    model = ZincDCTransformer(random_mask_prob=p_rm, seed=5, **c["cfg"]).to(gpu).train()
    _load_params(model, params, gpu)
    B, N = inp["node_features"].shape
    rms = []
    for blk in model.layers.blocks:
        m = blk.mha
        sd = (m.seed * 0x9E3779B97F4A7C15 + (m._calls + 1) * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        rms.append(torch.from_numpy(rng_ref.random_mask(sd, B, N, 8, p_rm)))
    out, dparams = CS.model_oracle(inp, params, c, rand_masks=rms)
    y = model(inp["node_features"].to(gpu), inp["feature_matrix"].to(gpu), inp["graph_matrix"].to(gpu))
    mae_loss(y, inp["target"].to(gpu)).backward()
    assert_close(y, out["y"], name="prediction (training)", rtol=2e-4, arel=5e-5)
    assert_close(model.layers.blocks[0].dense_qkv.kernel.grad, dparams["layer0.dense_qkv.kernel"], name="dWqkv0", **BWD)
    assert_close(model.fm_emb.grad, dparams["fm_emb.embeddings"], name="d fm_emb", **BWD)
    names = model.keras_named_parameters()
    for k in ("node_emb/embeddings", "fm_emb/embeddings", "adj_emb/kernel", "dense_qkv_00/kernel", "norm_edge_01/gamma",
              "fnn_lr1_edge_00/kernel", "norm_fnn_node_01/beta", "node_norm_final/gamma", "mlp_out_0/kernel", "target/bias"):
        assert k in names, k
    Ly = c["cfg"]["model_height"]
    assert f"dense_edge_r_{Ly - 1:0>2d}/kernel" not in names and f"fnn_lr1_edge_{Ly - 1:0>2d}/kernel" not in names


# ------------------------------------------------------------------------------ PATTERN (config 4) -----
def _pattern_case(B=3, N=13, seed=5):
    from oracle import egt_model_oracle as MO
    cfg = dict(model_width=32, edge_width=8, model_height=2, upto_hop=4, num_node_features=3, num_edge_features=0, num_targets=2)
    g = torch.Generator().manual_seed(seed)
    n = torch.tensor([N, 7, 10][:B])
    real = torch.arange(N)[None, :] < n[:, None]
    nf = torch.randint(0, 3, (B, N), generator=g); nf[~real] = -1
    adj = (torch.rand(B, N, N, generator=g) > 0.6).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float() * (real[:, :, None] & real[:, None, :]).float() * (1 - torch.eye(N))[None]
    y = torch.randint(0, 2, (B, N), generator=g); y[~real] = 0
    params = MO.init_zinc_params(cfg, dtype=torch.float32, generator=g)
    return cfg, nf, adj, y, params


def test_pattern_oracle_loss_semantics():
    from oracle import egt_model_oracle as MO
    w = MO.class_weights_from_sizes([979220, 209900])               # schemes/pattern/svd.py:20
    assert torch.allclose(w, torch.tensor([209900 / 1189120, 979220 / 1189120], dtype=torch.float64))
    logits = torch.zeros(1, 3, 2, dtype=torch.float64); y = torch.tensor([[0, 1, 1]]); mask = torch.tensor([[True, True, False]])
    loss = MO.weighted_sparse_xent_loss(logits, y, mask, w)          # uniform logits: xent = ln 2; padded slot weighs 0 but counts
    assert float(loss) == pytest.approx(float((w[0] + w[1]) * torch.log(torch.tensor(2.0, dtype=torch.float64)) / 3))


@pytest.mark.gpu
def test_pattern_model_vs_oracle(gpu, egt_lib):
    """PATTERN model (BASELINE config 4's dataset): edge_width 8, adjacency-only edge input, per-node logits, class-weighted
    cross-entropy -- logits, loss and every parameter gradient vs the fp64 oracle.  B * N * N is odd here: the width-8 edge
    FFN takes its padded-tail path."""
    from egt_amd import PatternDCTransformer, weighted_sparse_xent_loss, class_weights_from_sizes
    from oracle import egt_model_oracle as MO
    cfg, nf, adj, y, params = _pattern_case()
    model = PatternDCTransformer(model_width=32, model_height=2, upto_hop=4, random_mask_prob=0.0).to(gpu).eval()
    _load_params(model, params, gpu)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    lo, mo = MO.pattern_forward(nf, adj, p64, cfg)
    w64 = MO.class_weights_from_sizes([979220, 209900])
    loss_o = MO.weighted_sparse_xent_loss(lo, y, mo, w64)
    names = [k for k in p64 if k != "fm_emb.embeddings"]
    gro = torch.autograd.grad(loss_o, [p64[k] for k in names], allow_unused=True)
    logits, mask = model(nf.to(gpu), adj.to(gpu), return_mask=True)
    assert torch.equal(mask.cpu(), mo)
    loss = weighted_sparse_xent_loss(logits, y.to(gpu), mask, class_weights_from_sizes([979220, 209900], device=gpu))
    loss.backward()
    assert_close(logits, lo, name="logits", rtol=2e-4, arel=5e-5)
    assert_close(loss.reshape(1), loss_o.reshape(1), name="xent", rtol=2e-4, arel=5e-5)
    dead = {id(p) for p in model._dead_edge_params()}
    checked = 0
    for k, gref in zip(names, gro):
        prm = _grad_of(model, k)
        if prm is None or id(prm) in dead or gref is None:
            continue
        assert_close(prm.grad, gref, name=k, **BWD); checked += 1
    assert checked > 40
    assert "fm_emb/embeddings" not in model.keras_named_parameters() and not isinstance(model.fm_emb, torch.nn.Parameter)


# ------------------------------------------------------------------------------ CIFAR10 (config 3) -----
@pytest.mark.gpu
@pytest.mark.parametrize("B,N,De,K,F", [(2, 19, 8, 16, 1), (3, 30, 16, 4, 3), (1, 64, 64, 16, 4)])
def test_edge_embed_float_features_vs_oracle(B, N, De, K, F, gpu, egt_lib):
    """real-valued edge features: Masking(-1) + Dense ride as F planes behind the hop planes (cifar10/dc.py:70-73)"""
    from egt_amd import edge_embed
    from oracle import egt_model_oracle as MO, egt_oracle as O
    g = torch.Generator().manual_seed(B * 7 + N)
    adj = (torch.rand(B, N, N, generator=g) > 0.7).float(); adj = ((adj + adj.transpose(1, 2)) > 0).float()
    ff = torch.rand(B, N, N, F, generator=g)
    ff[adj == 0] = -1.0                                             # non-edges carry the mask value in every feature
    ff[0, 0, 1] = -1.0; ff[0, 0, 1, F - 1] = 0.5                    # a single differing feature keeps the pair (F > 1) / is the value (F = 1)
    Wa = torch.randn(K, De, generator=g) * 0.3; ba = torch.randn(De, generator=g) * 0.2
    We = torch.randn(F, De, generator=g); be = torch.randn(De, generator=g) * 0.2
    de = torch.randn(B, N, N, De, generator=g)
    p64 = [x.double().requires_grad_() for x in (Wa, ba, We, be)]
    xe, _ = MO.keras_masking(ff.double(), -1.0)
    e_o = O.dense(MO.stack_hops(adj.double(), K), p64[0], p64[1]) + O.dense(xe, p64[2], p64[3])
    gr = torch.autograd.grad(e_o, p64, de.double())
    pg = [x.to(gpu).requires_grad_() for x in (Wa, ba, We, be)]
    fm = torch.full((B, N, N), -1, dtype=torch.int32, device=gpu)
    e = edge_embed(fm, adj.to(gpu), torch.zeros(1, De, device=gpu), pg[0], pg[1], float_features=ff.to(gpu),
                   float_kernel=pg[2], float_bias=pg[3])
    e.backward(de.to(gpu))
    assert_close(e, e_o, name="e0", **FWD)
    for n, a, b in zip(("d adj_emb.kernel", "d adj_emb.bias", "d edge_emb.kernel", "d edge_emb.bias"), pg, gr):
        assert_close(a.grad, b, name=n, **BWD)


@pytest.mark.gpu
def test_cifar10_model_vs_oracle(gpu, egt_lib):
    """CIFAR10 model (BASELINE config 3's dataset): Masking + Dense embeddings of real-valued node / edge features,
    edge_width 8, graph-level logits, sparse categorical cross-entropy -- logits, loss, every parameter gradient."""
    from egt_amd import Cifar10DCTransformer, sparse_xent_loss
    from oracle import egt_model_oracle as MO
    cfg = dict(model_width=32, edge_width=8, model_height=2, upto_hop=4, num_node_features=1, num_edge_features=0, num_targets=10,
               float_node_features=5, float_edge_features=1)
    g = torch.Generator().manual_seed(21)
    B, N = 4, 18
    n = torch.tensor([18, 11, 15, 9]); real = torch.arange(N)[None, :] < n[:, None]
    nf = torch.rand(B, N, 5, generator=g); nf[~real] = -1.0
    adj = (torch.rand(B, N, N, generator=g) > 0.6).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float() * (real[:, :, None] & real[:, None, :]).float() * (1 - torch.eye(N))[None]
    fm = torch.rand(B, N, N, 1, generator=g); fm[adj == 0] = -1.0
    y = torch.randint(0, 10, (B,), generator=g)
    params = MO.init_zinc_params(cfg, dtype=torch.float32, generator=g)
    model = Cifar10DCTransformer(model_width=32, model_height=2, upto_hop=4, random_mask_prob=0.0).to(gpu).eval()
    _load_params(model, params, gpu)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    lo = MO.cifar10_forward(nf, fm, adj, p64, cfg)
    loss_o = MO.sparse_xent_loss(lo, y)
    names = [k for k in p64 if k not in ("fm_emb.embeddings", "node_emb.embeddings")]
    gro = torch.autograd.grad(loss_o, [p64[k] for k in names], allow_unused=True)
    logits = model(nf.to(gpu), fm.to(gpu), adj.to(gpu))
    loss = sparse_xent_loss(logits, y.to(gpu))
    loss.backward()
    assert_close(logits, lo, name="logits", rtol=2e-4, arel=5e-5)
    assert_close(loss.reshape(1), loss_o.reshape(1), name="xent", rtol=2e-4, arel=5e-5)
    dead = {id(p) for p in model._dead_edge_params()}
    checked = 0
    for k, gref in zip(names, gro):
        if k.startswith("node_emb.") or k.startswith("edge_emb."):
            prm = getattr(getattr(model, k.split(".")[0]), k.split(".")[1])
        else:
            prm = _grad_of(model, k)
        if prm is None or id(prm) in dead or gref is None:
            continue
        assert_close(prm.grad, gref, name=k, **BWD); checked += 1
    assert checked > 44
    names_k = model.keras_named_parameters()
    assert "node_emb/kernel" in names_k and "edge_emb/kernel" in names_k and "node_emb/embeddings" not in names_k


# ------------------------------------------------------------------ positional encodings (BASELINE configs 1 and 3) ---
def _load_pe(model, params):
    with torch.no_grad():
        for n in ("svd_emb", "eig_emb"):
            if hasattr(model, n):
                getattr(model, n).kernel.copy_(params[f"{n}.kernel"]); getattr(model, n).bias.copy_(params[f"{n}.bias"])


@pytest.mark.gpu
@pytest.mark.parametrize("train", [False, True])
def test_zinc_eig_model_vs_oracle(train, gpu, egt_lib):
    """BASELINE config 1 (configs/main/zinc/100k/egt_epe.json: model_width 48 -> d = 6, edge_width 48, use_eig, sel_eig_features 8,
    transform_eig False, random_neg True): eigenvectors zero-padded to model_width and ADDED to the node embedding
    (graph_model_base.py:388-414), in training with the sign flip of RandomNegEig (misc.py:76-94; the sample injected)."""
    from egt_amd import ZincDCTransformer, mae_loss
    from oracle import egt_model_oracle as MO
    cfg = dict(model_width=48, edge_width=48, model_height=2, upto_hop=4, use_eig=True, num_eig_features=20, sel_eig_features=8,
               transform_eig=False, random_neg=True)
    g = torch.Generator().manual_seed(77)
    B, N = 3, 14
    n = torch.tensor([14, 9, 11]); real = torch.arange(N)[None, :] < n[:, None]
    nf = torch.randint(0, 28, (B, N), generator=g); nf[~real] = -1
    adj = (torch.rand(B, N, N, generator=g) > 0.7).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float() * (real[:, :, None] & real[:, None, :]).float() * (1 - torch.eye(N))[None]
    fm = torch.where(adj > 0, torch.randint(0, 4, (B, N, N), generator=g), torch.tensor(-1))
    ev = torch.randn(B, N, 20, generator=g) * real[..., None]
    tgt = torch.randn(B, 1, generator=g)
    signs = MO.random_neg_signs(torch.rand(B, 1, 48, generator=g)) if train else None
    params = MO.init_zinc_params(cfg, dtype=torch.float32, generator=g)
    model = ZincDCTransformer(random_mask_prob=0.0, **cfg).to(gpu)
    model.train(train)
    _load_params(model, params, gpu)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    yo = MO.zinc_forward(nf, fm, adj, p64, cfg, pe=dict(eigen_vectors=ev, eig_signs=signs))
    lo = MO.mae_loss(yo, tgt.double())
    names = list(p64)
    gro = torch.autograd.grad(lo, [p64[k] for k in names], allow_unused=True)
    y = model(nf.to(gpu), fm.to(gpu), adj.to(gpu), eigen_vectors=ev.to(gpu), pe_signs=None if signs is None else signs.to(gpu))
    loss = mae_loss(y, tgt.to(gpu)); loss.backward()
    assert_close(y, yo, name="prediction", rtol=2e-4, arel=5e-5)
    dead = {id(p) for p in model._dead_edge_params()}
    n_ok = 0
    for k, gref in zip(names, gro):
        prm = _grad_of(model, k)
        if prm is None or id(prm) in dead or gref is None:
            continue
        assert_close(prm.grad, gref, name=k, **BWD); n_ok += 1
    assert n_ok > 40
    # the encoding matters (a model that dropped it would still pass a test without this line)
    y0 = MO.zinc_forward(nf, fm, adj, {k: v.double() for k, v in params.items()}, dict(cfg, use_eig=False))
    assert float((y0 - yo.detach()).abs().max()) > 1e-3
    with pytest.raises(ValueError):
        model(nf.to(gpu), fm.to(gpu), adj.to(gpu))          # use_eig without the input


@pytest.mark.gpu
def test_cifar10_svd_model_vs_oracle(gpu, egt_lib):
    """BASELINE config 3 as shipped (configs/main/cifar10/100k/egt_spe.json: use_svd, sel_svd_features 8, random_neg, scheme
    transform_svd=True): [B,N,F,2] singular pairs -> first 8 -> RandomNeg sign per (graph, pair) -> [U | V] -> Dense 'svd_emb'
    -> added to the node embedding (graph_model_base.py:322-349); logits and every gradient incl. svd_emb."""
    from egt_amd import Cifar10DCTransformer, sparse_xent_loss
    from oracle import egt_model_oracle as MO
    cfg = dict(model_width=64, edge_width=8, model_height=2, upto_hop=4, num_node_features=1, num_edge_features=0, num_targets=10,
               float_node_features=5, float_edge_features=1, use_svd=True, num_svd_features=16, sel_svd_features=8,
               transform_svd=True, random_neg=True)
    g = torch.Generator().manual_seed(23)
    B, N = 3, 20
    n = torch.tensor([20, 13, 16]); real = torch.arange(N)[None, :] < n[:, None]
    nf = torch.rand(B, N, 5, generator=g); nf[~real] = -1.0
    adj = (torch.rand(B, N, N, generator=g) > 0.6).float()
    adj = ((adj + adj.transpose(1, 2)) > 0).float() * (real[:, :, None] & real[:, None, :]).float() * (1 - torch.eye(N))[None]
    fm = torch.rand(B, N, N, 1, generator=g); fm[adj == 0] = -1.0
    sv = torch.randn(B, N, 16, 2, generator=g) * real[..., None, None]
    y = torch.randint(0, 10, (B,), generator=g)
    signs = MO.random_neg_signs(torch.rand(B, 1, 8, 1, generator=g))
    params = MO.init_zinc_params(cfg, dtype=torch.float32, generator=g)
    model = Cifar10DCTransformer(model_width=64, model_height=2, upto_hop=4, random_mask_prob=0.0, use_svd=True, num_svd_features=16,
                                 sel_svd_features=8, transform_svd=True, random_neg=True).to(gpu).train()
    _load_params(model, params, gpu); _load_pe(model, params)
    p64 = {k: v.double().requires_grad_() for k, v in params.items()}
    lo = MO.cifar10_forward(nf, fm, adj, p64, cfg, pe=dict(singular_vectors=sv, svd_signs=signs))
    loss_o = MO.sparse_xent_loss(lo, y)
    gk, gb = torch.autograd.grad(loss_o, [p64["svd_emb.kernel"], p64["svd_emb.bias"]])
    logits = model(nf.to(gpu), fm.to(gpu), adj.to(gpu), singular_vectors=sv.to(gpu), pe_signs=signs.to(gpu))
    sparse_xent_loss(logits, y.to(gpu)).backward()
    assert_close(logits, lo, name="logits", rtol=2e-4, arel=5e-5)
    assert_close(model.svd_emb.kernel.grad, gk, name="svd_emb.kernel", **BWD)
    assert_close(model.svd_emb.bias.grad, gb, name="svd_emb.bias", **BWD)
    assert "svd_emb/kernel" in model.keras_named_parameters()
    # device-drawn signs: a flip of a whole (graph, pair) column, so |PE contribution| statistics are unchanged; eval: no flip
    model.eval()
    l_eval = model(nf.to(gpu), fm.to(gpu), adj.to(gpu), singular_vectors=sv.to(gpu))
    lo_eval = MO.cifar10_forward(nf, fm, adj, {k: v.double() for k, v in params.items()}, cfg, pe=dict(singular_vectors=sv))
    assert_close(l_eval, lo_eval, name="eval logits (no sign flip)", rtol=2e-4, arel=5e-5)


@pytest.mark.gpu
def test_constrained_model_builds_its_edge_mask(gpu, egt_lib):
    """edge_channel_type='constrained' (18 reference configs): the model tiles the adjacency into the attention mask itself
    (AdjMatModel.get_edge_mask, graph_model_base.py:131-142) -- ADVICE r2: the scheme's batch_loss never passes one."""
    from egt_amd import ZincDCTransformer
    from oracle import egt_model_oracle as MO
    from oracle import egt_oracle as O
    inp, params, c = CS.make_model_case("zinc_small")
    cfg = dict(c["cfg"], edge_channel_type="constrained")
    model = ZincDCTransformer(random_mask_prob=0.0, **cfg).to(gpu).eval()
    _load_params(model, params, gpu)
    y = model(inp["node_features"].to(gpu), inp["feature_matrix"].to(gpu), inp["graph_matrix"].to(gpu))
    # oracle: the residual model with M = constrained_edge_mask(adjacency) in every block
    p = {k: v.double() for k, v in params.items()}
    h, e, mask = MO.zinc_embeddings(inp["node_features"], inp["feature_matrix"], inp["graph_matrix"], p, cfg)
    M = O.constrained_edge_mask(inp["graph_matrix"].double(), 8)
    for ii in range(cfg["model_height"]):
        bp = {k[len(f"layer{ii}."):]: v for k, v in p.items() if k.startswith(f"layer{ii}.") and ".ffn_" not in k}
        h, e = O.block_forward(h, e, mask, bp, num_heads=8, attn_mask=M, edge_channel_type="constrained")
        e = O.ffn_forward(e, {k.split(".", 2)[2]: v for k, v in p.items() if k.startswith(f"layer{ii}.ffn_edge.")}, activation="elu")
        h = O.ffn_forward(h, {k.split(".", 2)[2]: v for k, v in p.items() if k.startswith(f"layer{ii}.ffn_node.")}, activation="elu")
    h = O.layer_norm(h, p["node_norm_final.gamma"], p["node_norm_final.beta"])
    x = MO.mlp_out(MO.masked_global_avg_pool_1d(h, mask), p, 2, "elu")
    assert_close(y, O.dense(x, p["target.kernel"], p["target.bias"]), name="constrained prediction", rtol=2e-4, arel=5e-5)


@pytest.mark.gpu
def test_unbuilt_model_keys_are_refused(gpu, egt_lib):
    from egt_amd import ZincDCTransformer
    for kw in (dict(l2_reg=1e-4), dict(node_dropout=0.1), dict(distance_loss=0.5), dict(readout_edges=True)):
        with pytest.raises(NotImplementedError):
            ZincDCTransformer(model_width=16, edge_width=16, model_height=1, **kw)
    with pytest.raises(TypeError):
        ZincDCTransformer(model_width=16, edge_width=16, model_height=1, not_a_reference_key=1)


# ------------------------------------------------------------------ the data pipeline feeding the HIP model ---
@pytest.mark.gpu
def test_data_pipeline_batches_through_hip_model(gpu, egt_lib, tmp_path):
    """PackedStore -> GraphDataset (record maps, Laplacian eigenvectors, per-batch padding) -> the scheme driver -> the HIP
    model: two epochs of zinc.eig on a small synthetic store train (loss falls) and evaluate, and one batch of the pipeline
    gives the oracle's prediction."""
    from egt_amd import training as T
    from oracle import egt_model_oracle as MO
    from test_data import _store
    store = _store(tmp_path, n_train=48, n_val=16)
    cfg = dict(scheme="zinc.eig", model_name="p", num_epochs=2, initial_lr=2e-3, batch_size=16, model_width=48, edge_width=48,
               model_height=2, upto_hop=4, random_mask_prob=0.0, dataset_path=store, save_path=str(tmp_path / "run"), sel_eig_features=8)
    logs = []
    s = T.ZincEigScheme(cfg, device=gpu, print_fn=logs.append)
    s.execute_training()
    assert s.state.global_step == 6 and np.isfinite(s.history[-1]["val_mae"])
    assert s.history[-1]["loss"] < s.history[0]["loss"]
    b = next(iter(s.valset))
    assert b["eigen_vectors"].shape[-1] == 20 and b["graph_matrix"].dim() == 3 and "singular_vectors" not in b
    s.model.eval()
    y = s.model(b["node_features"].to(gpu), b["feature_matrix"].to(gpu), b["graph_matrix"].to(gpu), eigen_vectors=b["eigen_vectors"].to(gpu))
    # the same batch through the oracle with the trained weights
    named = s.model.keras_named_parameters()
    p = {}
    for k, v in named.items():
        lay, var = k.split("/")
        m = __import__("re").match(r"(.+)_(\d\d)$", lay)
        v = v.detach().double().cpu()
        if m and m.group(1).startswith(("norm_fnn_", "fnn_lr1_", "fnn_lr2_")):
            kind, tag = m.group(1).rsplit("_", 1)
            nm = {"norm_fnn": "norm_", "fnn_lr1": "lr1_", "fnn_lr2": "lr2_"}[kind] + var
            p[f"layer{int(m.group(2))}.ffn_{tag}.{nm}"] = v
        elif m:
            p[f"layer{int(m.group(2))}.{m.group(1)}.{var}"] = v
        else:
            p[f"{lay}.{var}"] = v
    Ly = 2
    for a in ("norm_gamma", "norm_beta", "lr1_kernel", "lr1_bias", "lr2_kernel", "lr2_bias"):   # dead in the Keras model: any value
        p[f"layer{Ly - 1}.ffn_edge.{a}"] = getattr(s.model.layers.ffn_edge[-1], a).detach().double().cpu()
    p[f"layer{Ly - 1}.dense_edge_r.kernel"] = s.model.layers.blocks[-1].dense_edge_r.kernel.detach().double().cpu()
    p[f"layer{Ly - 1}.dense_edge_r.bias"] = s.model.layers.blocks[-1].dense_edge_r.bias.detach().double().cpu()
    p["edge_norm_final.gamma"] = torch.ones(48, dtype=torch.float64); p["edge_norm_final.beta"] = torch.zeros(48, dtype=torch.float64)
    ocfg = dict(model_width=48, edge_width=48, model_height=2, upto_hop=4, use_eig=True, sel_eig_features=8, transform_eig=False)
    yo = MO.zinc_forward(b["node_features"], b["feature_matrix"], b["graph_matrix"], p, ocfg, pe=dict(eigen_vectors=b["eigen_vectors"]))
    assert_close(y, yo, name="pipeline batch vs oracle", rtol=5e-4, arel=1e-4)
