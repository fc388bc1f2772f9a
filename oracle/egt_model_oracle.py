"""CPU oracle of the whole ZINC model around the hot path — TEST INFRASTRUCTURE.

A torch-CPU restatement (any dtype; tests evaluate it in fp64) of what `DCSVDTransformer`
builds for scheme `zinc.svd` with the shipped ZINC configs (`use_svd: false`), composed from
oracle/egt_oracle.py for the layer stack.  Parity is UNPINNED by the reference (TensorFlow cannot
run here; the reference ships no vectors): correctness is by construction from the cited lines.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this module.

Model (all citations relative to /root/reference/):
  inputs   node_features [B,N] int (padding -1), feature_matrix [B,N,N] int (-1 = no edge / padding),
           graph_matrix [B,N,N] float adjacency           lib/models/zinc/dc.py:37-59, graph_model_base.py:42-52
  h0       Neg1MaskedEmbedding(num_node_features+1, Dh)(nodef)           zinc/dc.py:66-69, masking.py:5-43
  e0       Neg1MaskedEmbedding(num_edge_features+1, De)(fmat)            zinc/dc.py:70-73
           + Dense(De)(stack_hops(adj, upto_hop, clip))                  graph_model_base.py:97-127
           (Add of the edge embeddings)                                  graph_xformer_model_base.py:401-409
  mask     node_features != -1 (Embedding(mask_zero) on inputs+1)        masking.py:35-43
  layers   for ii: edge_update_residual; ffn_block                       graph_xformer_model_base.py:336-341
  final    node_norm_final / edge_norm_final                             :343-347
  readout  masked GlobalAveragePooling1D -> mlp_out (elu) -> Dense(1)    zinc/dc.py:100-120, :354-372
  loss     MeanAbsoluteError                                             lib/training/schemes/zinc/svd.py:37-39
"""
from __future__ import annotations

import math

import torch

from . import egt_oracle as O


def neg1_masked_embedding(x, table):
    """Neg1MaskedEmbedding.call (masking.py:31-40): Embedding lookup of inputs+1 (padding -1 -> row 0,
    which is an ordinary trainable row: mask_zero only produces the mask)."""
    return table[(x + 1).long()]


def stack_hops(adj, upto_hop, clip_hops=True):
    """AdjMatModel.create_embedding('adj') (graph_model_base.py:101-119): [A, clip(A.A), clip(A.clip(A.A)), ...]
    stacked on a new last axis; upto_hop == 1 is A[..., None]."""
    if upto_hop < 1:
        raise ValueError
    hops = [adj]
    hop = adj
    for _ in range(upto_hop - 1):
        hop = torch.matmul(adj, hop)
        if clip_hops:
            hop = hop.clamp(0.0, 1.0)
        hops.append(hop)
    return torch.stack(hops, dim=-1)


def masked_global_avg_pool_1d(h, mask):
    """keras GlobalAveragePooling1D with a mask: sum(h * m) / sum(m) over the node axis (zinc/dc.py:109)."""
    m = mask.to(h.dtype)[..., None]
    return (h * m).sum(dim=1) / m.sum(dim=1)


def activation_fn(x, name):
    if name == "elu":
        return torch.nn.functional.elu(x)
    if name == "relu":
        return torch.relu(x)
    raise ValueError(name)


def mlp_out(x, params, n_layers, activation="elu"):
    """GraphTransformerBase.mlp_out (graph_xformer_model_base.py:354-372)."""
    for ii in range(n_layers):
        x = activation_fn(O.dense(x, params[f"mlp_out_{ii}.kernel"], params[f"mlp_out_{ii}.bias"]), activation)
    return x


def random_neg_signs(uniform):
    """RandomNeg / RandomNegEig (lib/base/xformer_layers/misc.py:53-94): signs = where(U < 0.5, -1, 1) with U of shape
    [B,1,F,1] (SVD: one sign per graph and singular pair, shared by the nodes and by U / V) or [B,1,F] (eigenvectors)."""
    return torch.where(uniform < 0.5, -torch.ones_like(uniform), torch.ones_like(uniform))


def svd_embedding(singular_vectors, p, cfg, signs=None):
    """SVDFeatModel.create_embedding('svdf') (lib/models/graph_model_base.py:322-349): the first sel_svd_features singular
    pairs [B,N,sf,2] (zero-padded to model_width//2 when there is no transform), the training-time sign flip (signs
    [B,1,F,1] or None), U and V parts concatenated along the feature axis, Dense 'svd_emb' when transform_svd."""
    mw, sf = cfg["model_width"], cfg["sel_svd_features"]
    v = singular_vectors[:, :, :sf, :]
    if not cfg.get("transform_svd", False):
        pad = max(0, mw // 2 - sf)
        v = torch.nn.functional.pad(v, (0, 0, 0, pad))
    if signs is not None:
        v = v * signs.to(v.dtype)
    v = torch.cat(torch.unbind(v, dim=-1), dim=-1)                                            # tf.concat(tf.unstack(v, axis=-1), axis=-1)
    if cfg.get("transform_svd", False):
        v = O.dense(v, p["svd_emb.kernel"], p["svd_emb.bias"])
    return v


def eig_embedding(eigen_vectors, p, cfg, signs=None):
    """EigFeatModel.create_embedding('eigf') (lib/models/graph_model_base.py:388-414): the first sel_eig_features Laplacian
    eigenvectors [B,N,sf], zero-padded to model_width when there is no transform, sign flip (signs [B,1,F]), Dense 'eig_emb'
    when transform_eig."""
    mw, sf = cfg["model_width"], cfg["sel_eig_features"]
    v = eigen_vectors[:, :, :sf]
    if not cfg.get("transform_eig", False):
        v = torch.nn.functional.pad(v, (0, max(0, mw - sf)))
    if signs is not None:
        v = v * signs.to(v.dtype)
    if cfg.get("transform_eig", False):
        v = O.dense(v, p["eig_emb.kernel"], p["eig_emb.bias"])
    return v


def add_positional(h, p, cfg, pe):
    """combine_node_embeddings (graph_xformer_model_base.py:390-399): Add()([node embedding, PE embedding]).  pe: dict with
    'singular_vectors' / 'eigen_vectors' and optional 'svd_signs' / 'eig_signs' (the injected sample of the sign flip)."""
    if pe is None:
        return h
    if cfg.get("use_svd", False):
        h = h + svd_embedding(pe["singular_vectors"].to(h.dtype), p, cfg, pe.get("svd_signs"))
    if cfg.get("use_eig", False):
        h = h + eig_embedding(pe["eigen_vectors"].to(h.dtype), p, cfg, pe.get("eig_signs"))
    return h


def init_zinc_params(cfg, *, dtype=torch.float32, generator=None, randomize=True):
    """Keras defaults: Embedding 'uniform' U(-0.05, 0.05), Dense glorot_uniform / zeros, LN ones / zeros.
    randomize: perturb biases and LN parameters so the parity tests exercise them."""
    Dh, De, H, Ly = cfg["model_width"], cfg["edge_width"], cfg.get("num_heads", 8), cfg["model_height"]
    g = generator

    def glorot(fi, fo):
        lim = math.sqrt(6.0 / (fi + fo))
        return (torch.rand(fi, fo, generator=g, dtype=torch.float64) * 2 - 1) * lim

    def vec(n, base):
        if randomize:
            return base + 0.2 * torch.randn(n, generator=g, dtype=torch.float64)
        return torch.full((n,), float(base), dtype=torch.float64)

    p = {
        "node_emb.embeddings": torch.rand(cfg.get("num_node_features", 28) + 1, Dh, generator=g, dtype=torch.float64) * 0.1 - 0.05,
        "fm_emb.embeddings": torch.rand(cfg.get("num_edge_features", 4) + 1, De, generator=g, dtype=torch.float64) * 0.1 - 0.05,
        "adj_emb.kernel": glorot(cfg["upto_hop"], De), "adj_emb.bias": vec(De, 0.0),
        "node_norm_final.gamma": vec(Dh, 1.0), "node_norm_final.beta": vec(Dh, 0.0),
        "edge_norm_final.gamma": vec(De, 1.0), "edge_norm_final.beta": vec(De, 0.0),
    }
    if cfg.get("use_svd") and cfg.get("transform_svd"):
        p["svd_emb.kernel"] = glorot(2 * cfg["sel_svd_features"], Dh); p["svd_emb.bias"] = vec(Dh, 0.0)
    if cfg.get("use_eig") and cfg.get("transform_eig"):
        p["eig_emb.kernel"] = glorot(cfg["sel_eig_features"], Dh); p["eig_emb.bias"] = vec(Dh, 0.0)
    if cfg.get("float_node_features"):   # CIFAR10 / MNIST: Dense embeddings of real-valued features (cifar10/dc.py:66-73)
        p["node_emb.kernel"] = glorot(cfg["float_node_features"], Dh); p["node_emb.bias"] = vec(Dh, 0.0)
        p["edge_emb.kernel"] = glorot(cfg.get("float_edge_features", 1), De); p["edge_emb.bias"] = vec(De, 0.0)
    if randomize:   # embeddings of O(1) so the layer norms see real signal
        p["node_emb.embeddings"] = p["node_emb.embeddings"] * 20
        p["fm_emb.embeddings"] = p["fm_emb.embeddings"] * 20
    w = Dh
    for ii, f in enumerate(cfg.get("mlp_layers", [0.5, 0.25])):
        wo = round(f * Dh)
        p[f"mlp_out_{ii}.kernel"] = glorot(w, wo)
        p[f"mlp_out_{ii}.bias"] = vec(wo, 0.0)
        w = wo
    p["target.kernel"] = glorot(w, cfg.get("num_targets", 1))
    p["target.bias"] = vec(cfg.get("num_targets", 1), 0.0)
    for ii in range(Ly):
        for k, v in O.init_block_params(Dh, De, H, dtype=torch.float64, generator=g, randomize_norm=randomize).items():
            p[f"layer{ii}.{k}"] = v
        for tag, W in (("node", Dh), ("edge", De)):
            hid = round(W * cfg.get("ffn_multiplier", 2.0))
            p[f"layer{ii}.ffn_{tag}.norm_gamma"] = vec(W, 1.0)
            p[f"layer{ii}.ffn_{tag}.norm_beta"] = vec(W, 0.0)
            p[f"layer{ii}.ffn_{tag}.lr1_kernel"] = glorot(W, hid)
            p[f"layer{ii}.ffn_{tag}.lr1_bias"] = vec(hid, 0.0)
            p[f"layer{ii}.ffn_{tag}.lr2_kernel"] = glorot(hid, W)
            p[f"layer{ii}.ffn_{tag}.lr2_bias"] = vec(W, 0.0)
    return {k: v.to(dtype) for k, v in p.items()}


def class_weights_from_sizes(class_sizes):
    """WeightedSparseXEntropyLoss.__init__ (lib/base/genutil/losses.py:41-46): w = (sum - sizes) / sum(sum - sizes)."""
    cs = torch.as_tensor(class_sizes, dtype=torch.float64)
    w = cs.sum() - cs
    return w / w.sum()


def weighted_sparse_xent_loss(logits, y_true, mask, class_weights):
    """schemes/pattern/svd.py:34-39 + losses.py:5-23: per-node w[y] * sparse_categorical_crossentropy(from_logits);
    Keras multiplies the per-node losses by the output's mask (the node mask travels with h) and reduces with
    SUM_OVER_BATCH_SIZE = sum / number of (batch x node) elements, padded ones included (TF 2.1
    losses_utils.compute_weighted_loss)."""
    logp = torch.log_softmax(logits, dim=-1)
    y = y_true.clamp(min=0).long()
    xent = -logp.gather(-1, y[..., None])[..., 0]
    w = class_weights.to(logits.dtype)[y]
    per = w * xent * mask.to(logits.dtype)
    return per.sum() / per.numel()


def pattern_forward(node_features, graph_matrix, p, cfg, rand_masks=None, pe=None):
    """lib/models/sbm_pattern/dc.py:14-61 (DCSVDTransformer, use_svd false): node embedding, adjacency hop embedding as the
    only edge channel input, the layer loop, final norm, per-node mlp_out + Dense(num_target_labels) -> logits [B,N,C]."""
    H, Ly = cfg.get("num_heads", 8), cfg["model_height"]
    act = cfg.get("activation", "elu")
    dt = p["node_emb.embeddings"].dtype
    h = neg1_masked_embedding(node_features, p["node_emb.embeddings"])                      # sbm_pattern/dc.py:45-48
    h = add_positional(h, p, cfg, pe)
    hops = stack_hops(graph_matrix.to(dt), cfg["upto_hop"], cfg.get("clip_hops", True))
    e = O.dense(hops, p["adj_emb.kernel"], p["adj_emb.bias"])                               # graph_model_base.py:97-127
    mask = O.node_mask_from_features(node_features)
    for ii in range(Ly):
        bp = {k[len(f"layer{ii}."):]: v for k, v in p.items() if k.startswith(f"layer{ii}.") and ".ffn_" not in k}
        rm = None if rand_masks is None else rand_masks[ii]
        h, e = O.block_forward(h, e, mask, bp, num_heads=H, rand_mask=rm)
        fn = {k.split(".", 2)[2]: v for k, v in p.items() if k.startswith(f"layer{ii}.ffn_node.")}
        fe = {k.split(".", 2)[2]: v for k, v in p.items() if k.startswith(f"layer{ii}.ffn_edge.")}
        e = O.ffn_forward(e, fe, activation=act)
        h = O.ffn_forward(h, fn, activation=act)
    if cfg.get("do_final_norm", True):
        h = O.layer_norm(h, p["node_norm_final.gamma"], p["node_norm_final.beta"])
    x = mlp_out(h, p, len(cfg.get("mlp_layers", [0.5, 0.25])), act)                         # :55 (per node)
    return O.dense(x, p["target.kernel"], p["target.bias"]), mask                           # :56-58


def keras_masking(x, mask_value=-1.0):
    """keras.layers.Masking: a step whose features ALL equal mask_value is zeroed and masked out
    (outputs = inputs * any(inputs != mask_value, axis=-1)); returns (masked inputs, boolean mask)."""
    keep = (x != mask_value).any(dim=-1)
    return x * keep[..., None].to(x.dtype), keep


def cifar10_forward(node_features, feature_matrix, graph_matrix, p, cfg, rand_masks=None, pe=None):
    """lib/models/cifar10/dc.py:14-122 (DCSVDTransformer, use_svd false; scheme cifar10.svd): real-valued node features
    [B,N,5] and edge features [B,N,N,1] through Masking(-1) + Dense (:66-73), the adjacency hop embedding added to the
    edge embedding, the layer loop, final norm, masked GlobalAveragePooling1D, mlp_out, Dense(num_target_labels) -> logits."""
    H, Ly = cfg.get("num_heads", 8), cfg["model_height"]
    act = cfg.get("activation", "elu")
    dt = p["node_emb.kernel"].dtype
    mv = cfg.get("mask_value", -1.0)
    xn, mask = keras_masking(node_features.to(dt), mv)
    h = O.dense(xn, p["node_emb.kernel"], p["node_emb.bias"])                                # :68-70
    h = add_positional(h, p, cfg, pe)
    xe, _ = keras_masking(feature_matrix.to(dt), mv)
    e = O.dense(xe, p["edge_emb.kernel"], p["edge_emb.bias"])                                # :71-73
    hops = stack_hops(graph_matrix.to(dt), cfg["upto_hop"], cfg.get("clip_hops", True))
    e = e + O.dense(hops, p["adj_emb.kernel"], p["adj_emb.bias"])                            # edge_emb_add
    for ii in range(Ly):
        bp = {k[len(f"layer{ii}."):]: v for k, v in p.items() if k.startswith(f"layer{ii}.") and ".ffn_" not in k}
        rm = None if rand_masks is None else rand_masks[ii]
        h, e = O.block_forward(h, e, mask, bp, num_heads=H, rand_mask=rm)
        fn = {k.split(".", 2)[2]: v for k, v in p.items() if k.startswith(f"layer{ii}.ffn_node.")}
        fe = {k.split(".", 2)[2]: v for k, v in p.items() if k.startswith(f"layer{ii}.ffn_edge.")}
        e = O.ffn_forward(e, fe, activation=act)
        h = O.ffn_forward(h, fn, activation=act)
    if cfg.get("do_final_norm", True):
        h = O.layer_norm(h, p["node_norm_final.gamma"], p["node_norm_final.beta"])
    x = masked_global_avg_pool_1d(h, mask)                                                   # :107
    x = mlp_out(x, p, len(cfg.get("mlp_layers", [0.5, 0.25])), act)
    return O.dense(x, p["target.kernel"], p["target.bias"])                                  # :118-120


def sparse_xent_loss(logits, y_true):
    """keras.losses.SparseCategoricalCrossentropy(from_logits=True) (schemes/cifar10/svd.py:37-40): batch mean."""
    return -torch.log_softmax(logits, dim=-1).gather(-1, y_true.long()[..., None])[..., 0].mean()


def zinc_embeddings(node_features, feature_matrix, graph_matrix, p, cfg, pe=None):
    """get_embeddings (graph_xformer_model_base.py:411-431) for the zinc.svd model without SVD features."""
    dt = p["node_emb.embeddings"].dtype
    h = neg1_masked_embedding(node_features, p["node_emb.embeddings"])                      # zinc/dc.py:66-69
    h = add_positional(h, p, cfg, pe)                                                       # node_emb_add
    e_fm = neg1_masked_embedding(feature_matrix, p["fm_emb.embeddings"])                    # zinc/dc.py:70-73
    hops = stack_hops(graph_matrix.to(dt), cfg["upto_hop"], cfg.get("clip_hops", True))     # graph_model_base.py:101-119
    e_adj = O.dense(hops, p["adj_emb.kernel"], p["adj_emb.bias"])                           # :125-126
    e = e_adj + e_fm                                                                        # edge_emb_add, :403-404
    mask = O.node_mask_from_features(node_features)                                         # masking.py:42-43
    return h, e, mask


def zinc_forward(node_features, feature_matrix, graph_matrix, p, cfg, rand_masks=None, return_hidden=False, pe=None):
    """DCSVDTransformer.call -> prediction [B, num_targets] (graph_xformer_model_base.py:447-466).
    rand_masks: per-layer injected random attention masks (training with random_mask_prob > 0)."""
    H, Ly = cfg.get("num_heads", 8), cfg["model_height"]
    act = cfg.get("activation", "elu")
    h, e, mask = zinc_embeddings(node_features, feature_matrix, graph_matrix, p, cfg, pe)
    for ii in range(Ly):
        bp = {k[len(f"layer{ii}."):]: v for k, v in p.items()
              if k.startswith(f"layer{ii}.") and ".ffn_" not in k}
        rm = None if rand_masks is None else rand_masks[ii]
        h, e = O.block_forward(h, e, mask, bp, num_heads=H, rand_mask=rm)                   # layer/ii/attention, :338-339
        fn = {k.split(".", 2)[2]: v for k, v in p.items() if k.startswith(f"layer{ii}.ffn_node.")}
        fe = {k.split(".", 2)[2]: v for k, v in p.items() if k.startswith(f"layer{ii}.ffn_edge.")}
        e = O.ffn_forward(e, fe, activation=act)                                            # layer/ii/ffn, :309-324
        h = O.ffn_forward(h, fn, activation=act)
    if cfg.get("do_final_norm", True):                                                      # :343-347
        h = O.layer_norm(h, p["node_norm_final.gamma"], p["node_norm_final.beta"])
        e = O.layer_norm(e, p["edge_norm_final.gamma"], p["edge_norm_final.beta"])
    x = masked_global_avg_pool_1d(h, mask)                                                  # zinc/dc.py:109
    x = mlp_out(x, p, len(cfg.get("mlp_layers", [0.5, 0.25])), act)                         # :115
    y = O.dense(x, p["target.kernel"], p["target.bias"])                                    # :116-117
    if return_hidden:
        return y, h, e, mask
    return y


def mae_loss(y_pred, y_true):
    """keras.losses.MeanAbsoluteError (schemes/zinc/svd.py:37-39): mean |y - y'| over the batch."""
    return (y_pred - y_true).abs().mean()
