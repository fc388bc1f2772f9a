"""The ZINC model around the attention path: what the reference's
``lib.models.zinc.dc.DCSVDTransformer`` builds for scheme ``zinc.svd`` with the shipped configs
(``use_svd: false``, ``configs/main/zinc/*/egt.json``), as one torch module

    ZincDCTransformer(**model_config)(node_features, feature_matrix, graph_matrix) -> [B, num_targets]

Pair-sized work runs in the HIP kernels through the C-ABI: the edge-channel input embedding
(egt_edge_embed_fwd/bwd: hop stacking + adj_emb + fm_emb), every attention block (egt_block_*) and
every channel FFN (egt_ffn_*), the node mask producer (egt_node_mask_from_features).  Node-sized
[B,N,Dh] pieces (embedding lookup, final LayerNorm, masked mean pooling, the MLP head, the loss) are
torch ops.  Parameters carry the reference's Keras variable names (keras_named_parameters) so a
weight file of the reference loads unchanged.

Reference (relative to /root/reference/): lib/models/zinc/dc.py:17-120,
lib/models/graph_model_base.py:97-129, lib/models/graph_xformer_model_base.py:336-372,377-466,
lib/training/schemes/zinc/svd.py:27-42, lib/training/schemes/scheme_base.py:37-60.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import nn
import torch.nn.functional as F

from . import _lib as L
from .functional import _f32c, _need_gpu
from .layers import EGTLayerStack, KerasDense, KerasLayerNorm, LN_EPS
from .masks import node_mask_from_features


def _embed_desc(B, N, De, upto_hop, clip_hops, num_edge_features, num_float_features=0, mask_value=-1.0) -> L.EmbedDesc:
    return L.EmbedDesc(B=B, N=N, De=De, upto_hop=upto_hop, clip_hops=1 if clip_hops else 0,
                       num_edge_features=num_edge_features, dtype=L.EGT_F32, num_float_features=num_float_features,
                       mask_value=float(mask_value), reserved=0)


class _EdgeEmbed(torch.autograd.Function):
    @staticmethod
    def forward(ctx, fmat, adj, table, kernel, bias, clip_hops, ffeat=None, mask_value=-1.0):
        """kernel: [upto_hop + F, De] (adj_emb rows, then the rows of the real-valued features' Dense); ffeat [B,N,N,F]"""
        _need_gpu(fmat, adj, table)
        lib = L.load()
        fmat = fmat.to(torch.int32).contiguous()
        adj = _f32c(adj.to(torch.float32))
        table, kernel, bias = _f32c(table), _f32c(kernel), _f32c(bias)
        ffeat = None if ffeat is None else _f32c(ffeat.to(torch.float32))
        B, N, _ = adj.shape
        F_ = 0 if ffeat is None else ffeat.shape[-1]
        K, De = kernel.shape[0] - F_, kernel.shape[1]
        desc = _embed_desc(B, N, De, K, clip_hops, table.shape[0] - 1, F_, mask_value)
        if not lib.egt_edge_embed_supported(C.byref(desc)):
            raise ValueError(f"edge embedding kernel does not cover upto_hop={K}, edge_width={De}, "
                             f"num_edge_features={table.shape[0] - 1}")
        hops = torch.empty(K + F_, B, N, N, dtype=torch.float32, device=adj.device)   # plane-major (unit-stride planes)
        e = torch.empty(B, N, N, De, dtype=torch.float32, device=adj.device)
        L.check(lib.egt_edge_embed_fwd(C.byref(desc), L.ptr(fmat), L.ptr(adj), L.ptr(ffeat), L.ptr(table), L.ptr(kernel),
                                       L.ptr(bias), L.ptr(hops), L.ptr(e), L.current_stream()))
        ctx.desc = desc
        ctx.save_for_backward(fmat, hops, table, kernel, bias)
        ctx.mark_non_differentiable(hops)
        return e, hops

    @staticmethod
    def backward(ctx, de, _dhops):
        lib = L.load()
        fmat, hops, table, kernel, bias = ctx.saved_tensors
        desc = ctx.desc
        de = _f32c(de)
        dt, dk, db = torch.empty_like(table), torch.empty_like(kernel), torch.empty_like(bias)
        ws = torch.empty(lib.egt_edge_embed_workspace_bytes(C.byref(desc)), dtype=torch.uint8, device=de.device)
        L.check(lib.egt_edge_embed_bwd(C.byref(desc), L.ptr(fmat), L.ptr(hops), L.ptr(de), L.ptr(dt), L.ptr(dk),
                                       L.ptr(db), L.ptr(ws), L.current_stream()))
        return None, None, dt, dk, db, None, None, None


def edge_embed(feature_matrix, graph_matrix, fm_table, adj_kernel, adj_bias, clip_hops=True, return_hops=False,
               float_features=None, float_kernel=None, float_bias=None, mask_value=-1.0):
    """e0 = fm_table[feature_matrix + 1] + stack_hops(graph_matrix) @ adj_kernel + adj_bias
          [+ Dense(Masking(float_features))]  ->  [B,N,N,De].
    float_features [B,N,N,F] (F <= 4) with its Dense kernel [F,De] / bias: the real-valued edge features of the
    CIFAR10 / MNIST models (lib/models/cifar10/dc.py:70-73); they ride as F more planes behind the hop planes."""
    kernel, bias = adj_kernel, adj_bias
    if float_features is not None:
        kernel = torch.cat([adj_kernel, float_kernel], dim=0)        # autograd splits the gradient rows back
        bias = adj_bias + float_bias
    e, hops = _EdgeEmbed.apply(feature_matrix, graph_matrix, fm_table, kernel, bias, clip_hops, float_features, mask_value)
    return (e, hops) if return_hops else e


class ZincDCTransformer(nn.Module):
    """DCSVDTransformer for zinc.svd without SVD features; constructor kwargs are the reference's
    model_config keys (scheme_base.py:37-60, zinc/svd.py:27-35) with its defaults."""

    def __init__(self, model_width=64, edge_width=64, num_heads=8, model_height=10, gate_attention=True,
                 edge_channel_type='residual', upto_hop=16, clip_hops=True, random_mask_prob=0.1,
                 clip_logits_value=[-5, 5], mlp_layers=[.5, .25], activation='elu', do_final_norm=True,
                 ffn_multiplier=2., num_node_features=28, num_edge_features=4, num_targets=1,
                 readout_edges=False, num_virtual_nodes=0, node_dropout=0., edge_dropout=0.,
                 # positional encodings (SVDFeatModel / EigFeatModel, graph_model_base.py:284-414)
                 use_svd=False, num_svd_features=256, sel_svd_features=128, random_neg=False, transform_svd=False,
                 use_eig=False, num_eig_features=40, sel_eig_features=20, transform_eig=False,
                 # operator attributes handed to every attention block (graph_xformer_model_base.py:117-127)
                 scale_degree=False, scaler_type='log', attn_dropout=0., edge_activation=None,
                 # reference keys that are accepted at their default only (each would change the model: no silent ignore)
                 l2_reg=0, distance_loss=0., distance_target=8, add_n_norm=False, combine_layer_repr=False,
                 node2edge_xtalk=0., edge2node_xtalk=0., node2edge_embed=False, node_normalization='layer',
                 edge_normalization='layer', global_step_layer=False, max_length=None,
                 seed=0, ffn_matmul='f32', **unknown):
        super().__init__()
        if unknown:
            raise TypeError(f"{type(self).__name__}: unknown model_config keys {sorted(unknown)}")
        unsupported = dict(readout_edges=(readout_edges, False), num_virtual_nodes=(num_virtual_nodes, 0),
                           node_dropout=(node_dropout, 0), edge_dropout=(edge_dropout, 0), l2_reg=(l2_reg, 0),
                           distance_loss=(distance_loss, 0), add_n_norm=(add_n_norm, False),
                           combine_layer_repr=(combine_layer_repr, False), node2edge_xtalk=(node2edge_xtalk, 0),
                           edge2node_xtalk=(edge2node_xtalk, 0), node2edge_embed=(node2edge_embed, False),
                           node_normalization=(node_normalization, 'layer'), edge_normalization=(edge_normalization, 'layer'))
        bad = {k: v for k, (v, d) in unsupported.items() if v != d}
        if bad:    # the reference model applies every one of these: training "a different model without a warning" is not an option
            raise NotImplementedError(f"{type(self).__name__} covers the shipped configs; not built: {bad}")
        if edge_channel_type not in ('residual', 'constrained'):
            raise NotImplementedError("edge_channel_type must be residual or constrained")
        if use_svd and use_eig:
            raise NotImplementedError("use_svd and use_eig together (no reference model class mixes both)")
        self.cfg = dict(model_width=model_width, edge_width=edge_width, num_heads=num_heads, model_height=model_height,
                        upto_hop=upto_hop, clip_hops=clip_hops, mlp_layers=list(mlp_layers), activation=activation,
                        do_final_norm=do_final_norm, num_node_features=num_node_features,
                        num_edge_features=num_edge_features, num_targets=num_targets, ffn_multiplier=ffn_multiplier,
                        edge_channel_type=edge_channel_type,
                        use_svd=bool(use_svd), num_svd_features=num_svd_features, sel_svd_features=sel_svd_features,
                        transform_svd=bool(transform_svd), use_eig=bool(use_eig), num_eig_features=num_eig_features,
                        sel_eig_features=sel_eig_features, transform_eig=bool(transform_eig), random_neg=bool(random_neg))
        self.node_emb = nn.Parameter(torch.empty(num_node_features + 1, model_width).uniform_(-0.05, 0.05))   # keras 'uniform'
        self.fm_emb = nn.Parameter(torch.empty(num_edge_features + 1, edge_width).uniform_(-0.05, 0.05))
        self.adj_emb = KerasDense(upto_hop, edge_width)
        if use_svd and transform_svd:
            self.svd_emb = KerasDense(2 * sel_svd_features, model_width)                # graph_model_base.py:343-345
        if use_eig and transform_eig:
            self.eig_emb = KerasDense(sel_eig_features, model_width)                    # :408-410
        self.layers = EGTLayerStack(model_height=model_height, model_width=model_width, edge_width=edge_width,
                                    activation=activation, num_heads=num_heads, gate_attention=gate_attention,
                                    edge_channel_type=edge_channel_type, clip_logits_value=clip_logits_value,
                                    random_mask_prob=random_mask_prob, scale_degree=scale_degree, scaler_type=scaler_type,
                                    attn_dropout=attn_dropout, edge_activation=edge_activation, seed=seed,
                                    ffn_matmul=ffn_matmul, ffn_multiplier=ffn_multiplier)
        self.node_norm_final = KerasLayerNorm(model_width) if do_final_norm else None
        self.mlp_out = nn.ModuleList()
        w = model_width
        for f in mlp_layers:
            self.mlp_out.append(KerasDense(w, round(f * model_width)))
            w = round(f * model_width)
        self.target = KerasDense(w, num_targets)

    # ---- positional encodings: node_emb_add = Add()([node embedding, PE embedding]) (graph_xformer_model_base.py:390-399) ----
    def positional(self, h, singular_vectors=None, eigen_vectors=None, pe_signs=None):
        """h + the SVD / eigenvector embedding of the batch.  Training applies the reference's random sign flip
        (RandomNeg / RandomNegEig, misc.py:53-94): one sign per (graph, feature), drawn on the device unless `pe_signs`
        ([B,1,F,1] for SVD, [B,1,F] for eigenvectors) injects the sample."""
        c = self.cfg
        if c["use_svd"]:
            if singular_vectors is None:
                raise ValueError("use_svd=True: the batch must carry singular_vectors [B,N,num_svd_features,2]")
            v = singular_vectors.to(h.dtype)[:, :, :c["sel_svd_features"], :]
            if not c["transform_svd"]:
                v = F.pad(v, (0, 0, 0, max(0, c["model_width"] // 2 - c["sel_svd_features"])))
            if c["random_neg"] and self.training:
                sg = pe_signs if pe_signs is not None else \
                    torch.where(torch.rand(v.shape[0], 1, v.shape[2], 1, device=v.device) < 0.5, -1.0, 1.0)
                v = v * sg.to(v.dtype)
            v = torch.cat(torch.unbind(v, dim=-1), dim=-1)
            h = h + (self.svd_emb(v) if c["transform_svd"] else v)
        if c["use_eig"]:
            if eigen_vectors is None:
                raise ValueError("use_eig=True: the batch must carry eigen_vectors [B,N,num_eig_features]")
            v = eigen_vectors.to(h.dtype)[:, :, :c["sel_eig_features"]]
            if not c["transform_eig"]:
                v = F.pad(v, (0, max(0, c["model_width"] - c["sel_eig_features"])))
            if c["random_neg"] and self.training:
                sg = pe_signs if pe_signs is not None else \
                    torch.where(torch.rand(v.shape[0], 1, v.shape[2], device=v.device) < 0.5, -1.0, 1.0)
                v = v * sg.to(v.dtype)
            h = h + (self.eig_emb(v) if c["transform_eig"] else v)
        return h

    def edge_mask(self, graph_matrix, attn_mask):
        """'constrained' edge channels: M = the adjacency tiled over the heads (AdjMatModel.get_edge_mask,
        graph_model_base.py:131-142) unless the caller passes its own."""
        if attn_mask is None and self.cfg["edge_channel_type"] == 'constrained':
            from .masks import constrained_edge_mask
            return constrained_edge_mask(graph_matrix, self.cfg["num_heads"])
        return attn_mask

    # the Keras functional model contains only layers on a path to the outputs: with readout_edges=False the last
    # layer's dense_edge_r / edge FFN and edge_norm_final are NOT part of the reference model
    def _dead_edge_params(self):
        last = self.layers.blocks[-1]
        dead = [last.dense_edge_r.kernel, last.dense_edge_r.bias]
        if self.layers.ffn_edge is not None:
            dead += list(self.layers.ffn_edge[-1].parameters())
        return dead

    def keras_named_parameters(self):
        dead = {id(p) for p in self._dead_edge_params()}
        out = {"adj_emb/kernel": self.adj_emb.kernel, "adj_emb/bias": self.adj_emb.bias}
        if hasattr(self, "svd_emb"):
            out["svd_emb/kernel"], out["svd_emb/bias"] = self.svd_emb.kernel, self.svd_emb.bias
        if hasattr(self, "eig_emb"):
            out["eig_emb/kernel"], out["eig_emb/bias"] = self.eig_emb.kernel, self.eig_emb.bias
        if isinstance(self.node_emb, nn.Parameter):
            out["node_emb/embeddings"] = self.node_emb
        if isinstance(self.fm_emb, nn.Parameter):
            out["fm_emb/embeddings"] = self.fm_emb
        out.update({k: v for k, v in self.layers.keras_named_parameters().items() if id(v) not in dead})
        if self.node_norm_final is not None:
            out["node_norm_final/gamma"] = self.node_norm_final.gamma
            out["node_norm_final/beta"] = self.node_norm_final.beta
        for i, m in enumerate(self.mlp_out):
            out[f"mlp_out_{i}/kernel"], out[f"mlp_out_{i}/bias"] = m.kernel, m.bias
        out["target/kernel"], out["target/bias"] = self.target.kernel, self.target.bias
        return out

    def trainable_parameters(self):
        """the parameters the reference model owns (the dead last-layer edge parameters excluded)"""
        return list(self.keras_named_parameters().values())

    def embeddings(self, node_features, feature_matrix, graph_matrix):
        mask = node_mask_from_features(node_features)                                  # masking.py:42-43
        h = F.embedding((node_features + 1).long(), self.node_emb)                      # zinc/dc.py:66-69
        e = edge_embed(feature_matrix, graph_matrix, self.fm_emb, self.adj_emb.kernel, self.adj_emb.bias,
                       clip_hops=self.cfg["clip_hops"])                                 # :70-73 + graph_model_base.py:97-127
        return h, e, mask

    def forward(self, node_features, feature_matrix, graph_matrix, attn_mask=None, singular_vectors=None,
                eigen_vectors=None, pe_signs=None):
        h, e, mask = self.embeddings(node_features, feature_matrix, graph_matrix)
        h = self.positional(h, singular_vectors, eigen_vectors, pe_signs)
        h, e = self.layers(h, e, mask, self.edge_mask(graph_matrix, attn_mask), skip_last_edge_ffn=True)   # :336-341
        if self.node_norm_final is not None:
            h = self.node_norm_final(h)                                                 # :343-345
        m = mask.to(h.dtype)[..., None]
        x = (h * m).sum(dim=1) / m.sum(dim=1)                                           # node_glob_avg_pool, zinc/dc.py:109
        for lyr in self.mlp_out:                                                        # mlp_out, :354-372
            x = lyr(x)
            x = F.elu(x) if self.cfg["activation"] == 'elu' else torch.relu(x)
        return self.target(x)                                                           # zinc/dc.py:116-117


class PatternDCTransformer(ZincDCTransformer):
    """lib.models.sbm_pattern.dc.DCSVDTransformer for scheme pattern.svd (use_svd false): integer node features
    (3 values), the adjacency hop embedding as the ONLY edge-channel input (no feature matrix), per-node readout
    `mlp_out -> Dense(num_target_labels)` (sbm_pattern/dc.py:51-58).  edge_width 8 in the shipped configs."""

    def __init__(self, num_node_features=3, num_target_labels=2, edge_width=8, model_height=16, **kw):
        kw.pop("num_edge_features", None); kw.pop("num_targets", None)
        super().__init__(num_node_features=num_node_features, num_edge_features=0, num_targets=num_target_labels,
                         edge_width=edge_width, model_height=model_height, **kw)
        # no fm_emb in this model: the embedding kernel gets a one-row ZERO table (a constant buffer, not a parameter)
        del self.fm_emb
        self.register_buffer("fm_emb", torch.zeros(1, edge_width), persistent=False)

    def keras_named_parameters(self):
        out = super().keras_named_parameters()
        out.pop("fm_emb/embeddings", None)
        return out

    def forward(self, node_features, graph_matrix, attn_mask=None, return_mask=False, singular_vectors=None,
                eigen_vectors=None, pe_signs=None):
        fmat = torch.full(graph_matrix.shape, -1, dtype=torch.int32, device=graph_matrix.device)
        h, e, mask = self.embeddings(node_features, fmat, graph_matrix)
        h = self.positional(h, singular_vectors, eigen_vectors, pe_signs)
        h, e = self.layers(h, e, mask, self.edge_mask(graph_matrix, attn_mask), skip_last_edge_ffn=True)
        if self.node_norm_final is not None:
            h = self.node_norm_final(h)
        x = h
        for lyr in self.mlp_out:
            x = lyr(x)
            x = F.elu(x) if self.cfg["activation"] == 'elu' else torch.relu(x)
        y = self.target(x)                                                                # logits [B,N,C]
        return (y, mask) if return_mask else y


class Cifar10DCTransformer(ZincDCTransformer):
    """lib.models.cifar10.dc.DCSVDTransformer for scheme cifar10.svd (use_svd false; the MNIST model has the same
    structure): real-valued node features [B,N,5] and edge features [B,N,N,1], each through keras Masking(mask_value)
    + Dense (cifar10/dc.py:66-73); the adjacency hop embedding is added to the edge embedding; graph-level readout
    (masked mean pool -> mlp_out -> Dense(num_target_labels)).  edge_width 8 / model_height 4 in the shipped config."""

    def __init__(self, num_node_features=5, num_edge_features=1, num_target_labels=10, mask_value=-1., edge_width=8,
                 model_height=4, **kw):
        kw.pop("num_targets", None)
        super().__init__(num_node_features=1, num_edge_features=0, num_targets=num_target_labels,
                         edge_width=edge_width, model_height=model_height, **kw)
        del self.node_emb, self.fm_emb
        self.register_buffer("fm_emb", torch.zeros(1, edge_width), persistent=False)   # no integer feature matrix here
        self.mask_value = float(mask_value)
        self.node_emb = KerasDense(num_node_features, self.cfg["model_width"])
        self.edge_emb = KerasDense(num_edge_features, edge_width)

    def keras_named_parameters(self):
        out = super().keras_named_parameters()
        out.pop("fm_emb/embeddings", None); out.pop("node_emb/embeddings", None)
        out.update({"node_emb/kernel": self.node_emb.kernel, "node_emb/bias": self.node_emb.bias,
                    "edge_emb/kernel": self.edge_emb.kernel, "edge_emb/bias": self.edge_emb.bias})
        return out

    def embeddings(self, node_features, feature_matrix, graph_matrix):
        from .masks import node_mask_from_masking
        mask = node_mask_from_masking(node_features, self.mask_value)                   # keras Masking, cifar10/dc.py:68
        h = self.node_emb(node_features * mask[..., None].to(node_features.dtype))      # Masking zeroes the padded rows
        fmat = torch.full(graph_matrix.shape, -1, dtype=torch.int32, device=graph_matrix.device)
        e = edge_embed(fmat, graph_matrix, self.fm_emb, self.adj_emb.kernel, self.adj_emb.bias,
                       clip_hops=self.cfg["clip_hops"], float_features=feature_matrix, float_kernel=self.edge_emb.kernel,
                       float_bias=self.edge_emb.bias, mask_value=self.mask_value)       # :71-73 + graph_model_base.py:97-127
        return h, e, mask


def sparse_xent_loss(logits, y_true):
    """keras.losses.SparseCategoricalCrossentropy(from_logits=True) (schemes/cifar10/svd.py:37-40): batch mean."""
    return F.cross_entropy(logits, y_true.long())


def class_weights_from_sizes(class_sizes, device=None):
    """WeightedSparseXEntropyLoss (lib/base/genutil/losses.py:41-46)."""
    cs = torch.as_tensor(class_sizes, dtype=torch.float32, device=device)
    w = cs.sum() - cs
    return w / w.sum()


def weighted_sparse_xent_loss(logits, y_true, mask, class_weights):
    """schemes/pattern/svd.py:34-39: class-weighted sparse cross-entropy per node, masked by the node mask, divided by
    the number of (graph, node) slots (Keras SUM_OVER_BATCH_SIZE counts the padded slots too)."""
    logp = torch.log_softmax(logits, dim=-1)
    y = y_true.clamp(min=0).long()
    xent = -logp.gather(-1, y[..., None])[..., 0]
    per = class_weights[y] * xent * mask.to(logits.dtype)
    return per.sum() / per.numel()


def mae_loss(y_pred, y_true):
    """keras.losses.MeanAbsoluteError (schemes/zinc/svd.py:37-39)."""
    return (y_pred - y_true).abs().mean()
