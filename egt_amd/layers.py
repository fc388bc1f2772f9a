"""Host-side mirror of the reference's operator interface for the hot path.

* ``EGT``       — lib/models/egt_layers.py:4-218: same constructor kwargs and
                  defaults, same call convention
                  ``EGT(...)([QKV, E?, G?, M?], mask=None, training=None)
                  -> (V_att, H_hat, A_tild)``, same exceptions.
* ``EGTBlock``  — the closure ``edge_update_{residual,bias,none}(tag, h, e)`` +
                  ``mha_block`` (graph_xformer_model_base.py:106-223) as one
                  module ``(h, e, mask[, attn_mask]) -> (h', e')`` owning the
                  reference's eight named Keras sub-layers' parameters.
* ``EGTStack``  — the ``model_height`` loop (graph_xformer_model_base.py:336-339)
                  over the attention blocks.
All [B,N,N,*] arithmetic runs in the HIP kernels; node-side [B,N,Dh] Dense
layers of the composed path use torch (library GEMMs).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Optional

import torch
from torch import nn
import torch.nn.functional as F

from . import functional as EF

LN_EPS = 1e-3  # keras.layers.LayerNormalization default epsilon


class EGT(nn.Module):
    """Drop-in for lib/models/egt_layers.py:4 ``EGT`` (a parameter-free layer)."""

    def __init__(self,
                 num_heads=8,
                 clip_logits_value=[-5., 5.],
                 scale_degree=False,
                 scaler_type='log',
                 edge_input=True,
                 gate_input=True,
                 attn_mask=False,
                 num_virtual_nodes=0,
                 random_mask_prob=0.0,
                 attn_dropout=0.0,
                 name=None,
                 seed=0,
                 **kwargs):
        super().__init__()
        if scale_degree and not gate_input:                       # egt_layers.py:20-21
            raise ValueError('scale_degree requires gate_input')
        if scaler_type not in ('log', 'linear'):                  # egt_layers.py:23-24
            raise ValueError('scaler_type must be log or linear')
        self.supports_masking = True
        self.num_heads = num_heads
        self.clip_logits_value = clip_logits_value
        self.scale_degree = scale_degree
        self.edge_input = edge_input
        self.gate_input = gate_input
        self.attn_mask = attn_mask
        self.num_virtual_nodes = num_virtual_nodes
        self.random_mask_prob = random_mask_prob
        self.scaler_type = scaler_type
        self.attn_dropout = attn_dropout
        self.name = name
        self.seed = int(seed)
        self._calls = 0
        self.seed_device = None   # (DeviceSeeds, index): egt_amd.graph — the fused path then reads the seed on the device
        self.return_a_tild = True

    def get_config(self):
        # egt_layers.py:42-55 (the reference also omits attn_dropout here)
        return dict(name=self.name,
                    num_heads=self.num_heads,
                    clip_logits_value=self.clip_logits_value,
                    scale_degree=self.scale_degree,
                    edge_input=self.edge_input,
                    gate_input=self.gate_input,
                    attn_mask=self.attn_mask,
                    num_virtual_nodes=self.num_virtual_nodes,
                    random_mask_prob=self.random_mask_prob,
                    scaler_type=self.scaler_type)

    def next_seed(self):
        self._calls += 1
        return (self.seed * 0x9E3779B97F4A7C15 + self._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF

    def forward(self, inputs, mask=None, training=None, rand_mask=None, drop_keep=None):
        if training is None:
            training = self.training                               # egt_layers.py:58-59
        QKV, *inputs = inputs                                      # :62
        E = G = M = None
        if self.edge_input:
            E, *inputs = inputs                                    # :63
        if self.gate_input:
            G, *inputs = inputs                                    # :64
        if self.attn_mask:
            M, *inputs = inputs                                    # :65
        if isinstance(mask, (list, tuple)):
            mask = mask[0]                                         # :66
        assert QKV.shape[2] % (self.num_heads * 3) == 0            # :70
        stochastic = training and (self.random_mask_prob > 0.0 or self.attn_dropout > 0.0)
        if stochastic and self.seed_device is not None and rand_mask is None:
            raise RuntimeError("device-resident mask seeds (egt_amd.graph.DeviceSeeds) are a feature of the fused block / "
                               "stack path; this call runs the composed operator whose seed is a host argument")
        cfg = EF.AttnConfig(num_heads=self.num_heads,
                            clip_logits_value=None if self.clip_logits_value is None
                            else tuple(self.clip_logits_value),
                            scale_degree=self.scale_degree, scaler_type=self.scaler_type,
                            num_virtual_nodes=self.num_virtual_nodes,
                            random_mask_prob=self.random_mask_prob,
                            attn_dropout=self.attn_dropout, training=bool(training),
                            seed=self.next_seed() if stochastic else 0,
                            need_a_tild=self.return_a_tild)
        V_att, H_hat, A_tild = EF.egt_attention(QKV, E, G, M, mask, cfg=cfg,
                                                rand_mask=rand_mask, drop_keep=drop_keep)
        return V_att, H_hat, (A_tild if self.return_a_tild else None)

    def compute_mask(self, inputs, mask=None):                     # egt_layers.py:215-217
        if isinstance(mask, (list, tuple)):
            mask = mask[0]
        return [mask, None, None]


# The reference resolves ``layers.EGT`` through TrackedLayers(custom_layers, ...)
# (graph_xformer_model_base.py:12,80; track_layers/base.py:43-60); this namespace
# is what a maintainer registers first to swap the kernel for every scheme.
custom_layers = SimpleNamespace(EGT=EGT)


class KerasDense(nn.Module):
    """keras.layers.Dense parameters: kernel [in,out] Glorot-uniform, bias zeros."""

    def __init__(self, fan_in, fan_out):
        super().__init__()
        self.kernel = nn.Parameter(torch.empty(fan_in, fan_out))
        self.bias = nn.Parameter(torch.zeros(fan_out))
        nn.init.xavier_uniform_(self.kernel)

    def forward(self, x):
        if x.is_cuda and x.dim() >= 2 and x.dtype == self.kernel.dtype:
            # one library GEMM with the bias in its epilogue (x.W + b, keras.layers.Dense) instead of a GEMM and an add
            return torch.addmm(self.bias, x.reshape(-1, x.shape[-1]), self.kernel).view(*x.shape[:-1], self.kernel.shape[1])
        return x @ self.kernel + self.bias


class KerasLayerNorm(nn.Module):
    """keras.layers.LayerNormalization(axis=-1, epsilon=1e-3) parameters."""

    def __init__(self, width):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(width))
        self.beta = nn.Parameter(torch.zeros(width))

    def forward(self, x):
        return F.layer_norm(x, (x.shape[-1],), self.gamma, self.beta, LN_EPS)


class EGTBlock(nn.Module):
    """(h, e, mask[, attn_mask]) -> (h', e'): graph_xformer_model_base.py:192-223
    ('residual'/'constrained'), :173-190 ('bias'), :164-171 ('none'), each around
    mha_block (:106-145).  Sub-module names are the reference's Keras layer names
    (tag suffix dropped): norm_edge, attention_gates, dense_edge_b, norm_mha,
    dense_qkv, mha, dense_mha, dense_edge_r."""

    def __init__(self, model_width=128, edge_width=32, num_heads=8, gate_attention=True,
                 clip_logits_value=[-5, 5], edge_activation=None,
                 edge_channel_type='residual', scale_degree=False, scaler_type='log',
                 num_virtual_nodes=0, random_mask_prob=0., attn_dropout=0.,
                 node_dropout=0., edge_dropout=0., add_n_norm=False, seed=0, fused='auto'):
        super().__init__()
        if not gate_attention and scale_degree:                   # graph_xformer_model_base.py:47-48
            raise ValueError('scale_degree only works with gate_attention')
        if edge_channel_type not in ('none', 'constrained', 'bias', 'residual'):
            raise KeyError(edge_channel_type)                      # edge_update_fn_dict lookup, :328-334
        self.model_width, self.edge_width, self.num_heads = model_width, edge_width, num_heads
        self.edge_channel_type = edge_channel_type
        self.edge_activation = edge_activation
        self.node_dropout, self.edge_dropout, self.add_n_norm = node_dropout, edge_dropout, add_n_norm
        self.fused = fused
        has_edge = edge_channel_type != 'none'
        # gates exist only on the edge-carrying variants (edge_update_none passes gates=None)
        self.gated = bool(gate_attention) and has_edge
        if edge_channel_type in ('residual', 'constrained'):
            self.norm_edge = KerasLayerNorm(edge_width)
        if self.gated:
            self.attention_gates = KerasDense(edge_width, num_heads)
        if has_edge:
            self.dense_edge_b = KerasDense(edge_width, num_heads)
        self.norm_mha = KerasLayerNorm(model_width)
        self.dense_qkv = KerasDense(model_width, model_width * 3)
        self.mha = EGT(num_heads=num_heads, clip_logits_value=clip_logits_value,
                       scale_degree=scale_degree, scaler_type=scaler_type, edge_input=has_edge,
                       gate_input=self.gated, attn_mask=(edge_channel_type == 'constrained'),
                       num_virtual_nodes=num_virtual_nodes, random_mask_prob=random_mask_prob,
                       attn_dropout=attn_dropout, name='mha', seed=seed)
        self.mha.return_a_tild = False
        self.dense_mha = KerasDense(model_width, model_width)
        if edge_channel_type in ('residual', 'constrained'):
            self.dense_edge_r = KerasDense(num_heads, edge_width)
        if edge_channel_type == 'bias':
            # EGT-simple has no norm_edge / dense_edge_r (:173-190); the fused kernel takes identity LN
            # parameters and a zero update instead (EGT_BF_NO_EDGE_LN).  Constants, not parameters:
            # non-persistent buffers that follow .to(device) -- nothing is allocated per call.
            self.register_buffer('_id_gamma', torch.ones(edge_width), persistent=False)
            self.register_buffer('_id_beta', torch.zeros(edge_width), persistent=False)
            self.register_buffer('_zero_Wr', torch.zeros(num_heads, edge_width), persistent=False)
            self.register_buffer('_zero_br', torch.zeros(edge_width), persistent=False)

    # ---- composed path: HIP edge/attention kernels + torch node-side Dense ----
    @staticmethod
    def _dropout(x, rate, training, keep):
        """keras.layers.Dropout: inverted scaling on the kept elements, training only.  `keep`
        injects the sample (parity tests); otherwise torch's device RNG draws it."""
        if rate <= 0 or not training:
            return x
        if keep is not None:
            return x * (keep.to(x.dtype) / (1.0 - rate))
        return F.dropout(x, rate, True)

    def _mha_block(self, h, e_b, gates, mask, attn_mask, rand_mask, node_keep=None):
        y = h                                                       # :107
        if not self.add_n_norm:
            h = self.norm_mha(h)                                    # :109
        qkv = self.dense_qkv(h)                                     # :113
        inputs = [qkv]
        if self.mha.edge_input:
            inputs.append(e_b)
        if self.mha.gate_input:
            inputs.append(gates)
        if self.mha.attn_mask:
            if attn_mask is None:
                raise ValueError("edge_channel_type='constrained' needs attn_mask")
            inputs.append(attn_mask)
        v_att, h_hat, _ = self.mha(inputs, mask=mask, rand_mask=rand_mask)   # :117-131
        h = self.dense_mha(v_att)                                   # :136
        h = self._dropout(h, self.node_dropout, self.training, node_keep)   # :138-139
        h = h + y                                                   # :140
        if self.add_n_norm:
            h = self.norm_mha(h)                                    # :142-143
        return h, h_hat

    def forward(self, h, e, mask=None, attn_mask=None, rand_mask=None, node_keep=None, edge_keep=None):
        ect = self.edge_channel_type
        if ect == 'none':                                           # :164-171
            h, _ = self._mha_block(h, None, None, mask, attn_mask, rand_mask, node_keep)
            return h, e
        if self._use_fused(h, e, attn_mask, rand_mask):
            from .fused import block_fused
            self.last_path = "fused"
            return block_fused(self, h, e, mask, attn_mask, rand_mask)
        if self._use_pair(h, e, attn_mask, rand_mask, node_keep, edge_keep):
            from .pair import block_pair      # large heads (d = 64): fused pair operator + library GEMMs on the node side
            self.last_path = "fused-pair"
            return block_pair(self, h, e, mask)
        self.last_path = "composed"
        use_ln = ect in ('residual', 'constrained') and not self.add_n_norm
        ne = getattr(self, 'norm_edge', None)
        ag = getattr(self, 'attention_gates', None)
        gates, e_b, e = EF.edge_proj(
            e, ne.gamma if use_ln else None, ne.beta if use_ln else None,
            ag.kernel if ag is not None else None, ag.bias if ag is not None else None,
            self.dense_edge_b.kernel, self.dense_edge_b.bias,
            use_ln=use_ln, edge_activation=self.edge_activation, eps=LN_EPS,
            passthrough=True)                                                  # :195-208
        h, h_hat = self._mha_block(h, e_b, gates, mask, attn_mask, rand_mask, node_keep)  # :212
        if ect == 'bias':
            return h, e                                             # :190 (returns e0)
        if self.edge_dropout > 0 and self.training:
            # dropout sits between dense_edge_r and the residual add (:214-218)
            y = h_hat @ self.dense_edge_r.kernel + self.dense_edge_r.bias
            e = self._dropout(y, self.edge_dropout, True, edge_keep) + e
        else:
            e = EF.edge_update(e, h_hat, self.dense_edge_r.kernel, self.dense_edge_r.bias)
        if self.add_n_norm:
            e = self.norm_edge(e)                                   # :220-221
        return h, e

    def _use_pair(self, h, e, attn_mask, rand_mask, node_keep, edge_keep):
        if self.fused is False or self.fused == 'off' or node_keep is not None or edge_keep is not None:
            return False
        from . import pair as PZ
        return PZ.pair_supported(self, h, e, attn_mask, rand_mask)

    def _use_fused(self, h, e, attn_mask, rand_mask):
        if self.fused is False or self.fused == 'off':
            return False
        try:
            from . import fused as FZ
        except ImportError:
            if self.fused in (True, 'on'):
                raise
            return False
        ok = FZ.block_supported(self, h, e, attn_mask, rand_mask)
        if not ok and self.fused in (True, 'on') and not self._use_pair(h, e, attn_mask, rand_mask, None, None):
            raise RuntimeError("fused EGT block requested but this configuration is not covered by it")
        return ok

    def keras_named_parameters(self, tag: str):
        """Parameters keyed by the reference's Keras variable names for layer
        `tag` ('00', '01', ...), e.g. 'dense_qkv_00/kernel'."""
        out = {}
        for mod_name, mod in self.named_children():
            for p_name, p in mod.named_parameters(recurse=False):
                out[f"{mod_name}_{tag}/{p_name}"] = p
        return out


class EGTStack(nn.Module):
    """model_height attention blocks (graph_xformer_model_base.py:336-339); the
    ffn_block that alternates with them in the reference (:340-341) is outside
    this path."""

    def __init__(self, model_height=4, **block_kwargs):
        super().__init__()
        seed = block_kwargs.pop('seed', 0)
        self.stack_call = block_kwargs.pop('stack_call', True)
        self.grad_holder = SimpleNamespace(flat=None, sink=None)   # flat gradient buffer of the last fused backward; sink: bound buffer
        self._stack_ok = {}
        self.blocks = nn.ModuleList(
            [EGTBlock(seed=seed * 1000 + i, **block_kwargs) for i in range(model_height)])

    def fused_parameters(self):
        """Parameters in the order the fused stack lays their gradients out in
        grad_holder.flat (layer-major, the C-ABI egt_block_params order)."""
        from .fused import _GRAD_ORDER
        out = []
        for blk in self.blocks:
            for mod, attr in _GRAD_ORDER:
                m = getattr(blk, mod, None)
                if m is not None:
                    out.append(getattr(m, attr))
        return out

    def bind_flat_gradients(self):
        """One persistent flat gradient buffer for the whole stack: every parameter's .grad becomes (and stays) its view of
        it, the fused stack backward writes into it directly, and the data-parallel collective runs on it
        (grad_holder.flat).  For training loops that never reset .grad to None (the backward OVERWRITES: each parameter takes
        part in one stack call per step); undo with unbind_flat_gradients()."""
        ps = self.fused_parameters()
        flat = torch.zeros(sum(p.numel() for p in ps), dtype=torch.float32, device=ps[0].device)
        off = 0
        for p in ps:
            p.grad = flat[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.grad_holder.sink = self.grad_holder.flat = flat
        self.grad_holder.sink_params = ps          # the backward re-attaches the views if a zero_grad(set_to_none=True) dropped them
        return flat

    def unbind_flat_gradients(self):
        self.grad_holder.sink = None
        self.grad_holder.sink_params = None
        for p in self.fused_parameters():
            p.grad = None

    def forward(self, h, e, mask=None, attn_mask=None):
        if self.stack_call and h.is_cuda:
            from . import fused as FZ
            # the decision depends on construction-time attributes and on this key only: made once per geometry
            key = (tuple(h.shape), tuple(e.shape), h.dtype, e.dtype, h.device, attn_mask is None,
                   tuple(b.training for b in self.blocks))
            ok = self._stack_ok.get(key)
            if ok is None:
                ok = self._stack_ok[key] = bool(FZ.stack_supported(self, h, e, attn_mask))
            if ok:
                self.last_path = "fused-stack"
                return FZ.stack_fused(self, h, e, mask, attn_mask)   # one C-ABI call per direction
        if self.grad_holder.sink is not None:
            raise RuntimeError("bind_flat_gradients() needs the fused stack path (the per-block path accumulates through autograd); "
                               "call unbind_flat_gradients() for this geometry")
        self.last_path = "per-block"
        for blk in self.blocks:
            h, e = blk(h, e, mask, attn_mask)
        return h, e


class EGTLayerStack(nn.Module):
    """The reference's full layer loop (graph_xformer_model_base.py:336-341):
        for ii in range(model_height):  h, e = edge_update(tag, h, e);  h, e = ffn_block(tag, h, e)
    with the attention block on the fused path (EGTBlock) and one fused FFN per channel type
    (egt_amd.ffn.FFN; widths 64).  `edge_channel_type` in ('residual', 'constrained') updates
    both channels in ffn_block (:312-320); otherwise only the node channels (:322-323)."""

    def __init__(self, model_height=4, model_width=64, edge_width=64, activation='elu', ffn_matmul='f32', ffn_multiplier=2.0,
                 **block_kwargs):
        super().__init__()
        from .ffn import FFN as _FFN
        from functools import partial
        FFN = partial(_FFN, matmul=ffn_matmul, ffn_multiplier=ffn_multiplier)   # (the FFN refuses a multiplier it is not built for)
        # "f32" exact | "bf16x3" split products (fp32 tolerances) | "bf16"
        seed = block_kwargs.pop('seed', 0)
        self.blocks = nn.ModuleList(
            [EGTBlock(seed=seed * 1000 + i, model_width=model_width, edge_width=edge_width, **block_kwargs)
             for i in range(model_height)])
        ect = block_kwargs.get('edge_channel_type', 'residual')
        # FFN.__init__ raises ValueError for a width the fused FFN kernels do not cover (at
        # construction, not at the first forward: ADVICE r1)
        self.ffn_node = nn.ModuleList([FFN(model_width, activation=activation) for _ in range(model_height)])
        self.ffn_edge = nn.ModuleList([FFN(edge_width, activation=activation) for _ in range(model_height)]) \
            if ect in ('residual', 'constrained') else None
        self.overlap_ffn = False     # node FFN on a side stream beside the edge FFN (small per-GPU batches; see forward)
        self._side = None

    def keras_named_parameters(self):
        """every parameter under the reference's Keras variable name: attention sub-layers
        '<layer>_<ii>/<var>' (graph_xformer_model_base.py:337 tags), FFN sub-layers
        'norm_fnn_node_<ii>/gamma', 'fnn_lr1_edge_<ii>/kernel', ... (:310,314)."""
        out = {}
        for i, blk in enumerate(self.blocks):
            tag = f"{i:0>2d}"
            out.update(blk.keras_named_parameters(tag))
            out.update(self.ffn_node[i].keras_named_parameters(f"node_{tag}"))
            if self.ffn_edge is not None:
                out.update(self.ffn_edge[i].keras_named_parameters(f"edge_{tag}"))
        return out

    @torch.no_grad()
    def load_keras_weights(self, weights, strict=True):
        """weights: mapping Keras variable name -> array (e.g. read from the reference's .h5 / an .npz
        export).  Shapes are Keras' ([in,out] Dense kernels), so arrays are copied as they are."""
        named = self.keras_named_parameters()
        missing = [k for k in named if k not in weights]
        unexpected = [k for k in weights if k not in named]
        if strict and (missing or unexpected):
            raise KeyError(f"missing {missing[:4]}... unexpected {unexpected[:4]}...")
        for k, prm in named.items():
            if k in weights:
                w = torch.as_tensor(weights[k], dtype=prm.dtype)
                if tuple(w.shape) != tuple(prm.shape):
                    raise ValueError(f"{k}: shape {tuple(w.shape)} vs {tuple(prm.shape)}")
                prm.copy_(w.to(prm.device))
        return missing, unexpected

    def forward(self, h, e, mask=None, attn_mask=None, skip_last_edge_ffn=False):
        """skip_last_edge_ffn: a model whose readout ignores the edge channels (e.g. ZINC, readout_edges=False)
        does not contain the last layer's edge FFN in the reference either (a Keras functional model keeps only
        layers on a path to its outputs)."""
        last = len(self.blocks) - 1
        for i, blk in enumerate(self.blocks):
            h, e = blk(h, e, mask, attn_mask)                  # layer/{i}/attention
            edge_ffn = self.ffn_edge is not None and not (skip_last_edge_ffn and i == last)
            if self.overlap_ffn and edge_ffn and h.is_cuda:
                # the node and the edge FFN of a layer are independent (graph_xformer_model_base.py:309-324): the node FFN
                # (a few hundred workgroups at most) runs on a side stream beside the edge FFN; autograd runs its backward on
                # that stream too and orders the streams; inside a hipGraph capture this is a fork / join of the graph
                cur = torch.cuda.current_stream()
                if self._side is None:
                    self._side = torch.cuda.Stream()
                side = self._side
                side.wait_stream(cur)
                h.record_stream(side)
                with torch.cuda.stream(side):
                    h2 = self.ffn_node[i](h)
                e = self.ffn_edge[i](e)                         # layer/{i}/ffn
                cur.wait_stream(side)
                h2.record_stream(cur)
                h = h2
                continue
            if edge_ffn:                                        # layer/{i}/ffn
                e = self.ffn_edge[i](e)
            h = self.ffn_node[i](h)
        return h, e
