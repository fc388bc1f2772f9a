"""Scheme / config driver for the ZINC model: consumes the reference's
``configs/**/zinc/*.json`` unchanged — same keys, same defaults, unknown key -> KeyError — and runs the
same training protocol around egt_amd.model.ZincDCTransformer:

    python -m egt_amd.training cfg.json            (the reference: python run_training.py cfg.json)

Reference (relative to /root/reference/):
  config keys / lazily evaluated defaults   lib/training/training_base.py:80-112, lib/base/dotdict/dotdict.py:23-105
  scheme defaults and config -> model_config lib/training/schemes/scheme_base.py:6-60,92-111,116-160,
                                            lib/training/schemes/zinc/svd.py:12-42
  optimizer (Adam / RMSprop / SGD, clipvalue) lib/training/training_base.py:59-72
  state, save-best, reduce-LR-on-plateau, min_lr_factor, stopping_lr   :114-181
  per-epoch checkpoint + restore at train begin, max_to_keep=1         lib/base/callbacks/checkpoint.py:8-83
  ``save_when`` mini-DSL (event;cond;format) weight snapshots           :86-138
  warm-up + cosine schedule                   lib/base/genutil/warmup.py:41-75
  fit loop, finalize (final weight file)      lib/training/training_base.py:293-327
  data-parallel training                      :230-247 (MirroredStrategy) -> egt_amd.dp, one process per GPU

Data: ``load_data`` opens ``config.dataset_path`` through egt_amd.data (a PackedStore ``.npz``, or the
reference's ``.h5`` where h5py exists) or takes any iterable of batches in the reference's input format
(``SyntheticZinc`` / ``SyntheticPattern`` / ``SyntheticCifar10`` generate such batches; the datasets themselves are
not in this image).  Weight files are ``.npz`` keyed by the reference's Keras variable names instead of ``.h5``.
"""
from __future__ import annotations

import json
import math
import os
import sys
from typing import Callable, Dict, Iterable, Optional

import numpy as np
import torch

# ----------------------------------------------------------------------------------------- config --


class Config(dict):
    """HDict (dotdict.py:87-105): attribute access; a value that is a callable ``f(c)`` is a lazily evaluated
    default (the reference's ``HDict.L('c: expr')`` string lambdas) resolved against the final config."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        return v(self) if callable(v) else v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def get_dict(self):
        return {k: self[k] for k in self.keys()}


def read_config_from_file(config_file):          # training_base.py:15-17
    with open(config_file, "r") as fp:
        return json.load(fp)


def save_config_to_file(config, config_file):    # :19-21
    with open(config_file, "w") as fp:
        return json.dump(config, fp, indent="\t")


def default_config(scheme: str = "zinc.svd") -> Config:
    """TrainingBase.get_default_config (:80-112) + BaseDCModelScheme (scheme_base.py:7-35) + BaseAdjModelScheme
    (:93-101) + BaseSVDModelScheme (:117-126) + ZincDCSVD (zinc/svd.py:13-21), later ones overriding."""
    path = os.path
    c = Config(
        scheme=None, model_name="unnamed_model", distributed=False,
        batch_size=lambda c: 32 if c.distributed else 128,
        initial_lr=5e-4, gradient_clipval=None, num_epochs=1000,
        dataset_path="datasets/gnn_benchmark.h5",
        save_path=lambda c: path.join("models", c.model_name),
        checkpoint_path=lambda c: path.join(c.save_path, "checkpoint"),
        log_path=lambda c: path.join(c.save_path, "logs"),
        config_path=lambda c: path.join(c.save_path, "config"),
        summary_path=lambda c: path.join(c.save_path, "summary"),
        saved_model_path=lambda c: path.join(c.save_path, "saved", c.model_name),
        rlr_factor=0.5, rlr_patience=10, rlr_monitor=lambda c: c.save_best_monitor,
        min_lr_factor=0.01, stopping_lr=0., steps_per_epoch=None, validation_steps=None,
        save_best=True,
        save_when=lambda c: "" if not c.save_best else "epoch;" + c.save_best_monitor + "<=save_best_value;epoch{epoch:0>4d}",
        save_best_monitor="val_loss", stopping_patience=0,
        predictions_path=lambda c: path.join(c.save_path, "predictions"),
        weight_file=":", prediction_bmult=2, optimizer="adam",
        use_hipgraph=False,      # (not a reference key) forward + loss + backward of a training batch replayed from a hipGraph
                                 # captured per batch geometry: egt_amd.graph; for launch-bound per-GPU batches
    )
    c.update(  # BaseDCModelScheme
        model_name="dc", dataset_name="dataset",
        dataset_path=lambda c: f"datasets/{c.dataset_name.upper()}/{c.dataset_name.upper()}.h5",
        cache_dir=lambda c: f"data_cache/{c.dataset_name.upper()}/data",
        save_path=lambda c: path.join(f"models/{c.dataset_name.lower()}", c.model_name),
        model_width=48, model_height=4, edge_width=48, num_heads=8, gate_attention=True, scale_degree=False,
        l2_reg=0, dropout=0, attn_dropout=0.0, edge_dropout=None, mlp_layers=[.5, .25], edge_activation=None,
        edge_channel_type="residual", combine_layer_repr=False, max_shuffle_len=10000, ffn_multiplier=2.,
        warmup_steps=0, total_steps=None, random_mask_prob=0.,
    )
    c.update(  # BaseAdjModelScheme
        model_name="dc_mat", cache_dir=lambda c: f"data_cache/{c.dataset_name.upper()}/mat",
        upto_hop=1, distance_loss=0., distance_target=8,
    )
    c.update(  # BaseSVDModelScheme
        model_name="dc_svd", cache_dir=lambda c: f"data_cache/{c.dataset_name.upper()}/svd_{c.num_svd_features}",
        num_svd_features=16, sel_svd_features=8, use_svd=True, random_neg=True,
    )
    if scheme.endswith(".eig"):   # BaseEigModelScheme (scheme_base.py:155-165) instead of BaseSVDModelScheme
        for k in ("num_svd_features", "sel_svd_features", "use_svd", "random_neg"):
            c.pop(k)
        c.update(model_name="dc_eig", cache_dir=lambda c: f"data_cache/{c.dataset_name.upper()}/eig_{c.num_eig_features}",
                 num_eig_features=20, sel_eig_features=8, use_eig=True)
    if scheme == "cifar10.svd":   # CIFAR10DCSVD (schemes/cifar10/svd.py:13-20): rlr_monitor follows save_best_monitor
        c.update(dataset_name="cifar10", num_virtual_nodes=0, save_best_monitor="val_xent")
    elif scheme == "pattern.svd":   # SBMPDCSVD (schemes/pattern/svd.py:16-24)
        c.update(dataset_name="sbm_pattern", class_sizes=[979220, 209900], rlr_monitor="val_xent", save_best_monitor="val_xent")
    elif scheme == "pattern.eig":   # SBMPDCEig (schemes/pattern/eig.py:16-22): the monitors stay at the base default (val_loss)
        c.update(dataset_name="sbm_pattern", class_sizes=[979220, 209900])
    else:                         # ZincDCSVD (schemes/zinc/svd.py:13-21), ZincDCEig (schemes/zinc/eig.py:14-22)
        c.update(dataset_name="zinc", num_virtual_nodes=0, rlr_monitor="val_mae", save_best_monitor="val_mae")
    return c


SCHEMES = ("zinc.svd", "zinc.eig", "pattern.svd", "pattern.eig", "cifar10.svd")


def make_config(user: Optional[dict], scheme: Optional[str] = None) -> Config:
    """TrainingBase.__init__ (:24-32): defaults, then the user's keys; an unknown key is an error."""
    c = default_config(scheme or (user or {}).get("scheme") or "zinc.svd")
    if user is not None:
        for k in user.keys():
            if k not in c:
                raise KeyError(f'Unknown config "{k}"')
        c.update(user)
    return c


def model_config(c: Config) -> dict:
    """config -> model_config (scheme_base.py:37-60, :103-111, :151-160; zinc/svd.py:27-35; pattern/svd.py:30-32)."""
    mc = _model_config_common(c)
    if "num_virtual_nodes" in c:           # the ZINC scheme adds these two (zinc/svd.py:30-34)
        mc.update(readout_edges=False, num_virtual_nodes=c.num_virtual_nodes)
    return mc


def _model_config_common(c: Config) -> dict:
    return dict(
        model_width=c.model_width, edge_width=c.edge_width, num_heads=c.num_heads, gate_attention=c.gate_attention,
        scale_degree=c.scale_degree, random_mask_prob=c.random_mask_prob, attn_dropout=c.attn_dropout,
        model_height=c.model_height, l2_reg=c.l2_reg, node_dropout=c.dropout,
        edge_dropout=c.dropout if c.edge_dropout is None else c.edge_dropout,
        mlp_layers=c.mlp_layers, edge_channel_type=c.edge_channel_type, edge_activation=c.edge_activation,
        ffn_multiplier=c.ffn_multiplier, global_step_layer=True,
        upto_hop=c.upto_hop, distance_loss=c.distance_loss, distance_target=c.distance_target,
        **(dict(use_eig=c.use_eig, transform_eig=False, random_neg=True, num_eig_features=c.num_eig_features,
                sel_eig_features=c.sel_eig_features)                                    # BaseEigModelScheme, scheme_base.py:178-190
           if "use_eig" in c else
           dict(use_svd=c.use_svd, transform_svd=True, random_neg=c.random_neg, num_svd_features=c.num_svd_features,
                sel_svd_features=c.sel_svd_features)),                                  # BaseSVDModelScheme, :139-151
    )


# ------------------------------------------------------------------------------------------ state --
class TrainingState:
    """get_default_state (:114-131): counters + save-best + reduce-LR bookkeeping (checkpointed)."""

    def __init__(self, config: Config):
        self.current_epoch = 0
        self.global_step = 0
        self.has_best = bool(config.save_best)
        self.has_rlr = config.rlr_factor < 1.0
        self.save_best_value = math.inf
        self.save_best_epoch = 0
        self.last_reduce_lr = 0

    def items(self):
        d = dict(current_epoch=self.current_epoch, global_step=self.global_step)
        if self.has_best:
            d.update(save_best_value=self.save_best_value, save_best_epoch=self.save_best_epoch)
        if self.has_rlr:
            d.update(last_reduce_lr=self.last_reduce_lr)
        return d

    def load(self, d):
        for k, v in d.items():
            setattr(self, k, v)


def save_best_update(config: Config, state: TrainingState, get_lr: Callable[[], float], set_lr: Callable[[float], None],
                     logs: dict, print_fn=print) -> bool:
    """get_state_updates.save_best_update (:145-178), called at epoch end BEFORE current_epoch is advanced...
    the reference appends it after the counter update, so it sees the advanced epoch.  Returns stop_training."""
    monitor = config.save_best_monitor
    new_value = logs.get(monitor, math.inf)
    old_value, old_epoch, new_epoch = state.save_best_value, state.save_best_epoch, state.current_epoch
    if new_value < old_value:
        state.save_best_value, state.save_best_epoch = float(new_value), new_epoch
        print_fn(f"\nSAVE BEST: {monitor} improved from (epoch:{old_epoch},value:{old_value:0.5f}) to (epoch:{new_epoch},value:{new_value:0.5f})")
    else:
        print_fn(f"\nSAVE BEST: {monitor} did NOT improve from (epoch:{old_epoch},value:{old_value:0.5f})")
        if config.rlr_factor < 1.0:
            epoch_gap = new_epoch - max(old_epoch, state.last_reduce_lr)
            if epoch_gap >= config.rlr_patience:
                set_lr(max(get_lr() * config.rlr_factor, config.initial_lr * config.min_lr_factor))
                state.last_reduce_lr = new_epoch
                print_fn(f"\nRLR: {monitor} did NOT improve for {epoch_gap} epochs, new lr = {get_lr()}")
    if get_lr() < config.stopping_lr:
        print_fn(f"\nSTOP: lr fell below {config.stopping_lr}, STOPPING TRAINING!")
        return True
    return False


class SaveWhen:
    """SaveWhenCallback (checkpoint.py:86-138): '#'-separated 'event;condition;name-format' rules evaluated against
    the epoch logs + state; a rule that fires snapshots the weights."""

    def __init__(self, when: str):
        self.criterions = []
        for item in (when.split("#") if when else []):
            ev, cond, fmt = (e.strip() for e in item.split(";"))
            self.criterions.append((ev.lower(), cond, fmt))

    def fire(self, event: str, scope: dict):
        out = []
        for e, c, f in self.criterions:
            if e == event:
                try:
                    if eval(c, {"__builtins__": {}}, dict(scope)):   # noqa: S307 - the reference evals the config string
                        out.append(f.format(**scope))
                except NameError:
                    pass                                           # ':125-126: did not find log, IGNORING'
        return out


def warmup_cosine_lr(global_step: int, warmup_steps: int, max_lr: float, total_steps: Optional[int], min_lr: float = 0.):
    """WarmUpAndCosine.on_train_batch_begin (warmup.py:56-67): (lr or None = leave unchanged, stop_training)."""
    span = max_lr - min_lr
    if global_step < warmup_steps:
        return min_lr + span / warmup_steps * (global_step + 1), False
    if total_steps is not None:
        if global_step <= total_steps:
            return min_lr + span * math.cos(0.5 * math.pi / (total_steps - warmup_steps) * (global_step - warmup_steps)), False
        return None, True
    return None, False


# --------------------------------------------------------------------------------------------- data --
class SyntheticZinc:
    """Batches in the reference's input format (lib/data/datasets/zinc.py:9-142 after MatrixDataset + padded_batch):
    node_features [B,N] int (28 atom types, padding -1), feature_matrix [B,N,N] int (bond type, -1 elsewhere),
    graph_matrix [B,N,N] 0/1, target [B,1]; N = the batch's longest molecule (padded_batch pads to the batch max,
    lib/data/dataset_base.py:106-111), optionally rounded up to a multiple of `pad_multiple`."""

    def __init__(self, n_graphs=1024, batch_size=128, nodes=(9, 37), seed=0, pad_multiple=1, device="cpu"):
        g = torch.Generator().manual_seed(seed)
        self.n = torch.randint(nodes[0], nodes[1] + 1, (n_graphs,), generator=g)
        self.seed, self.batch_size, self.pad_multiple, self.device = seed, batch_size, pad_multiple, device
        # a learnable target: a fixed random function of the atom-type histogram and the bond count
        self.w = torch.randn(28, generator=g) * 0.3

    def __len__(self):
        return (len(self.n) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for b in range(len(self)):
            ns = self.n[b * self.batch_size:(b + 1) * self.batch_size]
            g = torch.Generator().manual_seed(self.seed * 100003 + b)
            B, N = len(ns), int(ns.max())
            N = (N + self.pad_multiple - 1) // self.pad_multiple * self.pad_multiple
            real = torch.arange(N)[None, :] < ns[:, None]
            nf = torch.randint(0, 28, (B, N), generator=g)
            nf[~real] = -1
            pr = (1.1 / ns.float().clamp(min=2))[:, None, None]
            adj = (torch.rand(B, N, N, generator=g) < pr).float()
            adj = ((adj + adj.transpose(1, 2)) > 0).float() * (real[:, :, None] & real[:, None, :]).float()
            adj = adj * (1 - torch.eye(N))[None]
            bond = torch.randint(0, 4, (B, N, N), generator=g)
            bond = torch.triu(bond, 1); bond = bond + bond.transpose(1, 2)
            fm = torch.where(adj > 0, bond, torch.tensor(-1))
            hist = torch.stack([(nf == t).sum(1) for t in range(28)], 1).float()
            tgt = (hist @ self.w / 10 + adj.sum((1, 2)) / 40)[:, None]
            yield dict(node_features=nf.int().to(self.device), feature_matrix=fm.int().to(self.device),
                       graph_matrix=adj.to(self.device), target=tgt.to(self.device))


class WithPositional:
    """Adds the positional-encoding input of a `*.svd` (use_svd) / `*.eig` scheme to the batches of a synthetic set, computed
    from each graph's adjacency with the data pipeline's own functions (egt_amd.data.svd_features / eigen_features):
    singular_vectors [B,N,F,2] or eigen_vectors [B,N,F], zero rows on the padding."""

    def __init__(self, base, kind: str, num_features: int):
        assert kind in ("svd", "eig")
        self.base, self.kind, self.F = base, kind, int(num_features)

    def __len__(self):
        return len(self.base)

    def __iter__(self):
        from .data import svd_features, eigen_features
        for b in self.base:
            adj = b["graph_matrix"].cpu().numpy()
            nf = b["node_features"].cpu().numpy()
            real = (nf != -1) if nf.ndim == 2 else (nf != -1).any(-1)
            B, N = adj.shape[:2]
            out = np.zeros((B, N, self.F, 2) if self.kind == "svd" else (B, N, self.F), dtype=np.float32)
            for i in range(B):
                n = int(real[i].sum())
                if n == 0:
                    continue
                A = adj[i, :n, :n]
                if self.kind == "svd":
                    out[i, :n] = svd_features(A, self.F)[0]
                else:
                    ed = np.argwhere(A > 0)
                    ev = eigen_features(ed, n, min(self.F, max(n - 1, 0)))
                    out[i, :n, :ev.shape[1]] = ev
            key = "singular_vectors" if self.kind == "svd" else "eigen_vectors"
            yield dict(b, **{key: torch.from_numpy(out).to(b["graph_matrix"].device)})


def _dist_on():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def _rank():
    return torch.distributed.get_rank() if _dist_on() else 0


def _barrier():
    if _dist_on():
        torch.distributed.barrier()


# ------------------------------------------------------------------------------------------ scheme --
class ZincSVDScheme:
    """lib.training.schemes.zinc.svd.SCHEME: TrainingBase protocol around the ZINC model."""
    SCHEME = "zinc.svd"
    MAX_GRAPHS = 16        # config.use_hipgraph: captured step graphs kept alive (least recently used geometry is dropped)

    def __init__(self, config: Optional[dict] = None, model_factory=None, device=None, print_fn=print):
        self.config_input = config
        self.config = make_config(config, self.SCHEME)
        if self.config.scheme not in (None, self.SCHEME):
            raise KeyError(f"scheme {self.config.scheme!r} given to {type(self).__name__} ({self.SCHEME}); built schemes: {SCHEMES}")
        self.state = TrainingState(self.config)
        self.model_factory = model_factory
        self.device = device
        self.print = print_fn
        self.stop_training = False
        self.history = []

    # ---- model / optimizer (:59-72, :227-247) ----
    def get_model_config(self):
        return model_config(self.config)

    def get_model(self):
        if self.model_factory is not None:
            return self.model_factory(self.get_model_config())
        from .model import ZincDCTransformer
        return ZincDCTransformer(**self.get_model_config())

    def get_optimizer(self, params):
        c = self.config
        opt = dict(adam=lambda p: torch.optim.Adam(p, lr=c.initial_lr, betas=(0.9, 0.999), eps=1e-7),     # Keras Adam epsilon
                   rmsprop=lambda p: torch.optim.RMSprop(p, lr=c.initial_lr, alpha=0.9, eps=1e-7),        # Keras rho=0.9
                   sgd=lambda p: torch.optim.SGD(p, lr=c.initial_lr))[c.optimizer]
        return opt(params)

    def get_loss(self):                              # zinc/svd.py:37-39
        from .model import mae_loss
        return mae_loss

    def get_metrics(self):                           # :41-42
        return ["mae"]

    def load_model(self):
        self.model = self.get_model()
        if self.device is not None:
            self.model = self.model.to(self.device)
        params = self.model.trainable_parameters() if hasattr(self.model, "trainable_parameters") else list(self.model.parameters())
        self.params = params
        self.optimizer = self.get_optimizer(params)
        self.loss_fn = self.get_loss()
        self.flat = None
        self._graphs, self._seeds = {}, None
        self._use_graph = bool(self.config.use_hipgraph)
        if self._use_graph and not (self.device is not None and torch.device(self.device).type == "cuda"):
            raise ValueError("config.use_hipgraph needs the model on a GPU (device='cuda')")
        if self._use_graph or (self.config.distributed and torch.distributed.is_available() and torch.distributed.is_initialized()):
            from .dp import FlatGradAllReduce
            self.flat = FlatGradAllReduce(params, direct=True)     # one flat-buffer all-reduce per step (MirroredStrategy, :230-247)
        if self._use_graph:                           # every captured geometry writes its gradients into this one flat buffer
            from .graph import DeviceSeeds
            from .layers import EGT
            if any(isinstance(m, EGT) for m in self.model.modules()):
                self._seeds = DeviceSeeds.attach(self.model, self.device)

    # ---- lr ----
    def get_lr(self):
        return self.optimizer.param_groups[0]["lr"]

    def set_lr(self, v):
        for g in self.optimizer.param_groups:
            g["lr"] = float(v)

    # ---- checkpoint (checkpoint.py:8-83: model + optimizer + state, max_to_keep = 1, restored at train begin) ----
    def _ckpt_file(self):
        return os.path.join(self.config.checkpoint_path, "ckpt.pt")

    def save_checkpoint(self):
        """rank 0 writes (the replicas are identical); everybody waits for the file"""
        if _rank() == 0:
            os.makedirs(self.config.checkpoint_path, exist_ok=True)
            tmp = self._ckpt_file() + ".tmp"
            extra = {}
            if getattr(self, "_seeds", None) is not None:
                extra["mask_seeds"] = self._seeds.state_dict()       # device-resident mask streams (use_hipgraph)
            extra["mask_calls"] = [int(m._calls) for m in self.model.modules() if hasattr(m, "_calls") and hasattr(m, "next_seed")]
            torch.save(dict(model=self.model.state_dict(), optimizer=self.optimizer.state_dict(), state=self.state.items(), **extra), tmp)
            os.replace(tmp, self._ckpt_file())
            self.print(f"Checkpoint saved to {self.config.checkpoint_path}")
        _barrier()

    def load_checkpoint(self):
        f = self._ckpt_file()
        if not os.path.exists(f):
            return False
        ck = torch.load(f, map_location=self.device or "cpu")
        self.model.load_state_dict(ck["model"])
        self.optimizer.load_state_dict(ck["optimizer"])
        self.state.load(ck["state"])
        mods = [m for m in self.model.modules() if hasattr(m, "_calls") and hasattr(m, "next_seed")]
        if len(ck.get("mask_calls", [])) == len(mods):        # the random-mask streams continue where the run stopped
            for m, n in zip(mods, ck["mask_calls"]):
                m._calls = n
        if getattr(self, "_seeds", None) is not None and "mask_seeds" in ck:
            self._seeds.load_state_dict(ck["mask_seeds"])
        self.print(f"Checkpoint loaded from {self.config.checkpoint_path}")
        return True

    def load_state(self):                            # :249-260
        os.makedirs(self.config.checkpoint_path, exist_ok=True)
        self.save_when = SaveWhen(self.config.save_when)
        self.load_checkpoint()

    # ---- weight files (Keras variable names; .npz instead of .h5) ----
    def _named(self):
        return self.model.keras_named_parameters() if hasattr(self.model, "keras_named_parameters") else dict(self.model.named_parameters())

    def save_weights(self, path):
        if _rank() == 0:
            os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
            np.savez(path, **{k: v.detach().cpu().numpy() for k, v in self._named().items()})
            self.print(f"Saved model to {path}")
        _barrier()

    @torch.no_grad()
    def load_weights(self, path):
        """model.load_weights(file, by_name=True) (:362): every variable the file names is copied; the others stay"""
        missing = []
        with np.load(path) as z:
            for k, v in self._named().items():
                if k in z.files:
                    v.copy_(torch.from_numpy(z[k]).to(v.device))
                else:
                    missing.append(k)
        if missing:   # by_name loading is silent about these in Keras; a run that continues from a partial file should say so
            self.print(f"load_weights: {len(missing)} variable(s) not in {path} keep their current values: " + ", ".join(missing[:8]) +
                       (" ..." if len(missing) > 8 else ""))
        return missing

    def config_summary(self):
        for k, v in self.config.get_dict().items():
            self.print(f"{k} : {v}")

    def save_config_file(self):                      # :187-190
        if _rank() == 0:
            os.makedirs(os.path.dirname(self.config.config_path), exist_ok=True)
            save_config_to_file(self.config.get_dict(), self.config.config_path + ".json")
            save_config_to_file(self.config_input, self.config.config_path + "_input.json")
        _barrier()

    # ---- data ----
    def load_data(self, trainset: Iterable = None, valset: Iterable = None, testset: Iterable = None,
                  splits=("training", "validation")):
        """:207-218.  Iterables of batches in the reference's input format are used as they are (SyntheticZinc ...);
        otherwise the scheme's dataset is opened from ``config.dataset_path`` through egt_amd.data (a PackedStore
        ``.npz`` or, where h5py exists, the reference's ``.h5``): record maps, excluded features, per-epoch shuffle,
        padded batches of ``batch_size``.  Under DP every rank takes its contiguous slice of each GLOBAL batch: the
        shuffle seed is rank 0's (broadcast), so the slices partition the batch; the validation / test splits are sharded
        the same way and their metric sums are all-reduced in ``evaluate`` (every rank sees the same logs, hence the same
        save-best / reduce-LR / stop decisions)."""
        if trainset is None:
            from .data import dataset_for_scheme
            c = self.config
            if not os.path.exists(c.dataset_path):
                raise FileNotFoundError(f"dataset_path {c.dataset_path!r} does not exist (pass iterables of batches, "
                                        "or a PackedStore .npz / reference .h5)")
            shard, seed = None, int(np.random.SeedSequence().entropy % (2 ** 62))
            if c.distributed and _dist_on():
                shard = (torch.distributed.get_rank(), torch.distributed.get_world_size())
                box = [seed]
                torch.distributed.broadcast_object_list(box, src=0)     # every rank shuffles like rank 0
                seed = box[0]
            kw = dict(num_eig_features=c.num_eig_features, use_eig=c.use_eig) if "use_eig" in c else \
                dict(num_svd_features=c.num_svd_features, use_svd=c.use_svd)
            self.dataset = dataset_for_scheme(self.SCHEME, c.dataset_path, max_shuffle_len=c.max_shuffle_len,
                                              splits=tuple(splits), seed=seed, **kw)
            # training_base.py:202-206: evaluation / prediction runs batch EVERY split (the training split included) at
            # batch_size * prediction_bmult; a training run batches every split at batch_size
            mult = c.prediction_bmult if (getattr(self, "eval_flag", False) or getattr(self, "pred_flag", False)) else 1
            bs = {sp: c.batch_size * mult for sp in splits}
            out = self.dataset.get_batched_data(bs, shard=shard)
            out = out if isinstance(out, tuple) else (out,)
            trainset, valset, testset = (list(out) + [None, None])[:3]
        self.trainset, self.valset, self.testset = trainset, valset, testset

    # ---- one step / one epoch ----
    def _batch(self, b):
        dev = self.device
        mv = (lambda t: t.to(dev)) if dev is not None else (lambda t: t)
        return mv(b["node_features"]), mv(b["feature_matrix"]), mv(b["graph_matrix"]), mv(b["target"])

    def _pe(self, b):
        """the positional-encoding inputs of the batch (singular_vectors / eigen_vectors), when the scheme uses them"""
        c, dev = self.config, self.device
        out = {}
        if c.get("use_svd") and "singular_vectors" in b:
            out["singular_vectors"] = b["singular_vectors"] if dev is None else b["singular_vectors"].to(dev)
        if c.get("use_eig") and "eigen_vectors" in b:
            out["eigen_vectors"] = b["eigen_vectors"] if dev is None else b["eigen_vectors"].to(dev)
        return out

    def batch_loss(self, batch):
        """(loss, metric sums) of one batch: scheme-specific"""
        nf, fm, adj, tgt = self._batch(batch)
        y = self.model(nf, fm, adj, **self._pe(batch))
        loss = self.loss_fn(y, tgt)
        sabs = (y - tgt).abs().sum().detach()
        return loss, dict(mae=(sabs, tgt.numel()), loss=(sabs, tgt.numel()))      # (the MAE loss IS the mae metric)

    def _graphed_loss(self, batch):
        """config.use_hipgraph: forward + loss + backward of a training batch as ONE hipGraph launch.  A graph is captured
        per batch geometry (the padded node count varies from batch to batch) around static copies of the batch tensors;
        all graphs accumulate into the scheme's one flat gradient buffer and advance the same device-resident mask seeds.
        The first batch of a new geometry pays the capture (one eager warm-up step, whose sample is discarded)."""
        from .graph import GraphedStep
        dev = self.device
        moved = {k: v.to(dev) for k, v in batch.items() if torch.is_tensor(v)}
        key = tuple((k, tuple(v.shape), str(v.dtype)) for k, v in sorted(moved.items()))
        ent = self._graphs.get(key)
        if ent is None:
            static = {k: v.clone() for k, v in moved.items()}

            def fn():
                self.flat.zero(); self.flat.rebind()
                loss, _ = self.batch_loss(static)
                loss.backward()
                return loss.detach()
            # one graph (with its own static batch and activation pool) per padded geometry: keep the most recently used
            # MAX_GRAPHS of them -- a dataset padded to each batch's longest graph meets dozens of geometries
            while len(self._graphs) >= self.MAX_GRAPHS:
                self._graphs.pop(next(iter(self._graphs)))
            ent = self._graphs[key] = (static, GraphedStep(fn, self._seeds, warmup=1))
        else:
            self._graphs[key] = self._graphs.pop(key)          # most recently used last
            for k, v in moved.items():
                ent[0][k].copy_(v, non_blocking=True)
        loss = ent[1].replay()
        self.flat.rebind()                           # .grad = this buffer's views, whichever graph ran last
        return loss

    def train_step(self, batch):
        c = self.config
        if c.warmup_steps > 0:
            lr, stop = warmup_cosine_lr(self.state.global_step, c.warmup_steps, c.initial_lr, c.total_steps)
            if lr is not None:
                self.set_lr(lr)
            self.stop_training |= stop
        self.model.train()
        if self._use_graph:
            loss = self._graphed_loss(batch)
        else:
            if self.flat is not None:
                self.flat.zero(); self.flat.rebind()
            else:
                self.optimizer.zero_grad(set_to_none=True)
            loss, _ = self.batch_loss(batch)
            loss.backward()
        if self.flat is not None:
            # a rank's slice of a short last batch can be one graph smaller than another's: weight by graph counts, so that
            # every graph of the GLOBAL batch counts once (the Keras loss is a mean over the global batch)
            lc, gc = batch.get("_local_count"), batch.get("_global_count")
            self.flat.all_reduce(average=True, local_count=lc, global_count=gc)
        if c.gradient_clipval is not None:           # Keras clipvalue: elementwise clip of every gradient
            torch.nn.utils.clip_grad_value_(self.params, c.gradient_clipval)
        self.optimizer.step()
        self.state.global_step += 1                  # on_batch_end (:136)
        return float(loss.detach())

    @torch.no_grad()
    def evaluate(self, dataset, max_steps=None):
        """validation metrics: {name: weighted mean}"""
        self.model.eval()
        acc = {}
        for i, b in enumerate(dataset):
            if max_steps is not None and i >= max_steps:
                break
            _, ms = self.batch_loss(b)
            for k, (sm, cnt) in ms.items():
                a = acc.setdefault(k, [0.0, 0.0]); a[0] += float(sm); a[1] += float(cnt)
        if self.config.distributed and _dist_on():    # the split is sharded: sums and counts of all ranks
            keys = self.metric_keys()                 # a fixed list per scheme: a rank without batches sends zeros of the same shape
            t = torch.tensor([x for k in keys for x in acc.get(k, [0.0, 0.0])], dtype=torch.float64,
                             device=self.device if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(t)
            acc = {k: [float(t[2 * i]), float(t[2 * i + 1])] for i, k in enumerate(keys)}
        return {k: v[0] / max(v[1], 1e-30) for k, v in acc.items()}

    LOSS_METRIC = "mae"      # the metric that IS the training loss of the scheme (None: none of them)

    def metric_keys(self):
        return sorted(set(self.get_metrics()) | {"loss"})

    def train_model(self):                           # model.fit (:293-302) with the callbacks' behaviour inlined
        c = self.config
        for epoch in range(self.state.current_epoch, c.num_epochs):
            if self.stop_training:
                break
            losses = []
            for i, b in enumerate(self.trainset):
                if c.steps_per_epoch is not None and i >= c.steps_per_epoch:
                    break
                losses.append(self.train_step(b))
                if self.stop_training:
                    break
            loss_mean = float(np.mean(losses)) if losses else math.nan
            if c.distributed and _dist_on():         # the logged training loss is the mean over all ranks (a rank-local value would let a train-metric monitor desynchronise them)
                t = torch.tensor([loss_mean if losses else 0.0, 1.0 if losses else 0.0], dtype=torch.float64,
                                 device=self.device if torch.distributed.get_backend() == "nccl" else "cpu")
                torch.distributed.all_reduce(t)
                loss_mean = float(t[0] / t[1]) if float(t[1]) > 0 else math.nan
            logs = dict(loss=loss_mean)
            if self.LOSS_METRIC is not None:
                logs[self.LOSS_METRIC] = logs["loss"]
            if self.valset is not None:
                v = self.evaluate(self.valset, c.validation_steps)
                logs.update({"val_" + k: x for k, x in v.items()})   # batch_loss always reports 'loss': val_loss is the validation LOSS
            # epoch-end order of the reference's callback list: training callbacks (SaveWhen) first, then the
            # checkpoint callback, whose on_epoch_end runs the state updates and saves (:249-256, checkpoint.py:66-83)
            scope = dict(logs); scope["epoch"] = epoch + 1; scope.update(self.state.items())
            for name in self.save_when.fire("epoch", scope):
                self.save_weights(os.path.join(os.path.dirname(c.saved_model_path), name + ".npz"))
            self.state.current_epoch += 1            # on_epoch_end[0] (:137)
            if c.save_best:
                self.stop_training |= save_best_update(c, self.state, self.get_lr, self.set_lr, logs, self.print)
            self.print(f"\nCHECKPOINT Epoch: {epoch + 1}  " + "  ".join(f"{k}={v:.5f}" for k, v in logs.items()) + f"  lr={self.get_lr():.3g}")
            self.save_checkpoint()
            self.history.append(dict(epoch=epoch + 1, lr=self.get_lr(), **logs))

    def finalize_training(self):                     # :321-327
        self.save_weights(self.config.saved_model_path + ".npz")
        self.print("DONE!!!")

    # ---- evaluation drivers (training_base.py:330-392) ----
    def get_latest_save_file(self):                  # :330-344 (.npz instead of .h5)
        import re
        from pathlib import Path
        pattern = re.compile(r"(?<=epoch)[0-9]+")
        cur_epoch, cur_file = 0, ""
        for fp in Path(self.config.saved_model_path).parent.glob("*.npz"):
            m = pattern.search(fp.name)
            e = 0 if m is None else int(m.group())
            if e > cur_epoch:
                cur_epoch, cur_file = e, str(fp)
        self.config["weight_file"] = cur_file

    def prepare_for_test(self, trainset=None, valset=None, testset=None):   # :347-363
        self.config_summary()
        self.load_data(trainset, valset, testset, splits=("training", "validation", "test"))
        self.load_model()
        c = self.config
        if c.weight_file == ":":
            self.get_latest_save_file()
        if c.weight_file == "":
            c["weight_file"] = c.saved_model_path + ".npz"
        if c.weight_file == "-":
            self.load_state()
            self.print("LOADED TRAINING STATE FOR PREDICTIONS!")
        else:
            self.load_weights(c.weight_file)
            self.print(f'LOADED WEIGHT FILE "{c.weight_file}" FOR PREDICTIONS!')

    def _report(self, split, lines):
        """print + append to predictions/<split>_evals.txt (schemes/zinc/_eval.py:10-12); rank 0 writes"""
        for ln in lines:
            self.print(ln)
        if _rank() == 0:
            with open(os.path.join(self.config.predictions_path, f"{split}_evals.txt"), "a") as fl:
                for ln in lines:
                    print(ln, file=fl)

    def do_evaluations_on_split(self, split):        # ZINCEval (schemes/zinc/_eval.py:6-12)
        mae = self.evaluate(getattr(self, split))["mae"]
        self._report(split, [f"{split} MAE = {mae:0.5f}"])

    def do_evaluations(self, trainset=None, valset=None, testset=None):   # :383-392
        self.eval_flag = True
        self.prepare_for_test(trainset, valset, testset)
        if _rank() == 0:
            os.makedirs(self.config.predictions_path, exist_ok=True)
        _barrier()
        for split in ("trainset", "valset", "testset"):
            if getattr(self, split, None) is None:
                continue
            self.print("=" * 40)
            self.print(f"Evaluation on {split}.")
            src = getattr(self, split)
            had = getattr(src, "collective", None)
            if hasattr(src, "collective"):
                src.collective = False     # an evaluation pass has ONE collective at its end: no graph is dropped
            try:
                self.do_evaluations_on_split(split)
            finally:
                if hasattr(src, "collective"):
                    src.collective = had
            self.print("")

    def execute_training(self, trainset=None, valset=None):   # :305-312
        self.config_summary()
        self.save_config_file()
        self.load_data(trainset, valset)
        self.load_model()
        self.load_state()
        self.train_model()
        self.finalize_training()


class PatternSVDScheme(ZincSVDScheme):
    """lib.training.schemes.pattern.svd.SCHEME (SBMPDCSVD): node classification with the class-weighted sparse
    cross-entropy (lib/base/genutil/losses.py), metrics xent + acc, monitors val_xent."""
    SCHEME = "pattern.svd"

    def get_model(self):
        if self.model_factory is not None:
            return self.model_factory(self.get_model_config())
        from .model import PatternDCTransformer
        return PatternDCTransformer(**self.get_model_config())

    def get_loss(self):
        from .model import weighted_sparse_xent_loss
        return weighted_sparse_xent_loss

    LOSS_METRIC = "xent"

    def get_metrics(self):
        return ["xent", "acc"]

    def batch_loss(self, batch):
        from .model import class_weights_from_sizes
        dev = self.device
        mv = (lambda t: t.to(dev)) if dev is not None else (lambda t: t)
        nf, adj, tgt = mv(batch["node_features"]), mv(batch["graph_matrix"]), mv(batch["target"])
        out = self.model(nf, adj, return_mask=True, **self._pe(batch))
        logits, mask = out
        if getattr(self, "_class_w", None) is None or self._class_w.device != logits.device:
            self._class_w = class_weights_from_sizes(self.config.class_sizes, device=logits.device)   # once: a host -> device copy cannot be captured
        w = self._class_w
        loss = self.loss_fn(logits, tgt, mask, w)
        m = mask.to(logits.dtype)
        hit = ((logits.argmax(-1) == tgt).to(logits.dtype) * m).sum().detach()
        # Keras feeds the mask as sample_weight: the metrics are means over the REAL nodes (losses.py:108-118)
        logp = torch.log_softmax(logits.detach(), -1).gather(-1, tgt.clamp(min=0).long()[..., None])[..., 0]
        xs = (-(logp) * w[tgt.clamp(min=0).long()] * m).sum()
        # `loss` of a batch is a mean over its padded (graph, node) slots (Keras SUM_OVER_BATCH_SIZE); the EPOCH figure Keras reports is
        # its compiled-loss Mean metric, which LossesContainer updates with sample_weight = the batch dimension (number of graphs):
        # N is padded per batch, so a slot-weighted epoch mean would differ from the reference's `loss` / `val_loss`
        graphs = torch.full((), float(m.shape[0]), device=m.device, dtype=m.dtype)   # (a fill kernel: capturable, unlike a host -> device copy)
        return loss, dict(xent=(xs, m.sum()), acc=(hit, m.sum()), loss=(loss.detach() * graphs, graphs))


    @torch.no_grad()
    def do_evaluations_on_split(self, split):        # SBMPATTERNEval (schemes/pattern/_eval.py:9-111)
        from sklearn.metrics import recall_score, accuracy_score, confusion_matrix
        self.model.eval()
        dev = self.device
        mv = (lambda t: t.to(dev)) if dev is not None else (lambda t: t)
        targs, preds = [], []
        for b in getattr(self, split):
            nf = mv(b["node_features"])
            logits = self.model(nf, mv(b["graph_matrix"]), **self._pe(b))
            keep = (nf >= 0).reshape(-1).cpu().numpy()                                   # collate_fn: node_features >= 0
            targs.append(b["target"].reshape(-1).cpu().numpy()[keep])
            preds.append(torch.softmax(logits, -1)[..., 1].reshape(-1).cpu().numpy()[keep])
        targs = np.concatenate(targs) if targs else np.zeros(0, dtype=np.int64)   # (a rank without a share of a tiny split)
        preds = np.concatenate(preds) if preds else np.zeros(0, dtype=np.float32)
        if self.config.distributed and _dist_on():   # the split is sharded: gather every rank's nodes (strategy ... concat, :62-78)
            box = [None] * torch.distributed.get_world_size()
            torch.distributed.all_gather_object(box, (targs, preds))
            targs, preds = np.concatenate([x[0] for x in box]), np.concatenate([x[1] for x in box])
        pred_class = np.round(preds).astype(targs.dtype)
        classes = (np.eye(targs.max() + 1)[targs]).sum(0)

        def accuracy_sbm(t, c):                      # :11-26 (mean per-class recall over the classes of the confusion matrix)
            cm = confusion_matrix(t, c).astype(np.float32)
            per = [cm[r, r] / float((t == r).sum()) if (t == r).sum() else 0.0 for r in range(cm.shape[0])]
            return float(np.sum(per)) / cm.shape[0]

        def weighted_log_loss(t, pr, w, eps=1 - 9):  # :34-39 (the reference's eps literal, 1-9, is kept: it clips to [-8, 9])
            sw = w[t.astype("int64")].astype("float32")
            t = np.clip(t.astype("float32"), 0., 1.)
            pr = np.clip(pr.astype("float32"), eps, 1. - eps)
            with np.errstate(divide="ignore", invalid="ignore"):
                return float((-(t * np.log(pr) + (1 - t) * np.log(1 - pr)) * sw).mean())

        cs = np.array(self.config.class_sizes, dtype="float32")
        cw = (cs.sum() - cs) / (cs.sum() - cs).sum()
        macro_rec = recall_score(targs, pred_class, average="macro")
        micro_rec = recall_score(targs, pred_class, average="micro")
        acc = accuracy_score(targs, pred_class)
        wacc = accuracy_sbm(targs, pred_class)
        ll = weighted_log_loss(targs, preds, cw)
        self.print(f"Binned classes:{classes}")
        self._report(split, [f"Accuracy = {acc:0.5%}", f"Micro Recall = {macro_rec:0.5%}", f"Macro Recall = {micro_rec:0.5%}",
                             f"Weighted Accuracy = {wacc:0.5%}", f"Log loss:{ll:0.5f}"])   # (labels as the reference prints them)


class ZincEigScheme(ZincSVDScheme):
    """lib.training.schemes.zinc.eig.SCHEME (ZincDCEig): the ZINC model with Laplacian-eigenvector positional encodings
    (EigenDataset + DCEigTransformer; BASELINE config 1, configs/main/zinc/100k/egt_epe.json)."""
    SCHEME = "zinc.eig"


class PatternEigScheme(PatternSVDScheme):
    """lib.training.schemes.pattern.eig.SCHEME (SBMPDCEig): PATTERN with eigenvector encodings; metric acc
    (schemes/pattern/eig.py:48-50), monitors at the base default."""
    SCHEME = "pattern.eig"

    def get_metrics(self):
        return ["acc"]

    LOSS_METRIC = None       # its only metric is the accuracy

    def batch_loss(self, batch):
        loss, ms = super().batch_loss(batch)
        return loss, dict(acc=ms["acc"], loss=ms["loss"])


class SyntheticPattern:
    """Batches in the PATTERN input format (lib/data/datasets/sbm_pattern.py): node_features [B,N] int in {0,1,2}
    (padding -1), graph_matrix [B,N,N] 0/1, target [B,N] int class per node (0 on padded slots).  A planted pattern:
    nodes of class 1 are more densely connected among themselves."""

    def __init__(self, n_graphs=512, batch_size=128, nodes=(44, 188), seed=0, pad_multiple=1, device="cpu"):
        g = torch.Generator().manual_seed(seed)
        self.n = torch.randint(nodes[0], nodes[1] + 1, (n_graphs,), generator=g)
        self.seed, self.batch_size, self.pad_multiple, self.device = seed, batch_size, pad_multiple, device

    def __len__(self):
        return (len(self.n) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for b in range(len(self)):
            ns = self.n[b * self.batch_size:(b + 1) * self.batch_size]
            g = torch.Generator().manual_seed(self.seed * 100003 + b)
            B, N = len(ns), int(ns.max())
            N = (N + self.pad_multiple - 1) // self.pad_multiple * self.pad_multiple
            real = torch.arange(N)[None, :] < ns[:, None]
            cls = (torch.rand(B, N, generator=g) < 0.18).long() * real
            same = (cls[:, :, None] == 1) & (cls[:, None, :] == 1)
            pr = torch.where(same, torch.tensor(0.5), torch.tensor(0.08))
            adj = (torch.rand(B, N, N, generator=g) < pr).float()
            adj = ((adj + adj.transpose(1, 2)) > 0).float() * (real[:, :, None] & real[:, None, :]).float() * (1 - torch.eye(N))[None]
            nf = torch.randint(0, 3, (B, N), generator=g); nf[~real] = -1
            yield dict(node_features=nf.int().to(self.device), graph_matrix=adj.to(self.device), target=cls.to(self.device))


class Cifar10SVDScheme(ZincSVDScheme):
    """lib.training.schemes.cifar10.svd.SCHEME (CIFAR10DCSVD): graph classification, SparseCategoricalCrossentropy from
    logits, metrics acc + xent, monitors val_xent."""
    SCHEME = "cifar10.svd"

    def get_model(self):
        if self.model_factory is not None:
            return self.model_factory(self.get_model_config())
        from .model import Cifar10DCTransformer
        return Cifar10DCTransformer(**self.get_model_config())

    def get_loss(self):
        from .model import sparse_xent_loss
        return sparse_xent_loss

    LOSS_METRIC = "xent"

    def get_metrics(self):
        return ["xent", "acc"]     # (the reference lists ['acc', xent]; the loss is the x-ent)

    def batch_loss(self, batch):
        nf, fm, adj, tgt = self._batch(batch)
        logits = self.model(nf, fm, adj, **self._pe(batch))
        loss = self.loss_fn(logits, tgt)
        n = tgt.numel()
        return loss, dict(xent=(loss.detach() * n, n), acc=((logits.argmax(-1) == tgt).sum().detach(), n), loss=(loss.detach() * n, n))


    def do_evaluations_on_split(self, split):        # schemes/cifar10/svd.py:45-53
        v = self.evaluate(getattr(self, split))
        self._report(split, [f"{split} accuracy = {v['acc']:0.5%}", f"{split} crossentropy = {v['xent']:0.6f}"])


class SyntheticCifar10:
    """Batches in the CIFAR10 superpixel-graph input format (lib/data/datasets/cifar10.py): node_features [B,N,5] float
    (RGB mean + x,y; padding -1), feature_matrix [B,N,N,1] float (edge feature on edges, -1 elsewhere), graph_matrix
    [B,N,N] 0/1 (8 nearest neighbours, symmetrised), target [B] int class in 0..9 (a function of the mean colour)."""

    def __init__(self, n_graphs=512, batch_size=128, nodes=(85, 150), seed=0, pad_multiple=1, device="cpu"):
        g = torch.Generator().manual_seed(seed)
        self.n = torch.randint(nodes[0], nodes[1] + 1, (n_graphs,), generator=g)
        self.seed, self.batch_size, self.pad_multiple, self.device = seed, batch_size, pad_multiple, device

    def __len__(self):
        return (len(self.n) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for b in range(len(self)):
            ns = self.n[b * self.batch_size:(b + 1) * self.batch_size]
            g = torch.Generator().manual_seed(self.seed * 100003 + b)
            B, N = len(ns), int(ns.max())
            N = (N + self.pad_multiple - 1) // self.pad_multiple * self.pad_multiple
            real = torch.arange(N)[None, :] < ns[:, None]
            cls = torch.randint(0, 10, (B,), generator=g)
            nf = torch.rand(B, N, 5, generator=g)
            nf[..., 0] = (nf[..., 0] * 0.5 + cls[:, None].float() / 20).clamp(0, 1)       # class-dependent colour channel
            pos = nf[..., 3:5]
            d = (pos[:, :, None, :] - pos[:, None, :, :]).pow(2).sum(-1) + (~real)[:, None, :].float() * 10 + torch.eye(N)[None] * 10
            knn = d.topk(min(8, N - 1), dim=-1, largest=False).indices
            adj = torch.zeros(B, N, N).scatter_(2, knn, 1.0)
            adj = ((adj + adj.transpose(1, 2)) > 0).float() * (real[:, :, None] & real[:, None, :]).float()
            fm = torch.where(adj > 0, torch.exp(-d).clamp(max=1.0), torch.tensor(-1.0))[..., None]
            nf = torch.where(real[..., None], nf, torch.tensor(-1.0))
            yield dict(node_features=nf.to(self.device), feature_matrix=fm.to(self.device), graph_matrix=adj.to(self.device),
                       target=cls.to(self.device))


def import_scheme(name: str):
    """lib/training/importer.py:3-11."""
    if name == "cifar10.svd":
        return Cifar10SVDScheme
    if name == "zinc.svd":
        return ZincSVDScheme
    if name == "pattern.svd":
        return PatternSVDScheme
    if name == "zinc.eig":
        return ZincEigScheme
    if name == "pattern.eig":
        return PatternEigScheme
    raise KeyError(f"scheme {name!r}: only {SCHEMES} are built")


def main(argv=None):
    """python -m egt_amd.training cfg.json [--synthetic N_GRAPHS] [--evaluate]   (run_training.py:5-10, do_evaluations.py).
    The scheme's dataset is read from config.dataset_path when that file exists (egt_amd.data); otherwise, or with
    --synthetic, the run trains on synthetic graphs in the reference's batch format (with the scheme's positional
    encodings computed by the data pipeline's functions).  Under `torchrun` (WORLD_SIZE > 1) the process group is created
    here: one process per GPU, RCCL ('nccl')."""
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m egt_amd.training cfg.json [--synthetic N_GRAPHS] [--evaluate]")
    config = read_config_from_file(argv[0])
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not _dist_on():
        torch.cuda.set_device(local)
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local))
    scheme = import_scheme(config["scheme"])(config, device=torch.device("cuda", local))
    evaluate = "--evaluate" in argv
    if "--synthetic" not in argv and os.path.exists(scheme.config.dataset_path):
        scheme.do_evaluations() if evaluate else scheme.execute_training()
        return
    n_graphs = int(argv[argv.index("--synthetic") + 1]) if "--synthetic" in argv else 2048
    c = scheme.config
    bs = c.batch_size
    name = config["scheme"].split(".")[0]
    data = {"pattern": SyntheticPattern, "cifar10": SyntheticCifar10}.get(name, SyntheticZinc)

    def mk(n, seed):
        d = data(n, bs, seed=seed)
        if c.get("use_eig"):
            return WithPositional(d, "eig", c.num_eig_features)
        if c.get("use_svd"):
            return WithPositional(d, "svd", c.num_svd_features)
        return d
    if evaluate:
        scheme.do_evaluations(mk(n_graphs, 1), mk(max(bs, n_graphs // 8), 2), mk(max(bs, n_graphs // 8), 3))
    else:
        scheme.execute_training(mk(n_graphs, 1), mk(max(bs, n_graphs // 8), 2))


if __name__ == "__main__":
    main()
