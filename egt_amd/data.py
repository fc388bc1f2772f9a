"""Data pipeline: record store -> per-graph matrices -> shuffled, padded batches.

What the reference does with h5py + tf.data (lib/data/reader.py:7-65, dataset_base.py:70-130,
graph.py:4-141, svd.py:7-74, eigen_gt.py:6-99, datasets/{zinc,sbm_pattern,cifar10,mnist}.py), restated
as a plain numpy pipeline that hands the model dictionaries of torch tensors in the reference's batch
format (``node_features``, ``feature_matrix``, ``graph_matrix``, ``singular_vectors`` ..., ``target``):

* record stores.  ``H5Store`` reads the reference's HDF5 layout (group ``/<DS>/<split>/<id>`` with datasets
  ``data/edges``, ``data/features/{nodes,edges}/feat``, ``targets/*`` and attribute ``data@num_nodes``) when
  h5py is importable; it raises ``ImportError`` otherwise (h5py is absent from this image).  ``PackedStore`` is
  the same logical layout in one ``.npz``: per (split, key) ONE concatenated array + row offsets, which is also
  the faster format (one mmap-able read per key instead of 10^5 tiny HDF5 datasets).  ``write_packed_store``
  writes it, ``convert_h5`` turns a reference ``.h5`` into it where h5py exists.
* per-record maps (applied once, results cached in memory like ``dataset.cache()``): adjacency with self
  loops / normalisation / Laplacian (graph.py:4-83), dense feature matrix with the +1 / -1 "mark invalid"
  trick (graph.py:17-38, zinc.py:107-116), SVD positional features (svd.py), Laplacian eigenvectors (eigen_gt.py).
* batching: token shuffle per epoch + a bounded shuffle buffer (dataset_base.py:77-78,105-108), ``padded_batch``
  to the batch's longest graph (schemes pass max_length=None, scheme_base.py:62-69) with the datasets' padding
  values, ``drop_remainder``, excluded features, ``CreateTargets``; optional background prefetch thread.
* DP: ``shard=(rank, world)`` gives each rank its contiguous slice of every GLOBAL batch (egt_amd.dp.shard_batch)
  re-padded to the slice's own longest graph.
"""
from __future__ import annotations

import queue
import threading
from dataclasses import dataclass, field
from typing import Callable, Dict, Iterable, Iterator, List, Optional, Sequence, Tuple, Union

import numpy as np

from .dp import shard_batch

Key = Union[str, Tuple[str, str]]          # dataset path inside a record, or (group, attribute)


# ----------------------------------------------------------------------------------------- stores ---
class RecordStore:
    """meta(db) / tokens(db, split, chunked) / read(token, key) — the three things reader.py does."""

    def meta(self, db_name: str) -> dict:
        raise NotImplementedError

    def tokens(self, db_name: str, split: str, chunked: bool = False) -> List[str]:
        raise NotImplementedError

    def read(self, token: str, key: Key):
        raise NotImplementedError

    def read_record(self, token: str, keys: Sequence[Key]) -> tuple:          # reader.py:34-35
        return tuple(self.read(token, k) for k in keys)

    def close(self):
        pass


class H5Store(RecordStore):
    """The reference's on-disk format, read through h5py (reader.py:7-35)."""

    def __init__(self, path: str):
        try:
            import h5py  # noqa: WPS433 (optional dependency)
        except ImportError as exc:
            raise ImportError("h5py is not installed: convert the dataset with egt_amd.data.convert_h5 on a machine "
                              "that has it, or use a PackedStore (.npz)") from exc
        self._f = h5py.File(path, "r")

    def meta(self, db_name):
        return dict(self._f[db_name].attrs.items())

    def tokens(self, db_name, split, chunked=False):
        top = self._f[db_name][split]
        if not chunked:
            return [f"/{db_name}/{split}/{t}" for t in top]
        return [f"/{db_name}/{split}/{g}/{t}" for g, grp in top.items() for t in grp]

    def read(self, token, key):
        grp = self._f[token]
        if isinstance(key, tuple):
            return grp[key[0]].attrs[key[1]]
        return grp[key][()]

    def close(self):
        self._f.close()


def _kname(key: Key) -> str:
    return f"{key[0]}@{key[1]}" if isinstance(key, tuple) else key


class PackedStore(RecordStore):
    """One .npz: ``<db>/<split>#tokens`` (record ids), and per key ``<db>/<split>/<key>#data`` (records
    concatenated along axis 0), ``#offsets`` (row offsets, len = records + 1; absent for scalars / attributes,
    whose #data has one row per record).  ``<db>#meta/<name>`` holds the dataset attributes."""

    def __init__(self, path: str):
        self._z = np.load(path, allow_pickle=False)
        self._cache: Dict[str, np.ndarray] = {}
        self._index: Dict[str, Dict[str, int]] = {}

    def _arr(self, name):
        if name not in self._cache:
            self._cache[name] = self._z[name]
        return self._cache[name]

    def meta(self, db_name):
        pre = f"{db_name}#meta/"
        return {k[len(pre):]: self._z[k][()] for k in self._z.files if k.startswith(pre)}

    def tokens(self, db_name, split, chunked=False):
        ids = [str(t) for t in self._arr(f"{db_name}/{split}#tokens")]
        self._index[f"/{db_name}/{split}"] = {t: i for i, t in enumerate(ids)}
        return [f"/{db_name}/{split}/{t}" for t in ids]

    def read(self, token, key):
        head, rec = token.rsplit("/", 1)
        if head not in self._index:
            _, db, split = head.split("/", 2)
            self.tokens(db, split)
        i = self._index[head][rec]
        base = f"{head[1:]}/{_kname(key)}"
        data = self._arr(base + "#data")
        if base + "#offsets" in self._z.files:
            off = self._arr(base + "#offsets")
            return data[off[i]:off[i + 1]]
        return data[i]


def open_store(path: str) -> RecordStore:
    return PackedStore(path) if str(path).endswith(".npz") else H5Store(path)


def write_packed_store(path: str, db_name: str, splits: Dict[str, List[dict]], keys: Dict[str, Key],
                       meta: Optional[dict] = None, ids: Optional[Dict[str, List[str]]] = None):
    """splits: {split: [record, ...]}, record = {record_name: array}; keys: record_name -> store key (the
    datasets' ``record_proto[...]['key']``).  Ragged arrays (leading axis varies) get offsets."""
    out = {}
    for split, recs in splits.items():
        names = ids[split] if ids else [f"{i:010d}" for i in range(len(recs))]   # the notebooks' 10-digit ids
        out[f"{db_name}/{split}#tokens"] = np.asarray(names)
        for rname, key in keys.items():
            vals = [np.asarray(r[rname]) for r in recs]
            base = f"{db_name}/{split}/{_kname(key)}"
            if isinstance(key, tuple) or all(v.shape == vals[0].shape for v in vals):
                out[base + "#data"] = np.stack(vals)       # read back as data[i]
            else:
                out[base + "#data"] = np.concatenate(vals, axis=0)
                out[base + "#offsets"] = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.int64)
    for k, v in (meta or {}).items():
        out[f"{db_name}#meta/{k}"] = np.asarray(v)
    np.savez(path, **out)


def convert_h5(h5_path: str, npz_path: str, dataset: str, splits=("training", "validation", "test")):
    """reference .h5 -> PackedStore (needs h5py)."""
    spec = SPECS[dataset]
    src = H5Store(h5_path)
    keys = {f.name: f.key for f in spec.fields}
    recs, ids = {}, {}
    for s in splits:
        toks = src.tokens(spec.db_name, s)
        ids[s] = [t.rsplit("/", 1)[1] for t in toks]
        recs[s] = [dict(zip(keys, src.read_record(t, list(keys.values())))) for t in toks]
    write_packed_store(npz_path, spec.db_name, recs, keys, src.meta(spec.db_name), ids)


# ------------------------------------------------------------------------------- per-graph matrices ---
def add_self_loop_edges(edges: np.ndarray, n: int) -> np.ndarray:                   # graph.py:4-13
    rng = np.arange(n, dtype=edges.dtype)
    return np.concatenate([edges, np.stack([rng, rng], axis=1)], axis=0)


def get_graph_matrix(edges, n: int, features=None, self_loop=False, increment_by_1=False,
                     decrement_by_1=False, dtype=np.float32) -> np.ndarray:
    """graph.py:15-42: scatter_nd ADDS duplicates; features may carry trailing dims."""
    edges = np.asarray(edges)
    if edges.dtype not in (np.int32, np.int64):
        edges = edges.astype(np.int32)
    if features is None:
        features = np.ones([len(edges)], dtype=dtype)
    elif increment_by_1:
        features = features + 1
    features = np.asarray(features)
    mat = np.zeros([n, n] + list(features.shape[1:]), dtype=features.dtype)
    np.add.at(mat, (edges[:, 0], edges[:, 1]), features)
    if self_loop:
        mat += np.eye(n, dtype=mat.dtype).reshape([n, n] + [1] * (mat.ndim - 2))
    if decrement_by_1:
        mat = mat - 1
    return mat


def _divide_no_nan(a, b):
    out = np.zeros(np.broadcast(a, b).shape, dtype=np.result_type(a, b))
    np.divide(a, b, out=out, where=(b != 0))
    return out


def normalize_adjacency(A: np.ndarray, symmetric=False) -> np.ndarray:              # graph.py:45-54
    d = A.sum(axis=1, keepdims=True)
    if not symmetric:
        return _divide_no_nan(A, d)
    dmh = _divide_no_nan(np.float32(1.0), np.sqrt(d))
    return dmh * A * np.swapaxes(dmh, 0, 1)


def get_adjacency(edges, n, normalize=True, symmetric=False, add_self_loops=True):  # graph.py:59-68
    if add_self_loops:
        edges = add_self_loop_edges(np.asarray(edges), n)
    A = get_graph_matrix(edges, n)
    return normalize_adjacency(A, symmetric) if normalize else A


def get_laplacian(edges, n, add_self_loops=True):                                    # graph.py:71-80
    return np.eye(n, dtype=np.float32) - get_adjacency(edges, n, True, True, add_self_loops)


def svd_features(A: np.ndarray, num_features=None, mult_sing_vals=True, norm_first=False, norm_symmetric=False):
    """svd.py:11-70 -> (singular_vectors [n, F, 2] = (U, V) columns scaled by sqrt(S), singular_values [F]).
    tf.linalg.svd returns V (A = U diag(S) V^H), numpy V^H; the sign of a singular pair is the solver's."""
    if norm_first:
        A = normalize_adjacency(A, norm_symmetric)
    U, S, Vh = np.linalg.svd(A.astype(np.float32))
    UV = np.stack([U, Vh.T], axis=0)
    if num_features is not None:
        pad = max(0, num_features - A.shape[0])
        S = np.pad(S, (0, pad))[:num_features]
        UV = np.pad(UV, ((0, 0), (0, 0), (0, pad)))[:, :, :num_features]
    if mult_sing_vals:
        UV = UV * np.sqrt(S)
    return np.transpose(UV, (1, 2, 0)).astype(np.float32), S.astype(np.float32)


def eigen_features(edges, n: int, dim: int, sparse=True) -> np.ndarray:
    """Laplacian positional encoding of one graph (what lib/data/eigen_gt.py:6-57 produces): columns 1..dim of the
    eigenvector matrix of the normalised Laplacian L = I - D^-1/2 W D^-1/2 (row degrees, isolated nodes get degree 1), ascending
    eigenvalue, where W counts the edge multiplicities of the edge list.
    * Symmetric edge list (ZINC, PATTERN: every edge is stored in both directions): L is real symmetric, and the dense symmetric
      eigensolver gives the whole spectrum exactly and deterministically (the reference's ARPACK call at tol 1e-2 returns the
      same subspace up to its tolerance and to the sign of each vector, which the model randomises anyway).
    * Directed edge list: L is NOT symmetric and its spectrum is not that of its symmetric part; the reference's dense form is
      followed as written (eigen_gt.py:54-56: np.linalg.eig, argsort of the complex eigenvalues, real part of the vectors).
    `sparse` is kept for signature compatibility and only switches to the iterative solver for symmetric graphs too large for a
    dense factorisation."""
    edges = np.asarray(edges).reshape(-1, 2)
    W = np.zeros((n, n), dtype=np.float64)
    np.add.at(W, (edges[:, 0], edges[:, 1]), 1.0)
    scale = 1.0 / np.sqrt(np.maximum(W.sum(axis=1), 1.0))
    L = np.eye(n) - scale[:, None] * W * scale[None, :]
    if not np.array_equal(W, W.T):
        val, vec = np.linalg.eig(L)
        vec = np.real(vec[:, np.argsort(val)])
        return np.ascontiguousarray(vec[:, 1:dim + 1]).astype("float32")
    L = 0.5 * (L + L.T)                       # (symmetric up to rounding: make it exactly so for eigh)
    if sparse and n > 4096:
        import scipy.sparse.linalg as spl
        val, vec = spl.eigsh(L, k=min(dim + 1, n - 1), which="SA")
    else:
        val, vec = np.linalg.eigh(L)
    vec = vec[:, np.argsort(val, kind="stable")]
    return np.ascontiguousarray(vec[:, 1:dim + 1]).astype("float32")


# ---------------------------------------------------------------------------------- dataset specs ---
@dataclass(frozen=True)
class Field:
    name: str
    key: Key
    dtype: type
    shape: tuple               # per-record shape, None = ragged
    pad: object                # padding value ('mask' -> the dataset's mask_value)
    padded: tuple              # padded_batch shape; 'L' -> max_length (None = the batch's longest)


@dataclass(frozen=True)
class DatasetSpec:
    db_name: str
    fields: Tuple[Field, ...]
    max_length: Optional[int]
    mask_value: object
    fm_tail: Optional[tuple]   # trailing dims of feature_matrix; None = the dataset has no edge features
    eigen_defaults: Optional[dict] = None


def _graph_fields(node_dtype, node_tail, edge_dtype, edge_tail, target_key, target_dtype, target_shape, target_padded, target_pad):
    f = [Field("num_nodes", ("data", "num_nodes"), np.int32, (), 0, ()),
         Field("edges", "data/edges", np.int32, (None, 2), -1, (None, 2)),
         Field("node_features", "data/features/nodes/feat", node_dtype, (None,) + node_tail, "mask", ("L",) + node_tail)]
    if edge_dtype is not None:
        f.append(Field("edge_features", "data/features/edges/feat", edge_dtype, (None,) + edge_tail, "mask", (None,) + edge_tail))
    f.append(Field("target", target_key, target_dtype, target_shape, target_pad, target_padded))
    return tuple(f)


SPECS: Dict[str, DatasetSpec] = {
    # datasets/zinc.py:9-97 (max_length 40, integer atom / bond types, one regression target)
    "zinc": DatasetSpec("ZINC", _graph_fields(np.int32, (), np.int32, (), "targets/value", np.float32, (1,), (1,), 0.0),
                        40, -1, (), dict(num_features=8, sparse=False)),
    # datasets/sbm_pattern.py:7-79 (no edge features; a class per node)
    "sbm_pattern": DatasetSpec("SBM_PATTERN", _graph_fields(np.int32, (), None, (), "targets/node_labels", np.int32, (None,), ("L",), 0),
                               None, -1, None, dict(num_features=2, sparse=True)),
    # datasets/cifar10.py:7-84 (5 real node features, 1 real edge feature, a class per graph)
    "cifar10": DatasetSpec("CIFAR10", _graph_fields(np.float32, (5,), np.float32, (1,), "targets/label", np.int32, (), (), 0),
                           150, -1.0, (1,)),
    # datasets/mnist.py (as CIFAR10 with 3 node features, max_length 75)
    "mnist": DatasetSpec("MNIST", _graph_fields(np.float32, (3,), np.float32, (1,), "targets/label", np.int32, (), (), 0),
                         75, -1.0, (1,)),
}

LEVELS = ("records", "matrix", "svd", "eigen")     # Dataset / MatrixDataset / SVDDataset / EigenDataset


class GraphDataset:
    """One class for the reference's Dataset -> MatrixDataset -> SVDDataset / EigenDataset towers
    (graph_dataset_base.py:11-167 + datasets/*.py), selected by ``level``; keyword names are the reference's."""

    def __init__(self, dataset: str, dataset_path: Optional[str] = None, level: str = "svd", *,
                 store: Optional[RecordStore] = None,
                 splits=("training", "validation"), shuffle_splits=("training",), max_shuffle_len=10000,
                 prefetch_batch=True, max_length="default", mask_value="default",
                 normalize=False, symmetric=False, laplacian=False, return_edges=False, matrix_pad_value=0.0,
                 mark_invalid_features=True, return_edge_features=False,
                 num_features=None, norm_for_svd=False, norm_sym_for_svd=False, mult_sing_vals=True,
                 return_mat=False, return_sing_vals=False, sparse=None, seed: Optional[int] = None):
        if dataset not in SPECS:
            raise KeyError(f"unknown dataset {dataset!r}")
        if level not in LEVELS:
            raise KeyError(f"unknown level {level!r}")
        self.spec = SPECS[dataset]
        self.level = level
        self.store = store if store is not None else open_store(dataset_path)
        self.splits, self.shuffle_splits = list(splits), list(shuffle_splits)
        self.max_shuffle_len, self.prefetch_batch = max_shuffle_len, prefetch_batch
        self.max_length = self.spec.max_length if max_length == "default" else max_length
        self.mask_value = self.spec.mask_value if mask_value == "default" else mask_value
        self.normalize, self.symmetric, self.laplacian = normalize, symmetric, laplacian
        self.matrix_pad_value = matrix_pad_value
        self.mark_invalid_features = mark_invalid_features
        ed = self.spec.eigen_defaults or {}
        self.num_features = num_features if num_features is not None else (16 if level == "svd" else ed.get("num_features", 8))
        self.sparse = sparse if sparse is not None else ed.get("sparse", True)
        self.norm_for_svd, self.norm_sym_for_svd, self.mult_sing_vals = norm_for_svd, norm_sym_for_svd, mult_sing_vals
        # include_if(...) conditions (graph_dataset_base.py:20-33,51,98-99; zinc.py:104)
        self._dropped = set()
        if level != "records":
            if not return_edges:
                self._dropped.add("edges")
            if self.spec.fm_tail is not None and not return_edge_features:
                self._dropped.add("edge_features")
        if level == "svd":
            if not return_mat:
                self._dropped.add("graph_matrix")
            if not return_sing_vals:
                self._dropped.add("singular_values")
        self._excluded: set = set()
        self.record_tokens: Dict[str, List[str]] = {}
        self._cache: Dict[str, Dict[str, dict]] = {}
        self._rng = np.random.default_rng(seed)

    # ---- metadata -------------------------------------------------------------------------------
    def get_metadata(self):
        return self.store.meta(self.spec.db_name)

    def get_paddings(self) -> dict:
        p = {"record_name": b""}
        for f in self.spec.fields:
            p[f.name] = self.mask_value if f.pad == "mask" else f.pad
        if self.level != "records":
            p["graph_matrix"] = self.matrix_pad_value
            if self.spec.fm_tail is not None:
                p["feature_matrix"] = self.mask_value
        if self.level == "svd":
            p.update(singular_values=0.0, singular_vectors=0.0)
        if self.level == "eigen":
            p["eigen_vectors"] = 0.0
        return p

    def get_padded_shapes(self) -> dict:
        L = self.max_length
        s = {"record_name": ()}
        for f in self.spec.fields:
            s[f.name] = tuple(L if d == "L" else d for d in f.padded)
        if self.level != "records":
            s["graph_matrix"] = (L, L)
            if self.spec.fm_tail is not None:
                s["feature_matrix"] = (L, L) + self.spec.fm_tail
        if self.level == "svd":
            s.update(singular_values=(self.num_features,), singular_vectors=(L, self.num_features, 2))
        if self.level == "eigen":
            s["eigen_vectors"] = (L, self.num_features)
        return s

    def exclude(self, names: Iterable[str]):
        """dataset.map(ExcludeFeatures(...)) — what TrainingBase.load_data does with get_excluded_features()"""
        self._excluded |= set(names)

    # ---- records --------------------------------------------------------------------------------
    def _load_record(self, token: str) -> dict:
        sp = self.spec
        vals = self.store.read_record(token, [f.key for f in sp.fields])
        rec = {"record_name": token.encode()}
        for f, v in zip(sp.fields, vals):
            rec[f.name] = np.asarray(v, dtype=f.dtype).reshape([d if d is not None else -1 for d in f.shape])
        if self.level == "records":
            return rec
        n = int(rec["num_nodes"])
        if self.laplacian:
            rec["graph_matrix"] = get_laplacian(rec["edges"], n)
        else:
            rec["graph_matrix"] = get_adjacency(rec["edges"], n, self.normalize, self.symmetric, True)
        if sp.fm_tail is not None:
            kw = dict(increment_by_1=True, decrement_by_1=True) if self.mark_invalid_features else {}
            rec["feature_matrix"] = get_graph_matrix(rec["edges"], n, rec["edge_features"], **kw)
        if self.level == "svd":
            rec["singular_vectors"], rec["singular_values"] = svd_features(
                rec["graph_matrix"], self.num_features, self.mult_sing_vals, self.norm_for_svd, self.norm_sym_for_svd)
        elif self.level == "eigen":
            rec["eigen_vectors"] = eigen_features(rec["edges"], n, self.num_features, self.sparse)
        for k in self._dropped:
            rec.pop(k, None)
        return rec

    def load_data(self):
        for s in self.splits:
            if s not in self.record_tokens:
                self.record_tokens[s] = self.store.tokens(self.spec.db_name, s)
                self._cache[s] = {}
        return self

    def record(self, split: str, token: str) -> dict:
        c = self._cache[split]
        if token not in c:
            c[token] = self._load_record(token)
        r = c[token]
        return {k: v for k, v in r.items() if k not in self._excluded} if self._excluded else r

    def iter_records(self, split: str, shuffle: Optional[bool] = None) -> Iterator[dict]:
        """one epoch of records: tokens reshuffled per epoch, then the bounded shuffle buffer"""
        self.load_data()
        toks = list(self.record_tokens[split])
        shuffle = (split in self.shuffle_splits) if shuffle is None else shuffle
        if not shuffle:
            for t in toks:
                yield self.record(split, t)
            return
        self._rng.shuffle(toks)
        buf_len = max(1, min(len(toks), self.max_shuffle_len))
        buf: List[str] = []
        for t in toks:                       # tf.data shuffle(buffer): emit a random slot, refill it
            if len(buf) < buf_len:
                buf.append(t)
                continue
            j = int(self._rng.integers(len(buf)))
            out, buf[j] = buf[j], t
            yield self.record(split, out)
        while buf:
            j = int(self._rng.integers(len(buf)))
            buf[j], buf[-1] = buf[-1], buf[j]
            yield self.record(split, buf.pop())

    # ---- batches --------------------------------------------------------------------------------
    def collate(self, recs: List[dict]) -> dict:
        """padded_batch: every component padded to its padded shape (None -> the batch's longest)"""
        pads, shapes = self.get_paddings(), self.get_padded_shapes()
        out = {}
        for k in recs[0]:
            if k == "record_name":
                out[k] = np.asarray([r[k] for r in recs])
                continue
            arrs = [np.asarray(r[k]) for r in recs]
            tgt = tuple(max(a.shape[i] for a in arrs) if d is None else d for i, d in enumerate(shapes[k]))
            for a in arrs:
                if any(x > y for x, y in zip(a.shape, tgt)):
                    raise ValueError(f"{k}: record shape {a.shape} exceeds the padded shape {tgt}")
            buf = np.full((len(arrs),) + tgt, pads[k], dtype=arrs[0].dtype)
            for i, a in enumerate(arrs):
                buf[(i,) + tuple(slice(0, s) for s in a.shape)] = a
            out[k] = buf
        return out

    def batches(self, split: str, batch_size: int, drop_remainder=False, map_fn: Optional[Callable] = None,
                shard: Optional[Tuple[int, int]] = None, as_torch=True, device=None, collective: Optional[bool] = None) -> Iterator:
        """one epoch of batches of ``split`` (get_batched_split + map + prefetch, dataset_base.py:100-130).
        collective: every step of the pass is a collective (gradient all-reduce), so every rank must take the same number of
        steps and a last batch with fewer graphs than ranks is dropped by all; False for evaluation / prediction passes, which
        keep every graph (a rank without a share skips the batch).  Default: the training split's passes are collective."""
        train = (split == "training") if collective is None else bool(collective)
        def gen():
            cur: List[dict] = []
            for r in self.iter_records(split):
                cur.append(r)
                if len(cur) == batch_size:
                    b = self._finish(cur, map_fn, shard, as_torch, device, train)
                    if b is not None:
                        yield b
                    cur = []
            if cur and not drop_remainder:
                b = self._finish(cur, map_fn, shard, as_torch, device, train)
                if b is not None:
                    yield b
        return _prefetch(gen(), 2) if self.prefetch_batch else gen()

    def _finish(self, recs, map_fn, shard, as_torch, device, train=True):
        counts = None
        if shard is not None:
            n_global = len(recs)
            if n_global < shard[1] and train:
                return None        # fewer graphs than ranks: EVERY rank drops this (last) TRAINING batch, so all take the same number of steps
            lo, hi = shard_batch(n_global, shard[1], shard[0])
            if hi == lo:
                return None        # validation / test: the reference evaluates every graph; a rank without a share of a short last batch
                                   # skips it and contributes zero sums and counts to the one all-reduce at the end of the split
            recs = recs[lo:hi]
            counts = (hi - lo, n_global)
        b = self.collate(recs)
        if as_torch:
            import torch
            b = {k: (torch.from_numpy(v).to(device) if device is not None else torch.from_numpy(v))
                 for k, v in b.items() if v.dtype.kind not in "SUO"}
        if counts is not None:     # the gradient all-reduce weights a rank by its share of the global batch (egt_amd.dp)
            b["_local_count"], b["_global_count"] = counts
        return map_fn(b) if map_fn is not None else b

    def get_batched_data(self, batch_size, drop_remainder=False, map_fns=None, **kw):
        """per split an object whose iteration is one (reshuffled) epoch — tf.data semantics"""
        def per_split(v):
            return v if isinstance(v, dict) else {s: v for s in self.splits}
        bs, dr, mf = per_split(batch_size), per_split(drop_remainder), per_split(map_fns)
        self.load_data()
        out = tuple(_Epochs(self, s, bs[s], dr[s], mf[s], kw) for s in self.splits)
        return out if len(out) > 1 else out[0]


class _Epochs:
    def __init__(self, ds, split, batch_size, drop_remainder, map_fn, kw):
        self.ds, self.split, self.a = ds, split, (batch_size, drop_remainder, map_fn)
        self.kw = kw
        self.collective = None     # None: by split name; the scheme driver sets False while it EVALUATES the training split

    def __iter__(self):
        return (b for b in self.ds.batches(self.split, *self.a, collective=self.collective, **self.kw) if b is not None)

    def __len__(self):
        n, bs = len(self.ds.record_tokens[self.split]), self.a[0]
        return n // bs if self.a[1] else -(-n // bs)


def _prefetch(it: Iterator, depth: int) -> Iterator:
    q: "queue.Queue" = queue.Queue(maxsize=depth)
    end = object()

    def work():
        try:
            for x in it:
                q.put(x)
            q.put(end)
        except BaseException as exc:  # noqa: BLE001 (re-raised in the consumer)
            q.put(exc)

    threading.Thread(target=work, daemon=True).start()
    while True:
        x = q.get()
        if x is end:
            return
        if isinstance(x, BaseException):
            raise x
        yield x


class CreateTargets:
    """pipeline.py:54-66: split a batch dict into (inputs, targets)"""

    def __init__(self, target_names):
        self.target_names = [target_names] if isinstance(target_names, str) else list(target_names)

    def __call__(self, inputs):
        X = {k: v for k, v in inputs.items() if k not in self.target_names}
        Y = {k: v for k, v in inputs.items() if k in self.target_names}
        return X, Y


def dataset_for_scheme(scheme: str, dataset_path: str, max_shuffle_len=10000, num_svd_features=16, num_eig_features=8,
                       use_svd=False, use_eig=True, splits=("training", "validation"), **kw) -> GraphDataset:
    """what ``get_dataset()`` builds for the schemes egt_amd.training runs: dataset class + dataset_config
    (scheme_base.py:62-69,125-133,167-171; schemes/{zinc,pattern,cifar10}/svd.py) and the excluded features
    (scheme_base.py:95-98,135-139)."""
    name, lvl = scheme.split(".")
    dataset = {"zinc": "zinc", "pattern": "sbm_pattern", "cifar10": "cifar10", "mnist": "mnist"}[name]
    level = {"svd": "svd", "eig": "eigen", "mat": "matrix"}[lvl]
    args = dict(dataset_path=dataset_path, max_length=None, max_shuffle_len=max_shuffle_len, splits=splits)
    if level == "svd":
        args.update(return_mat=True, normalize=False, num_features=num_svd_features, norm_for_svd=False)
    elif level == "eigen":
        args.update(num_features=num_eig_features)
    args.update(kw)
    ds = GraphDataset(dataset, level=level, **args)
    excl = ["record_name", "num_nodes"]
    if level == "svd" and not use_svd:
        excl.append("singular_vectors")
    if level == "eigen" and not use_eig:
        excl.append("eigen_vectors")                 # scheme_base.py:171-175
    ds.exclude(excl)
    return ds
