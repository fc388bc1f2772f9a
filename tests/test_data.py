"""Data pipeline: record store, per-graph matrices, shuffling, padded batches, DP shards,
and a scheme trained straight from a dataset file -- all on CPU.  Expected values are derived by hand from the
reference semantics (lib/data/graph.py, svd.py, eigen_gt.py, dataset_base.py, datasets/*.py)."""
import numpy as np
import pytest
import torch

from egt_amd import data as D
from egt_amd import training as T


def _zinc_records(n, rng, nodes=(5, 12)):
    recs = []
    for _ in range(n):
        k = int(rng.integers(nodes[0], nodes[1] + 1))
        src = np.arange(k - 1); dst = src + 1                       # a chain, both directions (molecular graphs are symmetric)
        edges = np.concatenate([np.stack([src, dst], 1), np.stack([dst, src], 1)]).astype(np.int32)
        ef = rng.integers(0, 3, size=len(edges) // 2).astype(np.int32)
        nf = rng.integers(0, 28, size=k).astype(np.int32)
        recs.append(dict(num_nodes=np.int32(k), edges=edges, node_features=nf, edge_features=np.concatenate([ef, ef]),
                         target=np.asarray([np.float32(nf.sum() / 100.0)])))
    return recs


def _store(tmp_path, name="zinc", n_train=50, n_val=10, seed=0, maker=_zinc_records):
    rng = np.random.default_rng(seed)
    spec = D.SPECS[name]
    keys = {f.name: f.key for f in spec.fields}
    path = str(tmp_path / f"{name}.npz")
    D.write_packed_store(path, spec.db_name, dict(training=maker(n_train, rng), validation=maker(n_val, rng)), keys,
                         meta=dict(num_graphs=n_train + n_val))
    return path


# ------------------------------------------------------------------------------------- matrices ---
def test_graph_matrix_scatter_semantics():
    edges = np.array([[0, 1], [1, 0], [0, 1], [2, 2]], np.int32)
    A = D.get_graph_matrix(edges, 3)
    assert A.dtype == np.float32
    assert np.array_equal(A, [[0, 2, 0], [1, 0, 0], [0, 0, 1]])            # scatter_nd adds duplicates
    assert np.array_equal(D.get_graph_matrix(edges, 3, self_loop=True), A + np.eye(3))
    # the 'mark invalid' trick (zinc.py:107-110): features + 1 scattered over zeros, then - 1 -> non-edges are -1
    feats = np.array([0, 2, 1, 1], np.int32)
    F = D.get_graph_matrix(edges[[0, 1, 3]], 3, feats[[0, 1, 3]], increment_by_1=True, decrement_by_1=True)
    assert F.dtype == np.int32 and np.array_equal(F, [[-1, 0, -1], [2, -1, -1], [-1, -1, 1]])
    Ff = D.get_graph_matrix(edges[[0, 1]], 3, np.array([[0.5], [0.25]], np.float32), increment_by_1=True, decrement_by_1=True)
    assert Ff.shape == (3, 3, 1) and Ff[0, 1, 0] == 0.5 and Ff[1, 0, 0] == 0.25 and Ff[2, 2, 0] == -1.0
    # float edge indices are cast (graph.py:25-26)
    assert np.array_equal(D.get_graph_matrix(edges.astype(np.float32), 3), A)


def test_adjacency_normalisation_and_laplacian():
    edges = np.array([[0, 1], [1, 0], [1, 2], [2, 1]], np.int32)
    A = D.get_adjacency(edges, 4, normalize=False, add_self_loops=True)   # node 3 is isolated: only its self loop
    assert np.array_equal(A, [[1, 1, 0, 0], [1, 1, 1, 0], [0, 1, 1, 0], [0, 0, 0, 1]])
    R = D.get_adjacency(edges, 4, normalize=True, add_self_loops=False)   # row-normalised; the empty row stays 0 (divide_no_nan)
    assert np.allclose(R.sum(1), [1, 1, 1, 0]) and not np.isnan(R).any()
    S = D.normalize_adjacency(A, symmetric=True)
    d = A.sum(1)
    assert np.allclose(S, A / np.sqrt(np.outer(d, d)))
    L = D.get_laplacian(edges, 4)
    assert np.allclose(L, np.eye(4) - S, atol=1e-6) and np.allclose(L, L.T)


def test_svd_features_reconstruct_the_matrix_and_pad():
    rng = np.random.default_rng(3)
    A = (rng.random((6, 6)) < 0.4).astype(np.float32)
    UV, S = D.svd_features(A, num_features=8)                              # more features than nodes: zero padded
    assert UV.shape == (6, 8, 2) and S.shape == (8,) and np.all(S[6:] == 0) and np.all(UV[:, 6:] == 0)
    assert np.all(np.diff(S[:6]) <= 1e-6)
    assert np.allclose(UV[:, :, 0] @ UV[:, :, 1].T, A, atol=1e-5)         # (U sqrt S)(V sqrt S)^T = A
    UV3, S3 = D.svd_features(A, num_features=3, mult_sing_vals=False)      # truncated, unscaled: orthonormal columns
    assert UV3.shape == (6, 3, 2) and np.allclose(UV3[:, :, 0].T @ UV3[:, :, 0], np.eye(3), atol=1e-5)
    assert np.allclose(S3, S[:3])
    UVn, _ = D.svd_features(A + np.eye(6, dtype=np.float32), 6, norm_first=True)
    An = D.normalize_adjacency(A + np.eye(6, dtype=np.float32))
    assert np.allclose(UVn[:, :, 0] @ UVn[:, :, 1].T, An, atol=1e-5)


def test_eigen_features_are_laplacian_eigenvectors():
    k = 9
    src = np.arange(k); dst = (src + 1) % k                                # a ring
    edges = np.concatenate([np.stack([src, dst], 1), np.stack([dst, src], 1)]).astype(np.int32)
    V = D.eigen_features(edges, k, 4, sparse=False)
    assert V.shape == (k, 4) and V.dtype == np.float32
    L = np.eye(k) - D.get_graph_matrix(edges, k) / 2.0                     # degree 2 everywhere
    lam = np.sort(np.linalg.eigvalsh(L))
    for j in range(4):                                                     # column j belongs to eigenvalue j+1 (the trivial one is skipped)
        v = V[:, j].astype(np.float64)
        assert np.allclose(L @ v, lam[j + 1] * v, atol=1e-5)
    Vs = D.eigen_features(edges, k, 2, sparse=True)                        # ARPACK path: same subspace (eigenvalue 1 - cos(2 pi / 9), multiplicity 2)
    P = V[:, :2].astype(np.float64)
    assert np.allclose(P @ (P.T @ Vs), Vs, atol=5e-2)


def test_eigen_features_of_a_directed_edge_list_follow_the_reference_dense_form():
    """lib/data/eigen_gt.py:38-57 on an edge list that is not stored in both directions: L = I - D^-1/2 W D^-1/2 with ROW
    degrees is not symmetric; the reference takes np.linalg.eig, sorts the (complex) eigenvalues and keeps the real part of
    the vectors.  Restated here with scipy exactly as the reference writes it."""
    import scipy.sparse as sp
    edges = np.array([[0, 1], [1, 2], [2, 0], [2, 3], [3, 4], [4, 2], [1, 4]], dtype=np.int32)
    n, dim = 5, 3
    A = sp.csr_matrix((np.ones(len(edges), dtype="float32"), (edges[:, 0], edges[:, 1])), shape=(n, n), dtype="float32")
    Nm = sp.diags(np.asarray(A.sum(axis=1)).squeeze().clip(1) ** -0.5, dtype=float)
    L = (sp.eye(n) - Nm * A * Nm).toarray()
    val, vec = np.linalg.eig(L)
    ref = np.real(vec[:, val.argsort()])[:, 1:dim + 1].astype("float32")
    got = D.eigen_features(edges, n, dim, sparse=False)
    assert got.shape == ref.shape
    for j in range(dim):   # an eigenvector is defined up to a (complex) phase: compare up to sign
        assert np.allclose(got[:, j], ref[:, j], atol=1e-5) or np.allclose(got[:, j], -ref[:, j], atol=1e-5)


# ---------------------------------------------------------------------------------------- store ---
def test_packed_store_round_trip(tmp_path):
    path = _store(tmp_path, n_train=7, n_val=3)
    st = D.open_store(path)
    assert int(st.meta("ZINC")["num_graphs"]) == 10
    toks = st.tokens("ZINC", "training")
    assert toks == [f"/ZINC/training/{i:010d}" for i in range(7)]
    ref = _zinc_records(7, np.random.default_rng(0))
    for t, r in zip(toks, ref):
        n, e, nf, ef, y = st.read_record(t, [("data", "num_nodes"), "data/edges", "data/features/nodes/feat",
                                             "data/features/edges/feat", "targets/value"])
        assert n == r["num_nodes"] and np.array_equal(e, r["edges"]) and np.array_equal(nf, r["node_features"])
        assert np.array_equal(ef, r["edge_features"]) and np.array_equal(y, r["target"])
    with pytest.raises(ImportError, match="h5py"):
        D.open_store(str(tmp_path / "x.h5"))                               # the HDF5 backend needs h5py (absent here)


# -------------------------------------------------------------------------------------- dataset ---
def test_record_maps_and_included_features(tmp_path):
    path = _store(tmp_path)
    ds = D.GraphDataset("zinc", path, level="svd", return_mat=True, num_features=4).load_data()
    tok = ds.record_tokens["training"][0]
    r = ds.record("training", tok)
    n = int(r["num_nodes"])
    assert set(r) == {"record_name", "num_nodes", "node_features", "target", "graph_matrix", "feature_matrix", "singular_vectors"}
    assert r["record_name"] == tok.encode()
    assert r["graph_matrix"].shape == (n, n) and np.all(np.diag(r["graph_matrix"]) == 1)          # self loops added
    assert r["feature_matrix"].shape == (n, n) and r["feature_matrix"].dtype == np.int32
    assert np.all(np.diag(r["feature_matrix"]) == -1) and r["feature_matrix"][0, 1] >= 0 and r["feature_matrix"][0, n - 1] == -1
    assert r["singular_vectors"].shape == (n, 4, 2)
    full = D.GraphDataset("zinc", path, level="svd", return_mat=True, return_edges=True, return_edge_features=True,
                          return_sing_vals=True).load_data().record("training", tok)
    assert {"edges", "edge_features", "singular_values"} <= set(full)
    rec = D.GraphDataset("zinc", path, level="records").load_data().record("training", tok)
    assert set(rec) == {"record_name", "num_nodes", "edges", "node_features", "edge_features", "target"}
    eig = D.GraphDataset("zinc", path, level="eigen").load_data().record("training", tok)
    assert eig["eigen_vectors"].shape == (n, 8)                                                     # ZINC EigenDataset defaults (zinc.py:128-141)
    with pytest.raises(KeyError):
        D.GraphDataset("tsp", path)


def test_padded_batches_shuffle_and_remainder(tmp_path):
    path = _store(tmp_path, n_train=50, n_val=10)
    ds = D.dataset_for_scheme("zinc.svd", path, max_shuffle_len=8, seed=5, prefetch_batch=False)
    tr, va = ds.get_batched_data(16, as_torch=False)
    assert len(tr) == 4 and len(va) == 1
    seen = []
    epochs = []
    for _ in range(2):
        order = []
        for b in tr:
            assert set(b) == {"node_features", "target", "graph_matrix", "feature_matrix"}        # excluded: record_name, num_nodes, singular_vectors
            B, N = b["node_features"].shape
            real = b["node_features"] != -1
            nn = real.sum(1)
            assert N == nn.max()                                                                    # padded to the batch's longest graph
            for i in range(B):
                k = nn[i]
                assert np.all(real[i, :k]) and not real[i, k:].any()
                assert np.all(b["graph_matrix"][i, k:] == 0) and np.all(b["graph_matrix"][i, :, k:] == 0)
                assert np.all(b["feature_matrix"][i, k:] == -1) and np.all(b["feature_matrix"][i, :, k:] == -1)
                assert np.isclose(b["target"][i, 0], b["node_features"][i, :k].sum() / 100.0)
            order += [tuple(r[r != -1]) for r in b["node_features"]]
        assert len(order) == 50
        epochs.append(order)
    assert sorted(epochs[0]) == sorted(epochs[1]) and epochs[0] != epochs[1]                        # every graph once per epoch, reshuffled
    v1 = [tuple(r[r != -1]) for b in va for r in b["node_features"]]
    v2 = [tuple(r[r != -1]) for b in va for r in b["node_features"]]
    ref = [tuple(r["node_features"]) for r in _zinc_records(60, np.random.default_rng(0))[50:]]
    assert v1 == v2 == ref                                                                          # validation: store order, never shuffled
    tr_drop, _ = ds.get_batched_data(16, drop_remainder=True, as_torch=False)
    assert len(tr_drop) == 3 and sum(len(b["target"]) for b in tr_drop) == 48
    # a fixed max_length pads every batch to it; a graph that does not fit is an error (tf padded_batch raises too)
    fixed = D.GraphDataset("zinc", path, level="matrix", max_length=14, prefetch_batch=False)
    b = next(iter(fixed.get_batched_data(8, as_torch=False)[0]))
    assert b["node_features"].shape == (8, 14) and b["graph_matrix"].shape == (8, 14, 14) and "edges" not in b
    with pytest.raises(ValueError, match="exceeds"):
        next(iter(D.GraphDataset("zinc", path, level="matrix", max_length=6, prefetch_batch=False).get_batched_data(50, as_torch=False)[0]))
    X, Y = next(iter(ds.get_batched_data(16, map_fns=D.CreateTargets("target"), as_torch=False)[0]))
    assert set(Y) == {"target"} and "target" not in X and "graph_matrix" in X


def test_dp_shards_partition_every_global_batch(tmp_path):
    path = _store(tmp_path, n_train=37, n_val=5)
    outs = []
    for rank in range(3):
        ds = D.dataset_for_scheme("zinc.svd", path, seed=11, prefetch_batch=False)                  # same seed -> same global order on every rank
        outs.append(list(ds.get_batched_data(10, shard=(rank, 3), as_torch=False)[0]))
    whole = D.dataset_for_scheme("zinc.svd", path, seed=11, prefetch_batch=False)
    glob = list(whole.get_batched_data(10, as_torch=False)[0])
    assert [len(o) for o in outs] == [4, 4, 4]
    for step, g in enumerate(glob):
        parts = [o[step]["node_features"] for o in outs]
        sizes = [len(p) for p in parts]
        assert sum(sizes) == len(g["target"]) and max(sizes) - min(sizes) <= 1
        rows = [tuple(r[r != -1]) for p in parts for r in p]
        assert rows == [tuple(r[r != -1]) for r in g["node_features"]]
        for p in parts:                                                                              # each shard padded to ITS longest graph
            assert p.shape[1] == (p != -1).sum(1).max()


def test_short_last_batch_under_dp_drops_only_for_training(tmp_path):
    """A last global batch with fewer graphs than ranks: the TRAINING split drops it on every rank (all ranks take the same
    number of collective steps); validation keeps it -- the ranks that get a share evaluate their graphs, the others skip the
    batch (the reference evaluates every graph; the metric sums are all-reduced once at the end of the split)."""
    path = _store(tmp_path, n_train=22, n_val=12)           # batches of 10 over 4 ranks: training 10 + 10 + 2, validation 10 + 2
    tr, va = [], []
    for rank in range(4):
        ds = D.dataset_for_scheme("zinc.svd", path, seed=3, prefetch_batch=False)
        t, v = ds.get_batched_data(10, shard=(rank, 4), as_torch=False)[:2]
        tr.append(list(t)); va.append(list(v))
    assert [len(o) for o in tr] == [2, 2, 2, 2]                                  # the 2-graph training batch is gone everywhere
    assert sorted(len(o) for o in va) == [1, 1, 2, 2]                           # two ranks hold one graph of the short validation batch
    assert sum(len(b["target"]) for o in va for b in o) == 12                    # every validation graph is evaluated exactly once


def test_pattern_and_cifar_formats(tmp_path):
    def pattern(n, rng):
        recs = []
        for _ in range(n):
            k = int(rng.integers(6, 15))
            e = rng.integers(0, k, size=(3 * k, 2)).astype(np.int32)
            recs.append(dict(num_nodes=np.int32(k), edges=e, node_features=rng.integers(0, 3, k).astype(np.int32),
                             target=rng.integers(0, 2, k).astype(np.int32)))
        return recs

    def cifar(n, rng):
        recs = []
        for _ in range(n):
            k = int(rng.integers(6, 15))
            e = rng.integers(0, k, size=(4 * k, 2)).astype(np.int32)
            recs.append(dict(num_nodes=np.int32(k), edges=e, node_features=rng.random((k, 5), dtype=np.float32),
                             edge_features=rng.random((4 * k, 1), dtype=np.float32), target=np.int32(rng.integers(0, 10))))
        return recs

    p = D.dataset_for_scheme("pattern.svd", _store(tmp_path, "sbm_pattern", 12, 4, maker=pattern), prefetch_batch=False)
    b = next(iter(p.get_batched_data(6)[0]))
    assert set(b) == {"node_features", "graph_matrix", "target"} and b["target"].shape == b["node_features"].shape
    assert b["target"].dtype == torch.int32 and b["graph_matrix"].dtype == torch.float32
    pad = b["node_features"] == -1
    assert torch.all(b["target"][pad] == 0)
    c = D.dataset_for_scheme("cifar10.svd", _store(tmp_path, "cifar10", 12, 4, maker=cifar), prefetch_batch=False)
    b = next(iter(c.get_batched_data(6)[0]))
    B, N, _ = b["node_features"].shape
    assert b["node_features"].shape == (B, N, 5) and b["feature_matrix"].shape == (B, N, N, 1) and b["target"].shape == (B,)
    k = int((b["node_features"][0] != -1).any(-1).sum())
    assert torch.all(b["node_features"][0, k:] == -1) and torch.all(b["feature_matrix"][0, k:] == -1)
    assert b["feature_matrix"][0, :k, :k].max() > 0        # duplicates of a pair ADD their (feature + 1)s, as scatter_nd does


def test_scheme_trains_from_a_dataset_file(tmp_path):
    """TrainingBase.load_data (:207-218) with no iterables: dataset_path -> dataset -> batches -> epochs."""
    from test_training import _Stub
    path = _store(tmp_path, n_train=96, n_val=32, seed=4, maker=lambda n, rng: _zinc_records(n, rng, (9, 30)))
    cfg = dict(scheme="zinc.svd", model_name="d", num_epochs=6, initial_lr=0.01, batch_size=32, use_svd=False,
               save_path=str(tmp_path / "run"), dataset_path=path)
    s = T.ZincSVDScheme(cfg, model_factory=_Stub, print_fn=lambda *a: None)
    s.execute_training()
    assert s.state.current_epoch == 6 and s.state.global_step == 18
    # (the epoch order is drawn from an unseeded generator, like tf.data's shuffle: compare the best epoch, not the last)
    assert min(h["val_mae"] for h in s.history[1:]) < s.history[0]["val_mae"]
    with pytest.raises(FileNotFoundError):
        T.ZincSVDScheme(dict(cfg, dataset_path=str(tmp_path / "missing.npz")), model_factory=_Stub).load_data()
