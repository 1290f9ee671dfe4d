"""Host data semantics (filter sets, id assignment, batching) against the oracle's restatement of the
reference (graph_data_loader.py:287-350,382-439; data_indexer.py:373-399,485-549)."""
import numpy as np
import pytest

from ampligraph_amd.datasets.filters import FilterIndex
from oracle import kge_oracle as O


def test_filter_index_matches_oracle_sets():
    rng = np.random.default_rng(0)
    N, R = 40, 3
    mk = lambda n: np.stack([rng.integers(0, N, n), rng.integers(0, R, n), rng.integers(0, N, n)], 1)
    train, valid, test = mk(600), mk(50), mk(80)
    fi = FilterIndex([train, valid, test], N, R)
    fs, fo = fi.as_lists(test)
    rs, ro = O.filter_sets(test, [train, valid, test])
    assert all(set(a) == set(b) and len(a) == len(b) for a, b in zip(fs, rs))
    assert all(set(a) == set(b) and len(a) == len(b) for a, b in zip(fo, ro))
    # unseen (p,o) / (s,p) groups give empty filters
    fi2 = FilterIndex([train[:5]], N, R)
    fs2, fo2 = fi2.as_lists(test)
    rs2, ro2 = O.filter_sets(test, [train[:5]])
    assert [sorted(a) for a in fs2] == [sorted(b) for b in rs2]
    assert [sorted(a) for a in fo2] == [sorted(b) for b in ro2]


def test_filter_index_reference_kat():
    # tests/ampligraph/datasets/test_graph_data_loader.py:76-93
    train = np.array([[1, 1, 2], [1, 1, 3], [1, 1, 4], [5, 1, 3], [5, 1, 4], [6, 1, 3], [6, 1, 2], [6, 1, 4], [6, 1, 7]])
    test = np.array([[3, 1, 2], [4, 1, 3], [5, 1, 4], [5, 1, 2], [1, 1, 5]])
    val = np.array([[3, 1, 6], [2, 1, 2], [1, 1, 6]])
    fi = FilterIndex([train, test, val], 8, 2)
    fs, fo = fi.as_lists(np.array([[1, 1, 2]]))
    assert set(fo[0]) == {2, 3, 4, 5, 6} and set(fs[0]) == {1, 6, 3, 5, 2}
    fi0 = FilterIndex([], 8, 2)
    fs, fo = fi0.as_lists(np.array([[1, 1, 2]]))
    assert len(fs[0]) == 0 and len(fo[0]) == 0


def test_evaluation_protocol_helpers():
    """rank_score docstring KAT (metrics.py:181-185) and the invariants of train_test_split_no_unseen (protocol.py:27-198)."""
    import numpy as np
    import pytest

    from ampligraph_amd.evaluation import filter_unseen_entities, rank_score, train_test_split_no_unseen

    assert rank_score(np.array([0, 0, 1, 0]), np.array([.434, .65, .21, .84])) == 4
    rng = np.random.default_rng(0)
    X = np.stack([rng.integers(0, 30, 400).astype(str), rng.integers(0, 4, 400).astype(str), rng.integers(0, 30, 400).astype(str)], 1)
    state = np.random.get_state()[1][:5].copy()
    tr, te = train_test_split_no_unseen(X, test_size=50, seed=3)
    assert (np.random.get_state()[1][:5] == state).all()        # the caller's numpy stream is left alone
    assert len(te) == 50 and len(tr) == 350
    assert set(te[:, 0]) | set(te[:, 2]) <= set(tr[:, 0]) | set(tr[:, 2]) and set(te[:, 1]) <= set(tr[:, 1])
    both = np.concatenate([tr, te])
    assert sorted(map(tuple, both)) == sorted(map(tuple, X))    # a partition of the input
    tr2, te2 = train_test_split_no_unseen(X, test_size=50, seed=3)
    assert np.array_equal(te, te2) and np.array_equal(tr, tr2)  # seeded
    tr3, te3 = train_test_split_no_unseen(X, test_size=0.1, seed=1, filtered_test_predicates=["1", "2"])
    assert len(te3) == int(0.1 * np.isin(X[:, 1], ["1", "2"]).sum()) and set(te3[:, 1]) <= {"1", "2"}
    X4 = np.concatenate([X, rng.random((400, 2)).astype(str)], 1)            # numeric edge columns behind the triple (FocusE input)
    tr4, te4 = train_test_split_no_unseen(X4, test_size=50, seed=3)
    assert tr4.shape == (350, 5) and te4.shape == (50, 5) and np.array_equal(te4[:, :3], te) and np.array_equal(tr4[:, :3], tr)
    chain = np.array([["a", "r", "b"], ["b", "r", "c"], ["c", "r", "d"]])   # every triple carries an entity seen once
    with pytest.raises(Exception):
        train_test_split_no_unseen(chain, test_size=2)

    class M:
        pass

    from ampligraph_amd.datasets.indexer import DataIndexer

    m = M()
    m.data_indexer = DataIndexer(X)
    Y = np.array([["1", "0", "2"], ["zz", "0", "2"], ["1", "0", "qq"]])
    assert filter_unseen_entities(Y, m).tolist() == [["1", "0", "2"]]
    assert filter_unseen_entities(Y[:1], m) is not None and len(filter_unseen_entities(Y[:1], m)) == 1


def test_row_source_equals_whole_table_draw():
    """initializers.RowSource: any row range of the entity table, and the relation table behind it in the stream, equal
    the whole-table draw (what lets a row-sharded rank initialise only its rows)."""
    from ampligraph_amd.latent_features.initializers import RowSource, initialise, stream_cost

    for e_init, r_init in (("glorot_uniform", "glorot_uniform"), ("he_uniform", "random_uniform"), ("zeros", "glorot_uniform"),
                           ("glorot_normal", "glorot_uniform")):
        rng = np.random.Generator(np.random.PCG64(11))
        ent = initialise(e_init, (333, 20), rng)
        rel = initialise(r_init, (5, 20), rng)
        cost = stream_cost(e_init, (333, 20))
        if e_init == "glorot_normal":
            assert cost is None      # rejection sampling: no fixed stream length, whole-table path is used
            continue
        se = RowSource(e_init, (333, 20), 11, 0)
        assert se.streams
        for lo, hi in ((0, 333), (0, 1), (100, 250), (332, 333), (7, 7)):
            assert np.array_equal(se.rows(lo, hi), ent[lo:hi])
        assert np.array_equal(RowSource(r_init, (5, 20), 11, cost).rows(0, 5), rel)
    arr = np.arange(60, dtype=np.float32).reshape(6, 10)
    assert np.array_equal(RowSource(arr, (6, 10), 0, 0).rows(2, 5), arr[2:5])
    with pytest.raises(ValueError):
        RowSource(arr, (7, 10), 0, 0)


def test_load_from_csv(tmp_path):
    """datasets.load_from_csv: strings, duplicates dropped in first-seen order, optional reciprocal relations
    (reference datasets.py:142-170,173-240)."""
    from ampligraph_amd.datasets import load_from_csv

    (tmp_path / "d.csv").write_text("a,y,b\nb,y,a\na,y,b\na,z,c\n")
    X = load_from_csv(str(tmp_path), "d.csv", sep=",")
    assert X.tolist() == [["a", "y", "b"], ["b", "y", "a"], ["a", "z", "c"]]
    Xr = load_from_csv(str(tmp_path), "d.csv", sep=",", add_reciprocal_rels=True)
    assert Xr.shape == (6, 3) and Xr[3].tolist() == ["b", "y_reciprocal", "a"] and Xr[5].tolist() == ["c", "z_reciprocal", "a"]


def test_device_filter_plumbing_with_a_stub_engine():
    """FilterIndex.device_filter hands the engine the sorted key / start arrays of the right side and returns its ranges with
    the matching id array (the kernel itself, amdkge_filter_ranges, is tested on the GPU): stub engine doing the search in numpy."""
    import torch

    class StubEngine:
        device = torch.device("cpu")

        def filter_ranges(self, keys, start, triples, side, n_ents, n_rels):
            t = triples.numpy().astype(np.int64)
            q = t[:, 1] * n_ents + t[:, 2] if side == 1 else t[:, 0] * n_rels + t[:, 1]
            k, s = keys.numpy(), start.numpy()
            pos = np.minimum(np.searchsorted(k, q), max(0, k.size - 1))
            hit = (k[pos] == q) if k.size else np.zeros(len(q), bool)
            return torch.as_tensor(np.where(hit, s[pos], 0)), torch.as_tensor(np.where(hit, s[np.minimum(pos + 1, s.size - 1)], 0))

    rng = np.random.default_rng(4)
    N, R = 40, 3
    X = np.stack([rng.integers(0, N, 400), rng.integers(0, R, 400), rng.integers(0, N, 400)], 1).astype(np.int32)
    T = np.stack([rng.integers(0, N, 90), rng.integers(0, R, 90), rng.integers(0, N, 90)], 1).astype(np.int32)
    fi = FilterIndex([X], N, R)
    eng = StubEngine()
    for sd, fn, ids in (("s", fi.subject_ranges, fi.s_ids), ("o", fi.object_ranges, fi.o_ids)):
        lo, hi = fn(T)
        l2, h2, got_ids = fi.device_filter(eng, torch.as_tensor(T), sd)
        assert np.array_equal(lo, l2.numpy()) and np.array_equal(hi, h2.numpy()) and np.array_equal(got_ids.numpy(), ids)
    assert fi.device_filter(eng, torch.as_tensor(T), "s")[2] is fi.device_filter(eng, torch.as_tensor(T), "s")[2]   # cached upload


def test_select_best_model_ranking_with_a_stub_model():
    """evaluation.select_best_model_ranking (reference protocol.py:447-933): grid and random search, selection on the odd
    validation rows, failing combinations recorded and skipped, retraining for the early-stopping epoch count, test metrics.
    The model is a stub whose ranks depend on its hyper-parameters (the real class is exercised by the GPU tests)."""
    from ampligraph_amd.evaluation import select_best_model_ranking

    log = []

    class Stub:
        def __init__(self, eta, k, scoring_type, seed=0):
            self.eta, self.k, self.scoring_type = eta, k, scoring_type

        def compile(self, loss, optimizer, entity_relation_regularizer, entity_relation_initializer):
            self.lr, self.loss, self.reg = optimizer.learning_rate, loss.name, entity_relation_regularizer
            if self.k == 13:
                raise RuntimeError("unlucky k")

        def fit(self, X, batch_size, epochs, **kw):
            log.append(("fit", self.k, self.eta, len(X), epochs, kw.get("validation_data") is None or len(kw["validation_data"])))

            class H:
                history = {"loss": [1.0] * min(epochs, 7)}
            return H

        def evaluate(self, X, use_filter, entities_subset=None, corrupt_side="s,o", verbose=False):
            log.append(("eval", self.k, len(X), sorted(use_filter) if use_filter else use_filter))
            base = 1 + abs(self.k - 20) + (0 if self.eta == 2 else 3)      # best: k=20, eta=2
            return np.full((len(X), 2), base, dtype=np.int32)

    rng = np.random.default_rng(0)
    Xtr, Xva, Xte = (rng.integers(0, 9, (n, 3)).astype(str) for n in (50, 11, 7))
    grid = {"k": [10, 13, 20], "eta": [1, 2], "epochs": 30, "optimizer_params": {"learning_rate": [0.1, 0.01]},
            "loss": "nll", "regularizer": "LP", "regularizer_params": {"p": 3, "lambda": 1e-4}}
    best, params, mrr, ranks, test_eval, hist = select_best_model_ranking(
        "ComplEx", Xtr, Xva, Xte, grid, early_stopping_params={"check_interval": 5}, retrain_best_model=True, _model_factory=Stub)
    assert len(hist) == 12 and sum("exception" in h["results"] for h in hist) == 4       # k = 13 fails in compile, 4 times
    assert (params["k"], params["eta"], params["early_stopping_epoch"]) == (20, 2, 7) and mrr == 1.0
    assert best.k == 20 and best.reg.p == 3 and ranks.shape == (7, 2) and test_eval["mrr"] == 1.0 and test_eval["hits_1"] == 1.0
    fits = [e for e in log if e[0] == "fit"]
    assert fits[0][3] == 50 and fits[0][5] == 6                 # early stopping validates on the even rows of X_valid
    assert fits[-1][3] == 50 + 6 and fits[-1][4] == 7 and fits[-1][5] is True   # retrained on train + valid, 7 epochs, no validation
    evals = [e for e in log if e[0] == "eval"]
    assert evals[0][2] == 5 and evals[0][3] == ["test", "train", "valid"] and evals[-1][2] == 7   # odd rows select; X_test last
    # random search: min(max_combinations, size of the list grid) distinct draws (callables count as one choice, as in the
    # reference's total_combinations), reproducible by seed
    runs = []
    for _ in range(2):
        out = select_best_model_ranking("ComplEx", Xtr, Xva, Xte, {"k": [10, 20, 30, 40], "eta": lambda: int(np.random.randint(1, 4))},
                                        max_combinations=5, param_grid_random_seed=3, use_test_for_selection=True,
                                        early_stopping=False, use_filter=False, _model_factory=Stub)
        runs.append([h["model_params"] for h in out[5]])
    assert runs[0] == runs[1] and len(runs[0]) == 4 and len({(p["k"], p["eta"]) for p in runs[0]}) == 4
    # nothing trainable: NaN metrics, no model
    none = select_best_model_ranking("ComplEx", Xtr, Xva, Xte, {"k": 13}, _model_factory=Stub)
    assert none[0] is None and np.isnan(none[4]["mrr"]) and len(none[5]) == 1


def test_get_invalid_keys_docstring_example():
    """data_indexer.py:563-565 (the reference's docstring KAT) and the index form."""
    import numpy as np

    from ampligraph_amd.datasets.indexer import DataIndexer

    train = np.array([["subj_a", "rel_a", "obj_b"], ["subj_c", "rel_b", "obj_c"], ["subj_a", "rel_b", "subj_c"]])
    ix = DataIndexer(train)
    X = np.array([["subj_a", "foo", "subj_c"], ["rel_a", "rel_b", "bar"], ["baz", "obj_b", "obj_c"]])
    s, p, o = ix.get_invalid_keys(X, data_type="raw")
    # ("rel_a" is no entity and "obj_b" no relation either: the reference's docstring lists only the obviously foreign keys)
    assert list(s) == ["rel_a", "baz"] and list(p) == ["foo", "obj_b"] and list(o) == ["bar"]
    Xi = np.array([[0, 0, 1], [99, 1, 2], [1, 7, -1]])
    s, p, o = ix.get_invalid_keys(Xi, data_type="ind")
    assert list(s) == [99] and list(p) == [7] and list(o) == [-1]
    import pytest

    with pytest.raises(Exception):
        ix.get_invalid_keys(X, data_type="nope")


def test_indexer_hash_path_equals_sort_path(monkeypatch):
    """Large text inputs are numbered / looked up through pandas' hash tables, small ones (and pandas-less installs) through
    numpy's sort + binary search: same first-seen ids (data_indexer.py:373-399), same dropped rows (:485-549)."""
    import ampligraph_amd.datasets.indexer as I

    if I._pd is None:
        pytest.skip("pandas not importable")
    rng = np.random.default_rng(3)
    n, N, R = 20000, 900, 11
    X = np.stack([np.char.add("e", rng.integers(0, N, n).astype(str)), np.char.add("r", rng.integers(0, R, n).astype(str)),
                  np.char.add("e", rng.integers(0, N, n).astype(str))], 1)
    a = I.DataIndexer(X)
    Q = X[:9000].copy()
    Q[::13, 2] = "unseen"
    Q[5::17, 1] = "unseen-rel"
    ia, ma = a.get_indexes(Q), a.valid_row_mask(Q)
    monkeypatch.setattr(I, "_pd", None)
    b = I.DataIndexer(X)
    assert np.array_equal(a._ent_raw, b._ent_raw) and np.array_equal(a._rel_raw, b._rel_raw)
    assert np.array_equal(a.get_indexes(X), b.get_indexes(X))
    assert np.array_equal(ia, b.get_indexes(Q)) and np.array_equal(ma, b.valid_row_mask(Q))
    # the reference's rule on its own docstring-sized example: ids in order of first appearance, subject before object
    assert list(a.get_indexes(X[:1])[0]) == [0, 0, 1 if X[0, 0] != X[0, 2] else 0]


def test_integer_label_lookup_table_cache():
    """Dense integer labels go through a cached lookup table (evaluate() maps three filter datasets per call): same ids as the
    binary search, unknown / negative / out-of-range keys drop the row, and two indexers do not share a table."""
    from ampligraph_amd.datasets.indexer import DataIndexer

    rng = np.random.default_rng(0)
    X = np.stack([rng.integers(100, 600, 5000), rng.integers(0, 7, 5000), rng.integers(100, 600, 5000)], 1).astype(np.int32)
    ix = DataIndexer(X)
    ref = DataIndexer(X.astype(str))                         # text labels: the hash / binary-search path
    q = X[rng.permutation(5000)[:800]]
    assert np.array_equal(ix.get_indexes(q), ref.get_indexes(q.astype(str)))
    assert np.array_equal(ix.get_indexes(q), ix.get_indexes(q))           # second call: cached table
    bad = q.copy()
    bad[3, 0], bad[10, 2], bad[20, 1], bad[30, 0] = 10 ** 6, -5, 99, 99    # above, below, unknown relation, inside the span but unseen?
    seen = set(X[:, 0]) | set(X[:, 2])
    keep = np.array([(r[0] in seen) and (r[2] in seen) and (0 <= r[1] < 7) for r in bad])
    out = ix.get_indexes(bad)
    assert out.shape[0] == int(keep.sum()) and np.array_equal(out, ix.get_indexes(bad[keep]))
    other = DataIndexer(X[::-1].copy())                      # different first-seen order: different ids, its own table
    assert not np.array_equal(other.get_indexes(q), ix.get_indexes(q))
    assert np.array_equal(other.get_indexes(q), DataIndexer(X[::-1].astype(str)).get_indexes(q.astype(str)))
    assert ix.get_indexes(q[:0]).shape == (0, 3)
