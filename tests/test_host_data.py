"""Host data semantics (filter sets, id assignment, batching) against the oracle's restatement of the
reference (graph_data_loader.py:287-350,382-439; data_indexer.py:373-399,485-549)."""
import numpy as np

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
