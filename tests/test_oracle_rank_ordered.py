"""The declared-order fp32 oracle for evaluate() (oracle/rank_ordered.py + oracle/csrc/rank_ordered.c) pinned on the CPU:
the C chain against an independent numpy restatement (fused multiply-add emulated exactly), the ordered oracle against
the fp64 oracle (equal on dyadic tables and on the reference's rank KAT; within the fragile bound on random tables)."""
import numpy as np
import pytest

from oracle import kge_oracle as O
from oracle import rank_ordered as RO


def test_wave_sum_is_the_documented_tree():
    rng = np.random.default_rng(0)
    x = (rng.normal(size=(5, 64)) * 10.0 ** rng.integers(-3, 4, size=(5, 64))).astype(np.float32)
    got = RO.wave_sum(x)
    f = np.float32
    for r in range(5):
        rows = []
        for q in range(4):
            v = [f(t) for t in x[r, q * 16:(q + 1) * 16]]
            while len(v) > 1:   # neighbours first: ((x0+x1)+(x2+x3)) + ...
                v = [f(v[i] + v[i + 1]) for i in range(0, len(v), 2)]
            rows.append(v[0])
        want = f(f(rows[3] + rows[2]) + f(rows[1] + rows[0]))
        assert got[r] == want


@pytest.mark.parametrize("mode,planes", [(RO.MODE_DOT, 1), (RO.MODE_L1, 1), (RO.MODE_L1_SUB, 1), (RO.MODE_ROT_O, 2), (RO.MODE_ROT_S, 2)])
def test_c_chain_equals_numpy_chain(mode, planes):
    rng = np.random.default_rng(mode)
    n, m, U = 7, 150, 37                      # ragged entity tile (150 = 2 * 64 + 22)
    qplanes = 4 if mode == RO.MODE_ROT_S else planes
    Q = (rng.normal(size=(n, qplanes * U)) * 0.7).astype(np.float32)
    E = (rng.normal(size=(m, planes * U)) * 0.7).astype(np.float32)
    plane = U if planes == 2 else 0
    ref = O.quantise(RO.chain_scores_numpy(mode, Q, E, U, plane, -1.0 if mode != RO.MODE_DOT else 1.0))
    qpos = np.sort(ref, axis=1)[:, m // 2].astype(np.int32)   # a threshold in the middle of each query's scores
    ids = rng.permutation(m)[:90].astype(np.int32)
    for sel in (None, ids):
        counts = np.zeros((n, 2), dtype=np.int32)
        RO.lib().ro_counts(mode, RO._ptr(Q), Q.shape[1], plane, RO._ptr(E), E.shape[1], plane, U, RO._ptr(sel),
                           m if sel is None else len(sel), RO._ptr(qpos), n, -1.0 if mode != RO.MODE_DOT else 1.0, RO._ptr(counts))
        r = ref if sel is None else ref[:, sel]
        assert np.array_equal(counts[:, 0], (qpos[:, None] < r).sum(1)) and np.array_equal(counts[:, 1], (qpos[:, None] == r).sum(1))
    pq = rng.integers(0, n, 40).astype(np.int64)
    pe = rng.integers(0, m, 40).astype(np.int64)
    out = np.zeros(40, dtype=np.int32)
    RO.lib().ro_pair_qscores(mode, RO._ptr(Q), Q.shape[1], plane, RO._ptr(E), E.shape[1], plane, U, RO._ptr(pq), RO._ptr(pe), 40,
                             -1.0 if mode != RO.MODE_DOT else 1.0, RO._ptr(out))
    assert np.array_equal(out, ref[pq, pe])


def test_fma_emulation_catches_double_rounding():
    """acc + q*e where the fp64 sum is a tie for fp32: the round-to-odd emulation and the C fmaf agree (a plain
    fp64 add followed by a cast would round the wrong way)."""
    f = np.float32
    q, e = f(1 + 2.0 ** -12), f(1 + 2.0 ** -12)          # product = 1 + 2^-11 + 2^-24 exactly
    acc = f(2.0 ** 30)                                    # ulp(acc) = 128 in fp32
    for extra in (f(64.0), f(-64.0), f(63.0)):
        Q, E = np.array([[extra, q]], dtype=f), np.array([[f(1.0), e]], dtype=f)
        # chain: fmaf(extra, 1, 0) = extra ; then fmaf(q, e, extra)  -- and one with a large accumulator
        Q2, E2 = np.array([[acc, extra, q]], dtype=f), np.array([[f(1.0), f(1.0), e]], dtype=f)
        for QQ, EE in ((Q, E), (Q2, E2)):
            ref = RO.chain_scores_numpy(RO.MODE_DOT, QQ, EE, QQ.shape[1], 0, 1.0)
            out = np.zeros(1, dtype=np.int32)
            RO.lib().ro_pair_qscores(RO.MODE_DOT, RO._ptr(QQ), QQ.shape[1], 0, RO._ptr(EE), EE.shape[1], 0, QQ.shape[1],
                                     RO._ptr(np.zeros(1, np.int64)), RO._ptr(np.zeros(1, np.int64)), 1, 1.0, RO._ptr(out))
            assert out[0] == O.quantise(ref)[0, 0]


@pytest.mark.parametrize("model", ["TransE", "DistMult", "ComplEx", "HolE"])
def test_ordered_oracle_equals_fp64_oracle_on_dyadic_tables(model):
    """Dyadic-rational tables: every product and partial sum is exact in fp32 in any order, so the two oracle modes must
    agree exactly -- all tie strategies, sides, filters, entities_subset."""
    rng = np.random.default_rng(5)
    N, R, k, n = 90, 3, 6, 40
    K = O.internal_k(model, k)
    ent = (rng.integers(-4, 5, size=(N, K)) / 8.0).astype(np.float32)
    rel = (rng.integers(-4, 5, size=(R, K)) / 8.0).astype(np.float32)
    X = np.stack([rng.integers(0, N, n), rng.integers(0, R, n), rng.integers(0, N, n)], 1)
    fs, fo = O.filter_sets(X, [X, np.stack([rng.integers(0, N, 300), rng.integers(0, R, 300), rng.integers(0, N, 300)], 1)])
    sub = rng.permutation(N)[:30]
    for strat in ("worst", "best", "middle"):
        for side in ("s", "o", "s,o", "s+o"):
            for subset in (None, sub):
                a = O.evaluate_ranks(model, ent, rel, X, fs, fo, side, strat, subset, max_rel_size=R)
                b = RO.evaluate_ranks(model, ent, rel, X, fs, fo, side, strat, subset, max_rel_size=R)
                assert np.array_equal(a, b), (model, strat, side, subset is not None)
    a = O.evaluate_ranks(model, ent, rel, X, None, None, "s,o", "worst", max_rel_size=R)
    assert np.array_equal(a, RO.evaluate_ranks(model, ent, rel, X, None, None, "s,o", "worst", max_rel_size=R))


def test_ordered_oracle_on_the_reference_rank_kat():
    """tests/ampligraph/latent_features/layers/scoring/test_AbstractScoringLayer.py:15-53 (DistMult, ranks [[4,3],[2,1]] 0-based
    + 1) through the ordered mode."""
    f = np.float32
    ent = np.array([[1, 1, 1], [2, 2, 2], [3, 3, 3], [4, 4, 4]], f)
    rel = np.array([[10, 10, 10], [100, 100, 100]], f)
    tri = np.array([[0, 0, 2], [1, 1, 3]])
    got = RO.evaluate_ranks("DistMult", ent, rel, tri, None, None, "s,o", "worst")
    assert np.array_equal(got, np.array([[4, 2], [3, 1]]) + 1)
    got = RO.evaluate_ranks("DistMult", ent, rel, tri, [[0], [1]], [[2], [3]], "s,o", "worst")
    assert np.array_equal(got, np.array([[3, 1], [2, 0]]) + 1)


@pytest.mark.parametrize("model", ["TransE", "DistMult", "ComplEx", "HolE", "RotatE"])
def test_ordered_oracle_within_fragile_bound_of_fp64_oracle(model):
    rng = np.random.default_rng(9)
    N, R, k, n = 700, 5, 50, 60
    K = O.internal_k(model, k)
    ent = (rng.normal(size=(N, K)) * 0.3).astype(np.float32)
    rel = (rng.normal(size=(R, K)) * 0.3).astype(np.float32)
    X = np.stack([rng.integers(0, N, n), rng.integers(0, R, n), rng.integers(0, N, n)], 1)
    a = O.evaluate_ranks(model, ent, rel, X, None, None, "s,o", "worst", max_rel_size=R)
    b = RO.evaluate_ranks(model, ent, rel, X, None, None, "s,o", "worst", max_rel_size=R)
    for c, side in enumerate(("s", "o")):
        frag = O.fragile_rank_mask(model, ent, rel, X, side, max_rel_size=R)
        assert (np.abs(a[:, c] - b[:, c]) <= 2 * frag).all()
    assert (a != b).mean() < 0.1
