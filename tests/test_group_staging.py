"""CPU test of the host-side staging of amdkge_session_group_rank for row-sharded groups (ampligraph_amd/csrc/kge_group_staging.h, used
by kge_session_group.hip rows_rank): the product's own C++ functions, compiled with g++ into a small harness (tests/csrc/staging_check.cpp)
that runs one thread per replica as the library does on distinct devices.  Checked against a numpy restatement: every distinct s / o
entity of a chunk gets exactly one scratch slot, exactly one replica (its owner under the reference's bucket rule,
/root/reference/ampligraph/datasets/graph_partitioner.py:339-344: contiguous ranges of ceil(N / W) ids) is asked to gather it, and the
re-indexed queries decode back to the original ids on every replica."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("staging") / "staging_check")
    subprocess.run(["g++", "-O1", "-std=c++17", "-pthread", "-Wall", "-Werror", os.path.join(ROOT, "tests", "csrc", "staging_check.cpp"), "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("W,N,nq,seed", [(1, 50, 40, 0), (2, 2601, 300, 1), (3, 100, 7, 2), (4, 1000, 500, 3), (8, 97, 200, 4), (4, 5, 64, 5)])
def test_staging_of_a_row_sharded_rank_chunk(harness, W, N, nq, seed):
    rng = np.random.default_rng(seed)
    rows_per = -(-N // W)
    T = np.stack([rng.integers(0, N, nq), rng.integers(0, 7, nq), rng.integers(0, N, nq)], 1).astype(np.int32)
    T[: min(5, nq), 2] = T[: min(5, nq), 0]       # s == o queries
    T[-1, 0] = N - 1                               # the last row of the (ragged) last shard
    text = f"{W} {rows_per} {N} {nq}\n" + " ".join(str(int(v)) for v in T.reshape(-1)) + "\n"
    out = subprocess.run([harness], input=text, capture_output=True, text=True, check=True).stdout.strip().split("\n")
    first = np.array(out[0].split(), dtype=np.int64)
    nu, U = int(first[0]), first[1:]
    want_U = np.unique(np.concatenate([T[:, 0], T[:, 2]]))
    assert nu == len(want_U) and np.array_equal(U, want_U)
    owners = np.zeros(nu, dtype=np.int64)
    for d in range(W):
        row = np.array(out[1 + d].split(), dtype=np.int64)
        lo, n_local = int(row[0]), int(row[1])
        assert lo == d * rows_per and n_local == max(0, min(N, lo + rows_per) - lo)
        x, idx = row[2:2 + 3 * nq].reshape(nq, 3), row[2 + 3 * nq:]
        assert len(idx) == nu
        # the queries in the replica's local index space: scratch slot -> entity id gives the original triples back
        assert np.array_equal(U[x[:, 0] - n_local], T[:, 0]) and np.array_equal(U[x[:, 2] - n_local], T[:, 2]) and np.array_equal(x[:, 1], T[:, 1])
        assert x[:, [0, 2]].min() >= n_local and x[:, [0, 2]].max() < n_local + nu
        owned = (U >= lo) & (U < lo + n_local)
        assert np.array_equal(idx >= 0, owned) and np.array_equal(idx[owned], U[owned] - lo)
        owners += owned
    assert np.array_equal(owners, np.ones(nu, dtype=np.int64))          # one contributor per row: the bit-wise sum over replicas IS the row
    assert out[1 + W].split() == ["0", "4", "7", "7"]                    # stage_csr_slice
