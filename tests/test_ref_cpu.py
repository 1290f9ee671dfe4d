"""The torch-CPU baseline that bench.py times as `cpu_baseline` must compute what the oracle computes."""
import numpy as np
import pytest
import torch

from oracle import kge_oracle as O
from oracle import ref_cpu


@pytest.mark.parametrize("model", ["TransE", "DistMult", "ComplEx", "HolE", "RotatE"])
def test_ref_cpu_step_matches_oracle(model):
    rng = np.random.default_rng(0)
    N, R, k, B, eta = 50, 4, 8, 20, 3
    K = O.internal_k(model, k)
    ent = (rng.normal(size=(N, K)) * 0.5).astype(np.float32)
    rel = (rng.normal(size=(R, K)) * 0.5).astype(np.float32)
    X = np.stack([rng.integers(0, N, B), rng.integers(0, R, B), rng.integers(0, N, B)], 1).astype(np.int32)
    negs = O.generate_corruptions(X, N, eta, 1, 2)
    tr = ref_cpu.RefCpuTrainer(model, ent, rel, eta, "self_adversarial", 1e-3, max_rel_size=R)
    st = O.TrainState(ent, rel, "adam", 1e-3)
    for step in range(2):
        L = tr.step(X, negs)
        Lo = O.train_step(st, model, X, eta, "self_adversarial", 0, 0, max_rel_size=R, negs=negs)
        assert abs(L - float(Lo)) < 1e-4 * abs(float(Lo))
    frac_close = np.mean(np.abs(tr.ent.detach().numpy() - st.ent) < 2e-5)
    assert frac_close > 0.99


def test_ref_cpu_ranks_match_oracle():
    rng = np.random.default_rng(1)
    N, R, k, n = 120, 3, 6, 40
    for model in ("DistMult", "ComplEx", "TransE"):
        K = O.internal_k(model, k)
        ent = (rng.integers(-4, 5, size=(N, K)) / 8.0).astype(np.float32)
        rel = (rng.integers(-4, 5, size=(R, K)) / 8.0).astype(np.float32)
        X = np.stack([rng.integers(0, N, n), rng.integers(0, R, n), rng.integers(0, N, n)], 1)
        fs, fo = O.filter_sets(X, [X])
        ref = O.evaluate_ranks(model, ent, rel, X, fs, fo, "s,o", "worst")
        got = ref_cpu.rank_batch(model, torch.tensor(ent), torch.tensor(rel), X,
                                 [torch.as_tensor(f, dtype=torch.int64) for f in fs],
                                 [torch.as_tensor(f, dtype=torch.int64) for f in fo]).numpy()
        assert (got == ref).all(), model
