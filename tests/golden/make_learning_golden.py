"""Generates tests/golden/learning_mrr_v1.npz: the ORACLE's side of the many-seed learning-parity test
(tests/test_gpu_learning.py::test_mean_mrr_over_seeds_matches_oracle) -- filtered MRR, hits@10 and the last epoch's loss of
the oracle replaying the 40-epoch Adam schedule on tests/planted.py's graph, one entry per (case, seed).  Oracle outputs, not
reference outputs (the reference cannot run here); tests/test_golden.py re-derives a sample of them on the CPU.

    python tests/golden/make_learning_golden.py        # rewrites learning_mrr_v1.npz (deterministic; ~10 min on 8 cores)
"""
import os
import sys
from multiprocessing import Pool

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# (model, loss, number of seeds).  The seed counts follow the spread of the per-seed MRR distance between two fp32
# evaluations of the same schedule (measured with the oracle against itself from tables nudged by one ulp: sd 0.0038 TransE /
# nll, 0.0009 TransE / pairwise, 0.0024 RotatE / self_adversarial, 0.0127 RotatE / nll): the standard error of the mean
# distance is at most a third of the north_star's +-0.002.
CASES = [("TransE", "nll", 64), ("TransE", "pairwise", 64), ("RotatE", "self_adversarial", 64), ("RotatE", "nll", 384)]


def one(job):
    from oracle import kge_oracle as O
    from planted import oracle_learning_run

    model, loss, seed = job
    hist, ranks, _ = oracle_learning_run(model, loss, seed)
    return O.mrr_score(ranks), O.hits_at_n_score(ranks, 10), hist[0], hist[-1]


def build():
    out = {}
    with Pool(os.cpu_count() or 4) as pool:
        for model, loss, n in CASES:
            res = np.asarray(pool.map(one, [(model, loss, s) for s in range(n)]), dtype=np.float64)
            out[f"{model}/{loss}"] = res   # columns: mrr, hits@10, first-epoch loss, last-epoch loss
            print(model, loss, n, "seeds: oracle MRR mean", res[:, 0].mean(), "sd", res[:, 0].std(), flush=True)
    return out


if __name__ == "__main__":
    np.savez_compressed(os.path.join(HERE, "learning_mrr_v1.npz"), **build())
