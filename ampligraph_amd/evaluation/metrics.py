"""Ranking metrics, same definitions as /root/reference/ampligraph/evaluation/metrics.py:58-62,
108-112,188-192 (plain numpy on the int32 ranks evaluate() returns)."""
import numpy as np


def _flat(ranks):
    r = np.asarray(ranks)
    return r.reshape(-1)


def mrr_score(ranks):
    r = _flat(ranks)
    return float(np.sum(1.0 / r) / len(r))


def mr_score(ranks):
    r = _flat(ranks)
    return float(np.sum(r) / len(r))


def hits_at_n_score(ranks, n):
    r = _flat(ranks)
    return float(np.sum(r <= n) / len(r))
