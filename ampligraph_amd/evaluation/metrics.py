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


def rank_score(y_true, y_pred, pos_lab=1):
    """Rank of the positive element among the scores (metrics.py:155-193): 1 + number of elements before it when sorting
    by decreasing score."""
    y_true, y_pred = np.asarray(y_true), np.asarray(y_pred)
    idx = np.argsort(y_pred)[::-1]
    return int(np.where(y_true[idx] == pos_lab)[0][0] + 1)
