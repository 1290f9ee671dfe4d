"""Ranking metrics and protocol helpers with the reference's names (ampligraph/evaluation/__init__.py)."""
from .metrics import hits_at_n_score, mr_score, mrr_score, rank_score
from .protocol import filter_unseen_entities, select_best_model_ranking, train_test_split_no_unseen

__all__ = ["mrr_score", "mr_score", "hits_at_n_score", "rank_score", "train_test_split_no_unseen", "filter_unseen_entities",
           "select_best_model_ranking"]
