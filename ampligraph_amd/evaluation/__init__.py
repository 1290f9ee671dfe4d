from .metrics import hits_at_n_score, mr_score, mrr_score  # noqa: F401
