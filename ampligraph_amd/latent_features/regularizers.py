"""Regulariser surface (/root/reference/ampligraph/latent_features/regularizers.py:14-73): LP with
keys `p` and `lambda` (defaults 2, 1e-5), the alias 'l3', and what tf.keras.regularizers.get accepts by name, which the
reference hands through (regularizers.py:59-73, EmbeddingLookupLayer.py:131-155): 'l1' / 'l2' (factor 0.01) and 'l1_l2'
(0.01 each) or its config dict {"class_name": "L1L2", "config": {"l1": ..., "l2": ...}}.  Every one of these is a sum of at
most two LP terms lambda * sum |x|^p over the WHOLE table; penalty and dense gradient are fused into the HIP optimizer sweep
(amdkge_opt.reg_p / reg_lambda / reg2_p / reg2_lambda).  The entity and the relation table take independent regularisers
(a [entity, relation] pair; either may be None)."""


class LPRegularizer:
    """lambda * sum |x|^p  (+ a second term lambda2 * sum |x|^p2 for Keras' l1_l2)."""

    def __init__(self, p=2, lam=1e-5, p2=None, lam2=0.0):
        if int(p) < 1 or (p2 is not None and int(p2) < 1):
            raise ValueError("LP regularizer needs p >= 1")
        self.p, self.lam = int(p), float(lam)
        self.p2, self.lam2 = (int(p2) if p2 is not None else int(p)), float(lam2)

    @property
    def terms(self):
        """[(p, lambda), ...] with zero-weight terms dropped."""
        return [(p, lam) for p, lam in ((self.p, self.lam), (self.p2, self.lam2)) if lam != 0.0]

    def __call__(self, x):
        """The penalty of a table (numpy), as the Keras regulariser object would return it."""
        import numpy as np

        ax = np.abs(np.asarray(x, dtype=np.float64))
        return float(sum(lam * (ax ** p).sum() for p, lam in self.terms))


def _l1l2(l1, l2):
    l1, l2 = float(l1 or 0.0), float(l2 or 0.0)
    if l1 == 0.0 and l2 == 0.0:
        return None
    if l1 == 0.0:
        return LPRegularizer(2, l2)
    return LPRegularizer(1, l1, 2, l2)


def get(identifier, hyperparams=None):
    hyperparams = dict(hyperparams or {})
    if identifier is None:
        return None
    if isinstance(identifier, LPRegularizer):
        return identifier
    if isinstance(identifier, dict):   # a Keras regulariser config
        name = str(identifier.get("class_name", "")).lower()
        cfg = dict(identifier.get("config", {}))
        if name in ("l1l2", "l1_l2"):
            return _l1l2(cfg.get("l1", 0.0), cfg.get("l2", 0.0))
        if name == "l1":
            return _l1l2(cfg.get("l1", 0.01), 0.0)
        if name == "l2":
            return _l1l2(0.0, cfg.get("l2", 0.01))
    if isinstance(identifier, str):
        if identifier == "l3":
            return LPRegularizer(3, hyperparams.get("lambda", 1e-5))
        if identifier == "LP":
            return LPRegularizer(hyperparams.get("p", 2), hyperparams.get("lambda", 1e-5))
        if identifier.lower() == "l1":
            return LPRegularizer(1, 0.01)   # tf.keras.regularizers.L1 default
        if identifier.lower() == "l2":
            return LPRegularizer(2, 0.01)   # tf.keras.regularizers.L2 default
        if identifier.lower() == "l1_l2":
            return _l1l2(0.01, 0.01)        # tf.keras.regularizers.l1_l2 defaults
    raise ValueError(f"Could not interpret regularizer identifier: {identifier!r}")
