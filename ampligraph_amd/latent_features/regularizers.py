"""Regulariser surface (/root/reference/ampligraph/latent_features/regularizers.py:14-73): LP with
keys `p` and `lambda` (defaults 2, 1e-5), the alias 'l3', and the Keras names 'l1'/'l2' (factor 0.01).
The penalty lambda*sum|x|^p over the WHOLE table and its dense gradient are fused into the HIP
optimizer sweep."""


class LPRegularizer:
    def __init__(self, p=2, lam=1e-5):
        if int(p) < 1:
            raise ValueError("LP regularizer needs p >= 1")
        self.p, self.lam = int(p), float(lam)


def get(identifier, hyperparams=None):
    hyperparams = dict(hyperparams or {})
    if identifier is None:
        return None
    if isinstance(identifier, LPRegularizer):
        return identifier
    if isinstance(identifier, str):
        if identifier == "l3":
            return LPRegularizer(3, hyperparams.get("lambda", 1e-5))
        if identifier == "LP":
            return LPRegularizer(hyperparams.get("p", 2), hyperparams.get("lambda", 1e-5))
        if identifier.lower() == "l1":
            return LPRegularizer(1, 0.01)   # tf.keras.regularizers.L1 default
        if identifier.lower() == "l2":
            return LPRegularizer(2, 0.01)   # tf.keras.regularizers.L2 default
    raise ValueError(f"Could not interpret regularizer identifier: {identifier!r}")
