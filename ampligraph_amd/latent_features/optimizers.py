"""Optimizer surface (/root/reference/ampligraph/latent_features/optimizers.py:255-291).  The reference
wraps a Keras *legacy* optimizer; here an Optimizer carries the hyper-parameters of the same update
rule and the dense sweep runs in HIP (ampligraph_amd/csrc/kge_opt.hip).  Supported: adam, adagrad, sgd
(the three the hot path names); defaults are Keras legacy's (epsilon 1e-7, Adagrad accumulator 0.1)."""
from .. import _ffi


class OptimizerWrapper:
    def __init__(self, name, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, **unused):
        name = name.lower()
        if name not in _ffi.OPTIMIZERS:
            raise ValueError("Could not interpret optimizer identifier: ", name)
        self.name = name
        self.learning_rate = float(learning_rate)
        self.beta_1, self.beta_2, self.epsilon = float(beta_1), float(beta_2), float(epsilon)
        self.iterations = 0

    def to_ffi(self, iteration, reg_p=2):
        return _ffi.Opt(_ffi.OPTIMIZERS[self.name], int(reg_p), self.learning_rate, self.beta_1, self.beta_2,
                        self.epsilon, 0.0, int(iteration))

    def get_config(self):
        return {"name": self.name, "learning_rate": self.learning_rate, "beta_1": self.beta_1,
                "beta_2": self.beta_2, "epsilon": self.epsilon}


def get(identifier, hyperparams=None):
    hyperparams = dict(hyperparams or {})
    if isinstance(identifier, OptimizerWrapper):
        return identifier
    if isinstance(identifier, str):
        lr = hyperparams.pop("learning_rate", 0.001)  # optimizers.py:284
        return OptimizerWrapper(identifier, learning_rate=lr, **hyperparams)
    # duck-typed Keras-like optimizer object: class name + learning_rate attribute
    name = type(identifier).__name__.lower()
    if name in _ffi.OPTIMIZERS and hasattr(identifier, "learning_rate"):
        cfg = {k: getattr(identifier, k) for k in ("beta_1", "beta_2", "epsilon") if hasattr(identifier, k)}
        return OptimizerWrapper(name, learning_rate=float(identifier.learning_rate), **cfg)
    raise ValueError("Could not interpret optimizer identifier: ", identifier)
