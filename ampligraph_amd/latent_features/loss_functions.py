"""Loss surface of the reference (/root/reference/ampligraph/latent_features/loss_functions.py):
same registry names, hyper-parameter keys, defaults and errors -- but a Loss object here only carries
parameters; the arithmetic (forward and hand-derived backward) runs inside the fused HIP training
kernel (ampligraph_amd/csrc/kge_train.hip)."""
from .. import _ffi

LOSS_REGISTRY = {}

DEFAULT_MARGIN = 1                  # loss_functions.py:23
DEFAULT_ALPHA_ADVERSARIAL = 0.5     # :26
DEFAULT_MARGIN_ADVERSARIAL = 3      # :29
DEFAULT_REDUCTION = "sum"


def register_loss(name):
    def deco(cls):
        LOSS_REGISTRY[name] = cls
        cls.name = name
        return cls
    return deco


class Loss:
    name = ""

    def __init__(self, hyperparam_dict=None, verbose=False):
        hyperparam_dict = dict(hyperparam_dict or {})
        self._loss_parameters = {"reduction": hyperparam_dict.get("reduction", DEFAULT_REDUCTION)}
        assert self._loss_parameters["reduction"] in ["sum", "mean"], "Invalid value for reduction!"
        self._init_hyperparams(hyperparam_dict)

    def _init_hyperparams(self, hyperparam_dict):
        pass

    def to_ffi(self):
        p = self._loss_parameters
        return _ffi.Loss(_ffi.LOSSES[self.name], 1 if p["reduction"] == "mean" else 0,
                         float(p.get("margin", 0.0)), float(p.get("alpha", 0.0)))


@register_loss("pairwise")
class PairwiseLoss(Loss):
    def _init_hyperparams(self, h):
        self._loss_parameters["margin"] = h.get("margin", DEFAULT_MARGIN)


@register_loss("nll")
class NLLLoss(Loss):
    pass


@register_loss("absolute_margin")
class AbsoluteMarginLoss(Loss):
    def _init_hyperparams(self, h):
        self._loss_parameters["margin"] = h.get("margin", DEFAULT_MARGIN)


@register_loss("self_adversarial")
class SelfAdversarialLoss(Loss):
    def _init_hyperparams(self, h):
        self._loss_parameters["margin"] = h.get("margin", DEFAULT_MARGIN_ADVERSARIAL)
        self._loss_parameters["alpha"] = h.get("alpha", DEFAULT_ALPHA_ADVERSARIAL)


@register_loss("multiclass_nll")
class NLLMulticlass(Loss):
    pass


def get(identifier, hyperparams=None):
    """loss_functions.py:720-766: Loss instance | registered name | callable."""
    if isinstance(identifier, Loss):
        return identifier
    if isinstance(identifier, str):
        if identifier not in LOSS_REGISTRY:
            raise ValueError("Could not interpret loss identifier:", identifier)
        return LOSS_REGISTRY[identifier](hyperparams or {})
    if callable(identifier):
        raise NotImplementedError(
            "user-defined Python loss callables cannot be fused into the HIP training kernel; "
            "use one of " + ", ".join(sorted(LOSS_REGISTRY)))
    raise ValueError("Could not interpret loss identifier:", identifier)
