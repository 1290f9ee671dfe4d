"""Mirror of ampligraph.latent_features for the hot path: the model class and the string->object
registries it is configured with."""
from . import loss_functions, optimizers, regularizers  # noqa: F401
from .loss_functions import (AbsoluteMarginLoss, NLLLoss, NLLMulticlass, PairwiseLoss,  # noqa: F401
                             SelfAdversarialLoss)
from .models import SCORING_LAYER_REGISTRY, ScoringBasedEmbeddingModel  # noqa: F401
