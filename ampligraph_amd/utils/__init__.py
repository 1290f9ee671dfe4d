"""Model persistence helpers with the reference's names (/root/reference/ampligraph/utils/model_utils.py:29-129)."""
from .model_utils import restore_model, save_model  # noqa: F401
