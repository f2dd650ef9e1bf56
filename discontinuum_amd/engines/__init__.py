from .base import BaseModel, DataMixin, ModelConfig, is_fitted  # noqa: F401
from .hip import MarginalHIP  # noqa: F401
