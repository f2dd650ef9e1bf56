"""The engine contract the model packages build on.

Interface parity with the reference (``src/discontinuum/engines/base.py:22-120``): a model is
``class X(SomeDataMixin, Engine)`` whose constructor passes ``model_config`` up and then calls
``build_datamanager``; engines implement ``fit / predict / build_model / build_datamanager``; public methods that
need a trained model are wrapped in ``@is_fitted``, which raises the reference's message.  What differs is the
wiring: the transform choice is a lookup on ``ModelConfig`` instead of an if / elif chain.
"""
from __future__ import annotations

import abc
import dataclasses
from functools import wraps

from .. import pipeline as _pl
from ..data_manager import DataManager

# target / error pipeline classes per ``ModelConfig.transform``
_TRANSFORMS = {
    "log": (_pl.LogStandardPipeline, _pl.LogErrorPipeline),
    "standard": (_pl.StandardPipeline, _pl.StandardErrorPipeline),
}
_NOT_FITTED = "The model hasn't been fitted yet, call .fit()."


@dataclasses.dataclass
class ModelConfig:
    """How the target is taken to model space: ``"log"`` (default) or ``"standard"``."""

    transform: str = "log"

    def target_and_error_pipelines(self):
        try:
            return _TRANSFORMS[self.transform]
        except KeyError:
            raise ValueError("Model config transform must be 'log' or 'standard'.") from None


def is_fitted(method):
    """Guard for methods that need a trained model: ``RuntimeError`` before ``fit`` has completed."""

    @wraps(method)
    def guarded(model, *args, **kwargs):
        if model.is_fitted:
            return method(model, *args, **kwargs)
        raise RuntimeError(_NOT_FITTED)

    return guarded


class BaseModel(abc.ABC):
    """State every engine starts from: the configuration it was given, no data manager yet, not fitted."""

    def __init__(self, model_config: dict | None = None):
        self.is_fitted = False
        self.dm = None
        self.model_config = {} if model_config is None else model_config

    @abc.abstractmethod
    def build_datamanager(self):
        """Create ``self.dm`` (a ``DataManager`` with this model's covariate pipelines)."""

    @abc.abstractmethod
    def build_model(self, X, y):
        """Return the probabilistic model for model-space inputs ``X`` and targets ``y``."""

    @abc.abstractmethod
    def fit(self, covariates, target, **kwargs):
        """Train on ``covariates`` / ``target`` (xarray objects); engines set ``is_fitted`` when they are done."""
        self.is_fitted = True
        return self

    @abc.abstractmethod
    def predict(self, covariates):
        """Predictions of a fitted model at new ``covariates``, in the original data space."""


class DataMixin:
    """``_build_datamanager(covariate_pipelines, model_config)`` for the model packages' ``build_datamanager``."""

    def _build_datamanager(self, covariate_pipelines: dict, model_config: ModelConfig | None = None):
        target_cls, error_cls = (model_config or ModelConfig()).target_and_error_pipelines()
        self.dm = DataManager(covariate_pipelines=covariate_pipelines, target_pipeline=target_cls,
                              error_pipeline=error_cls)
