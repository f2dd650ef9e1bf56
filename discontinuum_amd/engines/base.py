"""Engine base classes -- same contract as ``src/discontinuum/engines/base.py:22-120``."""
from __future__ import annotations

import functools
from abc import ABC, abstractmethod
from dataclasses import dataclass

from ..data_manager import DataManager
from ..pipeline import LogErrorPipeline, LogStandardPipeline, StandardErrorPipeline, StandardPipeline


@dataclass
class ModelConfig:
    """Configuration for model data transformations."""

    transform: str = "log"  # "log" | "standard"


class BaseModel(ABC):
    def __init__(self, model_config: dict | None = None):
        if model_config is None:
            model_config = {}
        self.model_config = model_config
        self.dm = None
        self.is_fitted = False

    @abstractmethod
    def fit(self, covariates, target, **kwargs):
        """Fit model to data."""
        self.is_fitted = True
        return self

    @abstractmethod
    def predict(self, covariates):
        """Use a fitted model to make predictions on new data."""

    @abstractmethod
    def build_model(self, X, y):
        pass

    @abstractmethod
    def build_datamanager(self):
        """Build DataManager for the model."""


class DataMixin:
    """Shared logic for building a DataManager with log/standard transforms."""

    def _build_datamanager(self, covariate_pipelines: dict, model_config: ModelConfig | None = None):
        if model_config is None:
            model_config = ModelConfig()
        if model_config.transform == "log":
            target_pipeline, error_pipeline = LogStandardPipeline, LogErrorPipeline
        elif model_config.transform == "standard":
            target_pipeline, error_pipeline = StandardPipeline, StandardErrorPipeline
        else:
            raise ValueError("Model config transform must be 'log' or 'standard'.")
        self.dm = DataManager(
            target_pipeline=target_pipeline,
            error_pipeline=error_pipeline,
            covariate_pipelines=covariate_pipelines,
        )


def is_fitted(func):
    """Decorator checks whether model has been fit."""

    @functools.wraps(func)
    def inner(self, *args, **kwargs):
        if not self.is_fitted:
            raise RuntimeError("The model hasn't been fitted yet, call .fit().")
        return func(self, *args, **kwargs)

    return inner
