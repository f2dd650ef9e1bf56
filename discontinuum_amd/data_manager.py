"""Between labelled raw data and the arrays the GP engine works on.

Public surface as in the reference (``src/discontinuum/data_manager.py:30-120``): ``DataManager(target_pipeline,
error_pipeline, covariate_pipelines)`` with ``fit / X / y / y_unc / Xnew / y_t / transform_covariates /
inverse_transform_covariates / get_dim`` and the ``data`` record.  The ORDER of ``covariate_pipelines`` is the column
order of the design matrix (time first) -- the device kernels rely on it.

Implementation notes: pipelines may arrive as classes (they are instantiated and fitted on the first ``fit``) or as
already fitted objects (kept as they are, which is how a checkpointed model is re-attached to its data); the
model-space arrays are computed eagerly in ``fit`` rather than on first access.
"""
from __future__ import annotations

import numpy as np

from . import pipeline as _pl
from .xr_compat import Dataset


class Data:
    """The raw record a manager was fitted on."""

    __slots__ = ("target", "covariates", "target_unc")

    def __init__(self, target, covariates, target_unc=None):
        self.target, self.covariates, self.target_unc = target, covariates, target_unc


def _fitted(pipe, sample):
    """A pipeline object ready to transform: classes are instantiated and fitted on ``sample``."""
    return pipe().fit(sample) if isinstance(pipe, type) else pipe


def _flat(values):
    return np.asarray(values).reshape(-1)


class DataManager:
    def __init__(self, target_pipeline=_pl.LogStandardPipeline, error_pipeline=_pl.LogErrorPipeline,
                 covariate_pipelines=None):
        self.target_pipeline = target_pipeline
        self.error_pipeline = error_pipeline
        self.covariate_pipelines = covariate_pipelines
        self.data = None
        self._model_space = {}

    # ------------------------------------------------------------------ fitting
    def fit(self, target, covariates, target_unc=None):
        self.data = Data(target, covariates, target_unc)
        self.target_pipeline = _fitted(self.target_pipeline, target)
        self.error_pipeline = _fitted(self.error_pipeline, target)
        for name in list(self.covariate_pipelines):
            self.covariate_pipelines[name] = _fitted(self.covariate_pipelines[name], covariates[name])
        space = {"X": self.transform_covariates(covariates), "y": _flat(self.target_pipeline.transform(target))}
        if target_unc is not None:
            space["y_unc"] = _flat(self.error_pipeline.transform(target_unc))
        self._model_space = space

    # ------------------------------------------------------------------ model-space views of the fitted record
    @property
    def X(self):
        return self._model_space["X"]

    @property
    def y(self):
        return self._model_space["y"]

    @property
    def y_unc(self):
        if "y_unc" not in self._model_space:  # same failure the reference gives when no uncertainty was supplied
            self._model_space["y_unc"] = _flat(self.error_pipeline.transform(self.data.target_unc))
        return self._model_space["y_unc"]

    # ------------------------------------------------------------------ transforms
    def transform_covariates(self, covariates):
        """(..., d) design matrix; the leading shape is that of the coordinates (a record or a grid)."""
        lead = ()
        for coord in covariates.coords:
            lead += tuple(np.shape(covariates.coords[coord]))
        columns = [_flat(pipe.transform(covariates[name])) for name, pipe in self.covariate_pipelines.items()]
        return np.stack(columns, axis=-1).reshape(lead + (len(columns),))

    def inverse_transform_covariates(self, X):
        raw = {}
        for column, (name, pipe) in enumerate(self.covariate_pipelines.items()):
            raw[name] = pipe.inverse_transform(X[:, column])
        return Dataset(raw)

    def Xnew(self, ds):
        return self.transform_covariates(ds)

    def y_t(self, y):
        return self.target_pipeline.inverse_transform(y)

    def y_t_device(self, t):
        """``y_t`` of a torch tensor: elementwise steps run where the tensor lives, then one copy to the host."""
        device_form = getattr(self.target_pipeline, "inverse_transform_device", None)
        out = device_form(t) if device_form is not None else None
        return out if out is not None else self.y_t(t.cpu().numpy())

    def get_dim(self, dim: str) -> int:
        """Column of ``dim`` in the design matrix (coordinates first, then data variables)."""
        order = [*self.data.covariates.coords, *self.data.covariates]
        return order.index(dim)
