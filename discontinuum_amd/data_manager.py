"""Raw (labelled arrays) <-> model space ((n, d) design matrix, targets, variances).

Same contract as ``src/discontinuum/data_manager.py:30-120``; the covariate order of
``covariate_pipelines`` defines the column order of X (time first), which the kernels rely on.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .pipeline import LogErrorPipeline, LogStandardPipeline
from .xr_compat import Dataset


@dataclass
class Data:
    target: object
    covariates: object
    target_unc: object = None


class DataManager:
    def __init__(self, target_pipeline=LogStandardPipeline, error_pipeline=LogErrorPipeline, covariate_pipelines=None):
        self.target_pipeline = target_pipeline
        self.error_pipeline = error_pipeline
        self.covariate_pipelines = covariate_pipelines
        self._cache = {}

    def fit(self, target, covariates, target_unc=None):
        self.data = Data(target, covariates, target_unc)
        self._cache.clear()
        # pipelines given as classes are instantiated and fitted once; fitted instances are kept
        if isinstance(self.target_pipeline, type):
            self.target_pipeline = self.target_pipeline().fit(target)
        if isinstance(self.error_pipeline, type):
            self.error_pipeline = self.error_pipeline().fit(target)
        for key, pipe in self.covariate_pipelines.items():
            if isinstance(pipe, type):
                self.covariate_pipelines[key] = pipe().fit(covariates[key])

    def transform_covariates(self, covariates):
        shape = tuple(s for c in covariates.coords for s in np.shape(covariates.coords[c]))
        X = np.empty(shape + (len(self.covariate_pipelines),))
        for col, (key, pipe) in enumerate(self.covariate_pipelines.items()):
            X[..., col] = np.asarray(pipe.transform(covariates[key])).reshape(-1)
        return X

    def inverse_transform_covariates(self, X):
        return Dataset({key: pipe.inverse_transform(X[:, col])
                        for col, (key, pipe) in enumerate(self.covariate_pipelines.items())})

    def _cached(self, name, fn):
        if name not in self._cache:
            self._cache[name] = fn()
        return self._cache[name]

    @property
    def y(self):
        return self._cached("y", lambda: np.asarray(self.target_pipeline.transform(self.data.target)).reshape(-1))

    @property
    def y_unc(self):
        return self._cached("y_unc", lambda: np.asarray(self.error_pipeline.transform(self.data.target_unc)).reshape(-1))

    @property
    def X(self):
        return self._cached("X", lambda: self.transform_covariates(self.data.covariates))

    def Xnew(self, ds):
        return self.transform_covariates(ds)

    def y_t(self, y):
        return self.target_pipeline.inverse_transform(y)

    def get_dim(self, dim: str) -> int:
        names = list(self.data.covariates.coords) + list(self.data.covariates)
        return names.index(dim)
