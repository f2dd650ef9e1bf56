"""Data transformations between raw and model space.

Behaviour follows ``src/discontinuum/pipeline.py:14-403``: the same named pipelines with the same
step order, clips and scalers, written on plain numpy (no sklearn ``Pipeline`` dependency) and on the
``xr_compat`` labelled arrays.  Known answer pinned by the reference's own test
(``src/discontinuum/tests/test_pipeline.py:14``) is checked in ``tests/test_pipeline.py``.
"""
from __future__ import annotations

import threading

import numpy as np
import torch
import pandas as pd
from scipy.stats import norm

from .xr_compat import DataArray


def datetime_to_decimal_year(x):
    """datetime64 array -> year + (julian day - julian day of 1 Jan) / days in year."""
    x = np.asarray(x)
    if not np.issubdtype(x.dtype, np.datetime64):
        raise ValueError("Array must contain numpy datetime64 objects.")
    # datetime64 arithmetic only (a pandas round trip costs ~2 ms per record, which is most of the set-up time of a
    # many-site fit): elapsed time since 1 January over the length of that year
    stamp = x.reshape(-1).astype("datetime64[ns]")
    year = stamp.astype("datetime64[Y]")
    jan1, next_jan1 = year.astype("datetime64[ns]"), (year + 1).astype("datetime64[ns]")
    frac = (stamp - jan1).astype(np.float64) / (next_jan1 - jan1).astype(np.float64)
    return year.astype(np.int64) + 1970 + frac


def decimal_year_to_datetime(x):
    x = np.asarray(x, dtype=float)
    whole = np.floor(x).astype(int)
    jan1 = pd.to_datetime(whole, format="%Y")
    length = 365 + jan1.is_leap_year
    stamp = jan1 + pd.to_timedelta((x - whole) * length, unit="D")
    return stamp.round("1s").to_numpy()


class _Step:
    def fit(self, X):
        return self

    def transform(self, X):
        return X

    def inverse_transform(self, X):
        return X

    # ``inverse_device``: the same inverse on a torch tensor that lives on the GPU (``sample()`` maps 10^7 draws back to
    # data space; doing that where they were drawn halves its wall time).  Steps without one make the caller use numpy.


class MetadataManager(_Step):
    """Strip / restore the labelled-array wrapper (attrs, name, dims)."""

    def fit(self, X):
        self.attrs_, self.name_, self.dims_ = X.attrs, X.name, X.dims
        return self

    def transform(self, X):
        return np.asarray(X.values).reshape(-1, 1)

    def inverse_transform(self, X):
        return DataArray(np.squeeze(X), attrs=self.attrs_, name=self.name_, dims=self.dims_)

    def inverse_device(self, t):
        return self.inverse_transform(_to_host(t))


class ClipTransformer(_Step):
    def __init__(self, min=None, max=None):  # noqa: A002
        self.min, self.max = min, max

    def transform(self, X):
        return np.clip(X, a_min=self.min, a_max=self.max)

    inverse_transform = transform

    def inverse_device(self, t):
        return t if self.min is None and self.max is None else t.clamp_(min=self.min, max=self.max)


class LogTransformer(_Step):
    def transform(self, X):
        return np.log(X)

    def inverse_transform(self, X):
        return np.exp(X)

    def inverse_device(self, t):
        return t.exp_()


class SquareTransformer(_Step):
    def transform(self, X):
        return X ** 2

    def inverse_transform(self, X):
        return np.sqrt(X)

    def inverse_device(self, t):
        return t.sqrt_()


class UnitScaler(_Step):
    def __init__(self, zero_value=0):
        self.zero = zero_value

    def fit(self, X):
        self.min_, self.max_ = X.min(), X.max()
        return self

    def transform(self, X):
        return self.zero + (X - self.min_) / (self.max_ - self.min_)

    def inverse_transform(self, X):
        return self.min_ + (X - self.zero) * (self.max_ - self.min_)

    def inverse_device(self, t):
        return t.sub_(float(self.zero)).mul_(float(self.max_ - self.min_)).add_(float(self.min_))


class StandardScaler(_Step):
    def __init__(self, with_mean=True, with_std=True):
        self.with_mean, self.with_std = with_mean, with_std

    def fit(self, X):
        if self.with_mean:
            self.mean_ = X.mean(axis=0)
        if self.with_std:
            self.scale_ = X.std(axis=0)
        return self

    def transform(self, X):
        if self.with_mean:
            X = X - self.mean_
        if self.with_std:
            X = X / self.scale_
        return X

    def inverse_transform(self, X):
        if self.with_std:
            X = X * self.scale_
        if self.with_mean:
            X = X + self.mean_
        return X

    def inverse_device(self, t):
        if self.with_std:
            t = t.mul_(_like(self.scale_, t))
        if self.with_mean:
            t = t.add_(_like(self.mean_, t))
        return t


class TimeTransformer(_Step):
    def transform(self, X):
        return datetime_to_decimal_year(X)

    def inverse_transform(self, X):
        return decimal_year_to_datetime(X)


def _like(value, t):
    return torch.as_tensor(np.asarray(value, dtype=np.float64).reshape(-1), dtype=t.dtype, device=t.device)


_PINNED = threading.local()  # one reused buffer PER THREAD: concurrent sample() calls must not share it


def _to_host(t):
    """Device tensor -> numpy through a reused page-locked buffer (a pageable copy of 90 MB takes 2-3x longer)."""
    if not t.is_cuda:
        return t.numpy()
    key = (t.dtype, t.numel())
    if getattr(_PINNED, "key", None) != key:
        _PINNED.key, _PINNED.buf = key, torch.empty(t.numel(), dtype=t.dtype, pin_memory=True)
    host = _PINNED.buf
    host.copy_(t.reshape(-1))
    return host.numpy().reshape(tuple(t.shape)).copy()


class Pipeline:
    """fit() threads the data through the steps; inverse_transform() walks them backwards."""

    def __init__(self, steps):
        self.steps = list(steps)

    def fit(self, X):
        for _name, step in self.steps:
            X = step.fit(X).transform(X)
        return self

    def transform(self, X):
        for _name, step in self.steps:
            X = step.transform(X)
        return X

    def inverse_transform(self, X):
        for _name, step in reversed(self.steps):
            X = step.inverse_transform(X)
        return X

    def inverse_transform_device(self, t):
        """``inverse_transform`` of a torch tensor, evaluated on its device (in place) and brought to the host at the
        metadata step; ``None`` if a step has no device form."""
        if not all(hasattr(step, "inverse_device") for _name, step in self.steps):
            return None
        for _name, step in reversed(self.steps):
            t = step.inverse_device(t)
        return t


def _recipe(doc, *steps, **methods):
    """A named pipeline class from its recipe: ``(step name, step class, constructor kwargs)`` triples.  Every instance
    builds fresh step objects, so fitted state is never shared between pipelines."""

    def __init__(self):
        Pipeline.__init__(self, [(name, cls(**kwargs)) for name, cls, kwargs in steps])

    return type("_", (Pipeline,), {"__init__": __init__, "__doc__": doc, **methods})


def _named(cls, name):
    cls.__name__ = cls.__qualname__ = name
    return cls


_META = ("metadata", MetadataManager, {})
_NONNEG = ("clip", ClipTransformer, {"min": 0})

# The reference's pipelines, step for step (src/discontinuum/pipeline.py:242-403): name -> recipe.
LogStandardPipeline = _named(_recipe(
    "Positive, log-normally distributed data: clip at 1e-6, log, standardise.",
    _META, ("clip", ClipTransformer, {"min": 1e-6}), ("log", LogTransformer, {}), ("scaler", StandardScaler, {})),
    "LogStandardPipeline")
NoOpPipeline = _named(_recipe("Non-negative data passed through unchanged.", _META, _NONNEG), "NoOpPipeline")
StandardPipeline = _named(_recipe(
    "Non-negative, roughly normal data: standardise.", _META, _NONNEG, ("scaler", StandardScaler, {})),
    "StandardPipeline")
UnitPipeline = _named(_recipe(
    "Non-negative data rescaled to [1, 2] (minimum -> 1, maximum -> 2).",
    _META, _NONNEG, ("scaler", UnitScaler, {"zero_value": 1})), "UnitPipeline")
TimePipeline = _named(_recipe(
    "Timestamps -> centred decimal years (mean removed, scale kept).",
    _META, ("decimal_year", TimeTransformer, {}), ("scaler", StandardScaler, {"with_std": False})), "TimePipeline")


def _zscore(ci):
    return norm.ppf(1 - (1 - ci) / 2)


def _additive_ci(self, mean, se, ci=0.95):
    half = se * _zscore(ci)
    return mean - half, mean + half


def _multiplicative_ci(self, mean, se, ci=0.95):
    factor = se ** _zscore(ci)
    return mean / factor, mean * factor


# error pipelines: ``inverse_transform`` takes a model-space VARIANCE to a standard error in data space
StandardErrorPipeline = _named(_recipe(
    "Variance of standardised data -> standard error; ``ci`` is mean -/+ z se.",
    _META, ("scaler", StandardScaler, {"with_mean": False}), ("square", SquareTransformer, {}), _NONNEG,
    ci=_additive_ci), "StandardErrorPipeline")
LogErrorPipeline = _named(_recipe(
    "Variance of log-standardised data -> geometric standard error; ``ci`` is mean / se^z .. mean * se^z.",
    _META, ("log", LogTransformer, {}), ("scaler", StandardScaler, {"with_mean": False}), ("square", SquareTransformer, {}),
    ("clip", ClipTransformer, {"min": 1e-6}), ci=_multiplicative_ci), "LogErrorPipeline")
