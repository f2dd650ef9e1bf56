"""``DataArray`` / ``Dataset``: xarray's when it is installed, otherwise a minimal labelled-array
stand-in with the handful of members the engine and the pipelines touch (``values``, ``data``,
``attrs``, ``name``, ``dims``, ``coords``, ``assign_coords``, ``__getitem__``, iteration).

The reference passes ``xarray`` objects across its engine boundary
(``src/discontinuum/engines/gpytorch.py:162-176, 461-501``); xarray is not part of this image, so the
stand-in keeps the boundary testable here while real xarray objects work unchanged where available.
"""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - depends on the environment
    from xarray import DataArray, Dataset  # type: ignore

    HAVE_XARRAY = True
except Exception:  # noqa: BLE001
    HAVE_XARRAY = False

    class DataArray:  # type: ignore[no-redef]
        def __init__(self, data, coords=None, dims=None, attrs=None, name=None):
            self.values = np.asarray(data)
            if dims is None:
                dims = tuple(f"dim_{i}" for i in range(self.values.ndim))
            self.dims = tuple(dims)
            self.attrs = dict(attrs or {})
            self.name = name
            self.coords = {}
            if coords is not None:
                items = coords.items() if hasattr(coords, "items") else zip(self.dims, coords)
                for k, v in items:
                    self.coords[k] = v if isinstance(v, DataArray) else _coord(k, v)

        @property
        def data(self):
            return self.values

        @property
        def shape(self):
            return self.values.shape

        @property
        def dtype(self):
            return self.values.dtype

        def assign_coords(self, coords):
            new = DataArray(self.values, dims=self.dims, attrs=self.attrs, name=self.name)
            new.coords = dict(self.coords)
            for k, v in dict(coords).items():
                new.coords[k] = v if isinstance(v, DataArray) else _coord(k, v)
            return new

        def __array__(self, dtype=None, copy=None):
            return self.values if dtype is None else self.values.astype(dtype)

        def __len__(self):
            return len(self.values)

        def min(self):
            return self.values.min()

        def max(self):
            return self.values.max()

        def __repr__(self):
            return f"DataArray(name={self.name!r}, dims={self.dims}, shape={self.values.shape})"

    def _coord(name, v):
        c = DataArray.__new__(DataArray)
        c.values = np.asarray(v.values if hasattr(v, "values") else v)
        c.dims, c.attrs, c.name, c.coords = (name,), {}, name, {}
        return c

    class Dataset:  # type: ignore[no-redef]
        def __init__(self, data_vars=None, coords=None, attrs=None):
            self.attrs = dict(attrs or {})
            self.coords = {k: (v if isinstance(v, DataArray) else _coord(k, v)) for k, v in dict(coords or {}).items()}
            self._vars = {}
            for k, v in dict(data_vars or {}).items():
                if isinstance(v, DataArray):
                    da = v
                elif isinstance(v, tuple):  # (dims, values[, attrs])
                    dims = (v[0],) if isinstance(v[0], str) else tuple(v[0])
                    da = DataArray(v[1], dims=dims, attrs=v[2] if len(v) > 2 else None, name=k)
                else:
                    da = DataArray(v, name=k)
                da.name = k
                da.coords = {c: self.coords[c] for c in da.dims if c in self.coords} or da.coords
                self._vars[k] = da

        def __getitem__(self, key):
            if key in self._vars:
                return self._vars[key]
            return self.coords[key]

        def __iter__(self):
            return iter(self._vars)

        def __contains__(self, key):
            return key in self._vars

        def __repr__(self):
            return f"Dataset(vars={list(self._vars)}, coords={list(self.coords)})"
