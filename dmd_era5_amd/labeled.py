"""Minimal labelled arrays for the host side of the path.

The reference passes ``xarray`` objects between its stages; xarray is not available
in this image, and the engine only ever needs four things from them: the values, the
dimension names, 1-D coordinate arrays and an attribute dict.  ``DataArray`` /
``Dataset`` below carry exactly that, with the attribute names xarray uses
(``.values .dims .coords .attrs .sizes .data_vars``), so the mirrored functions read
like the reference's.  Any object with ``.values`` (a real ``xr.DataArray`` included)
is accepted at the SVD boundary.

The ``space`` coordinate: the reference stores one Python tuple ``(level, lat, lon)``
per row (slice_tools.py:323,346); at 10^6-10^7 rows that is minutes of interpreter
time, so here it is an ``(m, 3)`` float64 array with the same three numbers per row.
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np


class Coord:
    """A coordinate: values indexed by one named dimension (or several for `space`)."""

    __slots__ = ("dims", "values")

    def __init__(self, dims, values):
        self.dims = (dims,) if isinstance(dims, str) else tuple(dims)
        self.values = np.asarray(values)

    @property
    def data(self):
        return self.values

    @property
    def shape(self):
        return self.values.shape

    def __len__(self):
        return len(self.values)

    def __getitem__(self, idx):
        return Coord(self.dims, self.values[idx])


def _as_coords(coords) -> "OrderedDict[str, Coord]":
    out: OrderedDict[str, Coord] = OrderedDict()
    for name, c in (coords or {}).items():
        if isinstance(c, Coord):
            out[name] = c
        elif isinstance(c, tuple) and len(c) == 2 and isinstance(c[0], (str, tuple, list)):
            out[name] = Coord(c[0], c[1])
        else:
            out[name] = Coord(name, c)
    return out


class LazyArray:
    """A file-backed array (like xarray's lazily opened variables): shape / dtype are known,
    the data are read on first use, or in slabs along the first axis by the streaming ingest."""

    def __init__(self, shape, dtype, read_all, read_slab, read_box=None):
        self.shape, self.dtype, self.ndim = tuple(shape), np.dtype(dtype), len(shape)
        self._read_all, self.read_slab = read_all, read_slab
        # read_box(starts, counts, out=None): a hyperslab (one rank's latitude band of a time slab)
        self.read_box = read_box or self._box_from_slab

    def _box_from_slab(self, starts, counts, out=None):
        slab = self.read_slab(starts[0], starts[0] + counts[0])
        box = slab[(slice(None),) + tuple(slice(a, a + c) for a, c in zip(starts[1:], counts[1:]))]
        if out is None:
            return np.ascontiguousarray(box)
        out[...] = box
        return out

    def __array__(self, dtype=None, copy=None):
        a = self._read_all()
        return a.astype(dtype) if dtype is not None else a


class DataArray:
    def __init__(self, values, dims, coords=None, attrs=None, name=None):
        self._values = values if hasattr(values, "shape") else np.asarray(values)
        self.dims = tuple(dims)
        if len(self.dims) != self._values.ndim:
            raise ValueError(f"dims {self.dims} do not match a {self._values.ndim}-D array")
        self.coords = _as_coords(coords)
        self.attrs = dict(attrs or {})
        self.name = name

    @property
    def values(self):
        if isinstance(self._values, LazyArray):
            self._values = np.asarray(self._values)
        return self._values

    @values.setter
    def values(self, v):
        self._values = v

    @property
    def lazy(self) -> "LazyArray | None":
        """The file-backed array if the data have not been loaded yet, else None."""
        return self._values if isinstance(self._values, LazyArray) else None

    @property
    def shape(self):
        return tuple(self._values.shape)

    @property
    def ndim(self):
        return self._values.ndim

    @property
    def dtype(self):
        return self._values.dtype

    @property
    def sizes(self):
        return dict(zip(self.dims, self._values.shape))

    def __repr__(self):
        return f"<DataArray {self.name or ''} {self.sizes} coords={list(self.coords)}>"


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.data_vars: OrderedDict[str, DataArray] = OrderedDict()
        self.coords = _as_coords(coords)
        self.attrs = dict(attrs or {})
        for name, da in (data_vars or {}).items():
            self[name] = da

    def __setitem__(self, name, da):
        if not isinstance(da, DataArray):
            raise TypeError("Dataset variables must be DataArray")
        da.name = name
        self.data_vars[name] = da
        for cn, c in da.coords.items():
            self.coords.setdefault(cn, c)

    def __getitem__(self, key):
        if isinstance(key, (list, tuple)):
            missing = [k for k in key if k not in self.data_vars]
            if missing:
                raise KeyError(f"variables not in dataset: {missing}")
            return Dataset({k: self.data_vars[k] for k in key}, self.coords, self.attrs)
        return self.data_vars[key]

    def __contains__(self, key):
        return key in self.data_vars

    @property
    def sizes(self):
        out = {}
        for da in self.data_vars.values():
            out.update(da.sizes)
        for name, c in self.coords.items():
            if c.dims == (name,):
                out.setdefault(name, len(c))
        return out

    def __repr__(self):
        return f"<Dataset vars={list(self.data_vars)} sizes={self.sizes}>"
