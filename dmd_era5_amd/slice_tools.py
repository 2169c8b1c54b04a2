"""Pre-processing of an ERA5 slice (host side, numpy) -- mirror of the reference's
src/dmd_era5/slice_tools/slice_tools.py on the light labelled types of
:mod:`dmd_era5_amd.labeled`.  Function names, argument meaning, error types and
messages follow the reference; each docstring cites the lines it mirrors.

These define the snapshot matrix X exactly (row order, embedding order, labels).
The big-data path of ``era5_svd.main`` does the same arithmetic on the device
(K5 centre/scale, zero-copy embedding); the functions here are the reference
semantics, used for small inputs, labels and tests.
"""
from __future__ import annotations

from datetime import datetime, timedelta

import numpy as np

from .labeled import Coord, DataArray, Dataset
from .logger import log_and_print, setup_logger

logger = setup_logger("ERA5Processing", "era5_processing.log")

SPATIAL_STACK_ORDER = ["level", "latitude", "longitude"]


def _to_datetime(t64) -> datetime:
    return np.datetime64(t64, "us").astype(datetime)


def _get_dataset_time_bounds(ds: Dataset) -> dict:
    """First / last timestamp as naive datetimes (slice_tools.py:106-123)."""
    t = ds.coords["time"].values
    return {"first": _to_datetime(t[0]), "last": _to_datetime(t[-1])}


def _isel(ds: Dataset, dim: str, index) -> Dataset:
    """Positional selection along one dimension for every variable and coordinate."""
    coords = {}
    for name, c in ds.coords.items():
        if dim in c.dims:
            ax = c.dims.index(dim)
            coords[name] = Coord(c.dims, c.values[(slice(None),) * ax + (index,)])
        else:
            coords[name] = c
    out = Dataset(coords=coords, attrs=ds.attrs)
    for name, da in ds.data_vars.items():
        vals = da.values
        if dim in da.dims:
            ax = da.dims.index(dim)
            vals = vals[(slice(None),) * ax + (index,)]
        out[name] = DataArray(vals, da.dims, {k: coords[k] for k in da.coords if k in coords}, da.attrs)
    return out


def slice_era5_dataset(ds: Dataset, start_datetime=None, end_datetime=None, levels=None) -> Dataset:
    """Time-range and level selection (slice_tools.py:20-103): inclusive time slice,
    levels returned in the requested order; ValueError when the range leaves the data,
    start >= end, or a level is missing."""
    start = datetime.fromisoformat(start_datetime) if isinstance(start_datetime, str) else start_datetime
    end = datetime.fromisoformat(end_datetime) if isinstance(end_datetime, str) else end_datetime
    bounds = _get_dataset_time_bounds(ds)
    start = start or bounds["first"]
    end = end or bounds["last"]
    if start < bounds["first"] or end > bounds["last"]:
        msg = (f"Time range ({start} to {end}) is outside dataset"
               f"bounds ({bounds['first']} to {bounds['last']}).")
        log_and_print(logger, msg, "error")
        raise ValueError(msg)
    if start >= end:
        msg = "Start datetime must be before end datetime."
        log_and_print(logger, msg, "error")
        raise ValueError(msg)
    have = list(ds.coords["level"].values)
    levels = levels or have
    missing = [lv for lv in levels if lv not in have]
    if missing:
        msg = f"Requested level is not available in the dataset.Available levels: {have}"
        log_and_print(logger, msg, "error")
        raise ValueError(msg)
    t = ds.coords["time"].values
    keep = np.nonzero((t >= np.datetime64(start)) & (t <= np.datetime64(end)))[0]
    out = _isel(ds, "time", keep if len(keep) != len(t) else slice(None))
    out = _isel(out, "level", np.array([have.index(lv) for lv in levels]))
    log_and_print(logger, f"Dataset slicing completed successfully using {start}to {end} and levels {levels}")
    return out


def nearest_resample_index(times: np.ndarray, delta_time: timedelta, _force_pandas: bool = False):
    """Bin labels and nearest-sample indices of ``ds.resample(time=dt).nearest()``
    (slice_tools.py:139).  xarray delegates to pandas: labels are the resample bins
    (anchored at midnight of the first day), each filled with the nearest original
    sample -- so pandas itself computes both here.  The common case needs no pandas (whose
    import costs ~0.9 s of a 3 s `main()` at cfg2 scale): samples spaced by exactly
    ``delta_time``, a divisor of one day, and sitting on the bins (first sample a whole number
    of steps after midnight) are their own labels -- checked against pandas in the tests."""
    if not _force_pandas:
        t = np.asarray(times).astype("datetime64[ns]").astype(np.int64)
        dt = int(round(delta_time.total_seconds() * 1e9))
        day = 86_400_000_000_000
        if t.size and 0 < dt <= day and day % dt == 0 and (t[0] % day) % dt == 0 and np.all(np.diff(t) == dt):
            return t.astype("datetime64[ns]"), np.arange(t.size)
    import pandas as pd

    idx = pd.DatetimeIndex(times)
    labels = pd.Series(0, index=idx).resample(delta_time).first().index
    take = idx.get_indexer(labels, method="nearest")
    return labels.values.astype("datetime64[ns]"), take


def resample_era5_dataset(ds: Dataset, delta_time: timedelta) -> Dataset:
    """Nearest-neighbour resampling along time (slice_tools.py:126-141)."""
    labels, take = nearest_resample_index(ds.coords["time"].values, delta_time)
    identity = len(take) == len(ds.coords["time"]) and np.array_equal(take, np.arange(len(take)))
    out = ds if identity else _isel(ds, "time", take)
    if not identity:
        out.coords["time"] = Coord("time", labels)
        for da in out.data_vars.values():
            da.coords["time"] = out.coords["time"]
    log_and_print(logger, f"Resampled the dataset with time delta: {delta_time}")
    return out


def standardize_data(data: Dataset, dim: str = "time", scale: bool = True):
    """Mean-centre and optionally scale to unit variance along ``dim``
    (slice_tools.py:144-179): mean; data - mean; std (ddof 0) of the centred data;
    data / std.  Returns (data, mean, std or None) as Datasets."""
    log_and_print(logger, f"Standardizing data along {dim} dimension...")
    cen, mean, std = Dataset(attrs=data.attrs), Dataset(attrs=data.attrs), Dataset(attrs=data.attrs)
    for name, da in data.data_vars.items():
        ax = da.dims.index(dim)
        rdims = tuple(d for d in da.dims if d != dim)
        rcoords = {k: c for k, c in da.coords.items() if dim not in c.dims}
        mu = da.values.mean(axis=ax, keepdims=True, dtype=da.values.dtype)
        c = da.values - mu
        mean[name] = DataArray(np.squeeze(mu, axis=ax), rdims, rcoords)
        if scale:
            sd = c.std(axis=ax, keepdims=True, dtype=c.dtype)
            c = c / sd
            std[name] = DataArray(np.squeeze(sd, axis=ax), rdims, rcoords)
        cen[name] = DataArray(c, da.dims, da.coords, da.attrs)
    return cen, mean, (std if scale else None)


def _apply_delay_embedding_np(X: np.ndarray, d: int) -> np.ndarray:
    """(n_samples*d, n_time-d+1) delay embedding, F-ordered like the reference's
    (slice_tools.py:182-211): row k*m + s, column t  ==  X[s, t+k]."""
    if X.ndim != 2:
        raise ValueError("Input array must be 2D.")
    if not isinstance(d, int) or isinstance(d, bool) or d <= 0:
        raise ValueError("Delay must be an integer greater than 0.")
    m, n = X.shape
    nt = n - d + 1
    out = np.empty((d * m, nt), dtype=X.dtype, order="F")
    for k in range(d):
        out[k * m:(k + 1) * m] = X[:, k:k + nt]
    return out


def delay_coords(space: np.ndarray, original_variable: np.ndarray, time: np.ndarray, d: int) -> dict:
    """Coordinates of the embedded array (slice_tools.py:258-269): space / variable
    labels tiled d times, time = time[d-1:], delay = repeat(flip(arange(d)), m)."""
    m = len(original_variable)
    reps = (d,) + (1,) * (space.ndim - 1)
    return {
        "space": Coord("space", np.tile(space, reps)),
        "time": Coord("time", time[d - 1:]),
        "original_variable": Coord("space", np.tile(original_variable, d)),
        "delay": Coord("space", np.repeat(np.flip(np.arange(d)), m)),
    }


def apply_delay_embedding(X: DataArray, d: int) -> DataArray:
    """Delay-embed a (space, time) DataArray (slice_tools.py:214-274)."""
    if not isinstance(X, DataArray):
        raise ValueError("Input data must be a xr.DataArray")
    if sorted(X.dims) != ["space", "time"]:
        raise ValueError("Input data must have dimensions ('space', 'time').")
    if sorted(X.coords) != ["original_variable", "space", "time"]:
        raise ValueError("Input data must have coordinates ('space', 'time', 'original_variable').")
    out = DataArray(
        _apply_delay_embedding_np(X.values, d), ("space", "time"),
        delay_coords(X.coords["space"].values, X.coords["original_variable"].values,
                     X.coords["time"].values, d),
        X.attrs)
    out.attrs["delay_embedding"] = d
    return out


def space_labels(levels, lats, lons) -> np.ndarray:
    """(level, lat, lon) of every row of one variable, level slowest, lon fastest."""
    L, A, O = np.meshgrid(np.asarray(levels, dtype=np.float64), np.asarray(lats, dtype=np.float64),
                          np.asarray(lons, dtype=np.float64), indexing="ij")
    return np.stack([L.ravel(), A.ravel(), O.ravel()], axis=1)


def flatten_era5_variables(era5_ds: Dataset) -> DataArray:
    """Stack (level, latitude, longitude) into ``space`` and concatenate the variables
    along it (slice_tools.py:277-365): row v*m_v + ((l*n_lat + i)*n_lon + j)."""
    names = list(era5_ds.data_vars)
    coords = sorted(era5_ds.coords)
    has_time = coords == sorted(SPATIAL_STACK_ORDER + ["time"])
    if not has_time and coords != sorted(SPATIAL_STACK_ORDER):
        raise ValueError("Input dataset must have coordinates ('latitude', 'longitude', 'level') "
                         "or ('latitude', 'longitude', 'level', 'time').")
    mats = []
    for name in names:
        da = era5_ds[name]
        order = [da.dims.index(dn) for dn in SPATIAL_STACK_ORDER]
        if has_time:
            v = np.transpose(da.values, order + [da.dims.index("time")])
            mats.append(v.reshape(-1, v.shape[-1]))
        else:
            mats.append(np.transpose(da.values, order).reshape(-1))
    data = np.concatenate(mats, axis=0)
    one = space_labels(*(era5_ds.coords[c].values for c in SPATIAL_STACK_ORDER))
    cds = {
        "space": Coord("space", np.tile(one, (len(names), 1))),
        "original_variable": Coord("space", np.repeat(names, one.shape[0])),
    }
    dims = ("space",)
    if has_time:
        cds["time"] = era5_ds.coords["time"]
        dims = ("space", "time")
    out = DataArray(data, dims, cds, era5_ds.attrs)
    out.attrs["original_variables"] = names
    out.attrs["space_coords"] = list(SPATIAL_STACK_ORDER)
    return out


def space_coord_to_level_lat_lon(ds: Dataset) -> Dataset:
    """Replace the (level, lat, lon) ``space`` coordinate by an integer range plus three
    per-row coordinates, which is what can be stored in NetCDF (slice_tools.py:368-414)."""
    if "space" not in ds.coords:
        raise ValueError("Input dataset must have a 'space' coordinate.")
    if all(c in ds.coords for c in SPATIAL_STACK_ORDER):
        log_and_print(logger, "Dataset already has separate coordinates for level, latitude, and longitude.")
        return ds
    sp = np.asarray(ds.coords["space"].values)
    if sp.dtype == object:  # tuples, as the reference builds them
        sp = np.array([tuple(x) for x in sp], dtype=np.float64)
    level = sp[:, 0]
    if level.size and np.all(level == np.round(level)):
        # the reference takes the labels out of the (level, lat, lon) tuples of its MultiIndex, so
        # `level` keeps the integer dtype of the ERA5 pressure levels (int64 in the result file,
        # media/svd_netcdf_contents.png); the (m, 3) label array used here is float64 throughout
        level = level.astype(np.int64)
    new = {"level": Coord("space", level), "latitude": Coord("space", sp[:, 1]),
           "longitude": Coord("space", sp[:, 2]), "space": Coord("space", np.arange(sp.shape[0], dtype=np.int64))}
    ds.coords.update(new)
    for da in ds.data_vars.values():
        if "space" in da.dims:
            da.coords.update(new)
    return ds
