"""Drop-in for the reference's SVD driver ``dmd_era5.era5_svd.era5_svd``
(ref: /root/reference/src/dmd_era5/era5_svd/era5_svd.py).

Same public names (``__all__`` of ref: src/dmd_era5/era5_svd/__init__.py:10-17), same
arguments, same return shapes, same exception messages:

    svd_on_era5(da, parsed_config) -> (U (m,k), s (k,), V (k,n))      ref :230-263
    combine_svd_results(U, s, V, coords, X=, X_mean=, X_std=)           ref :266-333
    add_config_attributes(ds, parsed_config)                            ref :42-66
    retrieve_era5_slice / retrieve_svd_results (working-directory branch) ref :69-227
    main(config, write_to_netcdf=False, use_dvc=False)                  ref :336-453
    python -m dmd_era5_amd.era5_svd  (and the ``dmd_era5`` alias package) ref :456-478

What differs is where the arithmetic runs: ``svd_on_era5`` hands X to the MI355X
engine (:mod:`dmd_era5_amd.engine`), and ``main`` never materialises X on the host at
all -- each variable of the slice is uploaded once as ``(time, space)`` row blocks,
centred / scaled in place by K5, delay-embedded as a zero-copy view and decomposed on
the device (the reference holds ~5 copies of X in host RAM).
DVC (``use_dvc=True``) is out of scope and raises.

Optional engine keys in the config dict (absent from the reference, defaults reproduce
its behaviour): ``svd_seed`` (int, makes "randomized" reproducible), ``n_oversamples``,
``n_iter``.
"""
from __future__ import annotations

import logging
import os
import sys
from datetime import datetime

import numpy as np

from . import io_netcdf
from .config_parser import config_parser
from .config_reader import config_reader
from .labeled import Coord, DataArray, Dataset
from .logger import log_and_print, setup_logger
from .slice_tools import (
    apply_delay_embedding,
    delay_coords,
    flatten_era5_variables,
    nearest_resample_index,
    resample_era5_dataset,
    slice_era5_dataset,
    space_coord_to_level_lat_lon,
    space_labels,
    standardize_data,
)

__all__ = [
    "svd_on_era5",
    "combine_svd_results",
    "retrieve_era5_slice",
    "retrieve_svd_results",
    "add_config_attributes",
    "main",
]

logger = setup_logger("ERA5-SVD", "era5_svd.log")
_console = logging.StreamHandler(sys.stdout)
_console.setFormatter(logging.Formatter("%(asctime)s - %(name)s - %(levelname)s - %(message)s"))
logger.addHandler(_console)

_ENGINE_KEYS = {"svd_seed": "random_state", "n_oversamples": "n_oversamples", "n_iter": "n_iter"}


def _engine_opts(parsed_config: dict) -> dict:
    if parsed_config["svd_type"] != "randomized":
        return {}
    return {dst: parsed_config[src] for src, dst in _ENGINE_KEYS.items() if src in parsed_config}


def add_config_attributes(ds: Dataset, parsed_config: dict) -> Dataset:
    """Configuration settings as attributes of the result (ref :42-66)."""
    a = ds.attrs
    a["source_path"] = parsed_config["source_path"]
    a["n_components"] = parsed_config["n_components"]
    a["variables"] = parsed_config["variables"]
    a["levels"] = parsed_config["levels"]
    a["mean_center"] = int(parsed_config["mean_center"])
    a["scale"] = int(parsed_config["scale"])
    a["delay_embedding"] = parsed_config["delay_embedding"]
    a["svd_type"] = parsed_config["svd_type"]
    a["era5_slice_path"] = parsed_config["era5_slice_path"]
    a["date_processed"] = datetime.now().isoformat()
    a["save_data_matrix"] = int(parsed_config["save_data_matrix"])
    return ds


def _as_str_list(obj) -> list[str]:
    if isinstance(obj, str):
        return obj.split(",") if "," in obj else [obj]
    return [str(x) for x in np.atleast_1d(obj).tolist()]


def _as_int_list(obj) -> list[int]:
    arr = np.atleast_1d(obj)
    if not np.issubdtype(arr.dtype, np.integer):
        raise ValueError("Levels must be integers.")
    return [int(x) for x in arr.tolist()]


def _no_dvc(use_dvc: bool) -> None:
    if use_dvc:
        raise NotImplementedError("DVC data versioning is outside the scope of dmd_era5_amd "
                                  "(SURVEY.md section 2 row 6); call with use_dvc=False")


def retrieve_era5_slice(parsed_config: dict, use_dvc: bool = False):
    """ERA5 slice from the working directory (ref :69-154, no-DVC branch): accepted iff the
    requested variables / levels are contained in the file's and ``source_path`` matches."""
    _no_dvc(use_dvc)
    path = parsed_config["era5_slice_path"]
    if not os.path.exists(path):
        log_and_print(logger, "ERA5 slice not found in working directory.", "warning")
        return None, False
    log_and_print(logger, "ERA5 slice found in working directory.")
    ds = io_netcdf.open_dataset(path)
    want_v, want_l = parsed_config["variables"], parsed_config["levels"]
    ok = (sorted(want_v) == sorted(set(_as_str_list(ds.attrs["variables"])) & set(want_v))
          and sorted(want_l) == sorted(set(_as_int_list(ds.attrs["levels"])) & set(want_l))
          and parsed_config["source_path"] == ds.attrs["source_path"])
    if ok:
        log_and_print(logger, "ERA5 slice matches configuration.")
        return ds, False
    log_and_print(logger, "ERA5 slice does not match configuration.")
    log_and_print(logger, "ERA5 slice in working directory does not match configuration.", "warning")
    return None, False


def retrieve_svd_results(parsed_config: dict, use_dvc: bool = False):
    """Cached SVD result from the working directory (ref :157-227, no-DVC branch).  The key
    is the attribute set of ref :178-188 (svd_type is *not* part of it, as in the reference)."""
    _no_dvc(use_dvc)
    path = parsed_config["save_path"]
    if not os.path.exists(path):
        log_and_print(logger, "SVD results not found in working directory.", "warning")
        return None, False
    log_and_print(logger, "SVD results found in working directory.")
    ds = io_netcdf.open_dataset(path)
    a = ds.attrs
    ok = (parsed_config["source_path"] == a["source_path"]
          and parsed_config["n_components"] == a["n_components"]
          and parsed_config["variables"] == _as_str_list(a["variables"])
          and parsed_config["levels"] == _as_int_list(a["levels"])
          and parsed_config["mean_center"] == a["mean_center"]
          and parsed_config["scale"] == a["scale"]
          and parsed_config["delay_embedding"] == a["delay_embedding"])
    if ok:
        log_and_print(logger, "SVD results match configuration.")
        return ds, False
    log_and_print(logger, "SVD results do not match configuration.")
    log_and_print(logger, "SVD results in working directory do not match configuration.", "warning")
    return None, False


def svd_on_era5(da, parsed_config: dict):
    """Rank-``n_components`` SVD of the pre-processed slice (ref :230-263) on the MI355X.

    ``da``: anything with ``.values`` of shape (space, time) -- our DataArray or a real
    ``xr.DataArray`` -- or the ndarray itself.  Returns numpy ``(U, s, V)`` in X's dtype."""
    from .engine import svd_numpy

    X = da.values if hasattr(da, "values") else np.asarray(da)
    svd_type = parsed_config["svd_type"]
    n_components = parsed_config["n_components"]
    if svd_type == "standard":
        log_and_print(logger, "Performing standard SVD...")
    elif svd_type == "randomized":
        log_and_print(logger, "Performing randomized SVD...")
    else:
        raise ValueError(f"SVD type {svd_type} is not supported.")
    U, s, V = svd_numpy(X, svd_type, n_components, **_engine_opts(parsed_config))
    log_and_print(logger, f"{svd_type.capitalize()} SVD complete.")
    return U, s, V


def combine_svd_results(U, s, V, coords, **kwargs) -> Dataset:
    """U(space, components), s(components), V(components, time) [+ X, X_mean, X_std] as
    one Dataset with the coordinates of the decomposed array (ref :266-333)."""
    comp = np.arange(U.shape[1])
    row = {k: coords[k] for k in ("space", "original_variable", "delay") if k in coords}
    cds = dict(coords)
    cds["components"] = Coord("components", comp)
    ds = Dataset(coords=cds)
    ds["U"] = DataArray(U, ("space", "components"), {**row, "components": cds["components"]})
    ds["s"] = DataArray(s, ("components",), {"components": Coord("components", np.arange(s.shape[0]))})
    ds["V"] = DataArray(V, ("components", "time"),
                        {"components": Coord("components", np.arange(V.shape[0])), "time": coords["time"]})
    for key in ("X", "X_mean", "X_std"):
        if kwargs.get(key) is not None:
            ds[key] = kwargs[key]
    return ds


# --------------------------------------------------------------------------------------
# main: the device pipeline
# --------------------------------------------------------------------------------------
SLAB_BYTES = 256 << 20   # host staging slab of the streaming ingest


def plan_selection(ds: Dataset, levels, delta_time):
    """Index form of ``slice_era5_dataset(ds, levels=...)`` + ``resample_era5_dataset``
    (ref era5_svd.py:385-388): which level indices and which time indices the SVD uses, and
    the resulting time coordinate -- computed from the coordinates only, so that a
    file-backed slice is never loaded whole.  Same validation / messages as slice_tools."""
    have = list(ds.coords["level"].values)
    levels = levels or have
    missing = [lv for lv in levels if lv not in have]
    if missing:
        msg = f"Requested level is not available in the dataset.Available levels: {have}"
        log_and_print(logger, msg, "error")
        raise ValueError(msg)
    level_idx = np.array([have.index(lv) for lv in levels])
    times = ds.coords["time"].values
    if len(times) < 2:
        raise ValueError("Start datetime must be before end datetime.")
    labels, take = nearest_resample_index(times, delta_time)
    return level_idx, np.asarray(levels), take, labels


def _upload_variable(da: DataArray, level_idx, take, device, kern, center, scale, stats):
    """One variable (time, level, lat, lon) -> centred/scaled row blocks (time, rows) in HBM.

    Streams time slabs: file/host -> pinned staging -> device slab -> strided device copy
    into each row block.  Row order inside the variable: level slowest, longitude fastest."""
    import torch

    from . import svd as dsvd

    order = [da.dims.index(x) for x in ("time", "level", "latitude", "longitude")]
    if order != [0, 1, 2, 3]:
        raise ValueError(f"variable {da.name}: expected dims (time, level, latitude, longitude), got {da.dims}")
    n = len(take)
    nlev_all, nlat, nlon = da.shape[1:]
    m_v = len(level_idx) * nlat * nlon
    ranges = dsvd.split_rows(m_v)
    blocks = [torch.empty((n, b - a), dtype=torch.float32, device=device) for a, b in ranges]
    lazy = da.lazy
    host = None if lazy is not None else da.values
    rows = max(1, SLAB_BYTES // max(1, nlev_all * nlat * nlon * 4))
    contiguous = np.array_equal(take, np.arange(take[0], take[0] + n)) if n else True
    all_levels = len(level_idx) == nlev_all and np.array_equal(level_idx, np.arange(nlev_all))
    nbytes = 0
    # fast path: fp32 file-backed variable, contiguous snapshots, every level: the slab is read
    # straight into one of two pinned staging buffers and copied to the device asynchronously,
    # so the next read overlaps the previous host->device copy
    direct = lazy is not None and contiguous and all_levels and lazy.dtype == np.float32
    pinned = [torch.empty((rows, m_v), dtype=torch.float32).pin_memory() for _ in range(2)] if direct else None
    events = [None, None]
    for it, j0 in enumerate(range(0, n, rows)):
        j1 = min(n, j0 + rows)
        if direct:
            buf = pinned[it & 1]
            if events[it & 1] is not None:
                events[it & 1].synchronize()        # the copy that last used this buffer is done
            view = buf[: j1 - j0].numpy().reshape((j1 - j0,) + tuple(lazy.shape[1:]))
            lazy.read_slab(int(take[j0]), int(take[j1 - 1]) + 1, view)
            dev = buf[: j1 - j0].to(device, non_blocking=True)
            nbytes += (j1 - j0) * m_v * 4
        else:
            if contiguous:
                t0, t1 = int(take[j0]), int(take[j1 - 1]) + 1
                slab = lazy.read_slab(t0, t1) if lazy is not None else host[t0:t1]
            else:  # resampled: gather the selected snapshots
                idx = take[j0:j1]
                if lazy is not None:
                    lo, hi = int(idx.min()), int(idx.max()) + 1
                    slab = lazy.read_slab(lo, hi)[idx - lo]
                else:
                    slab = host[idx]
            if not all_levels:
                slab = slab[:, level_idx]
            slab = np.ascontiguousarray(slab.reshape(j1 - j0, m_v), dtype=np.float32)
            nbytes += slab.nbytes
            dev = torch.from_numpy(slab).to(device, non_blocking=False)
        for (a, b), Xb in zip(ranges, blocks):
            Xb[j0:j1].copy_(dev[:, a:b])
        if direct:
            events[it & 1] = torch.cuda.Event()
            events[it & 1].record()
        del dev
    for Xb in blocks:
        if center:
            mu, sd = kern.row_center_scale_(Xb, bool(scale))
            stats["mean"].append(mu)
            if scale:
                stats["std"].append(sd)
    return blocks, m_v, nbytes


def _device_pipeline(ds: Dataset, parsed_config: dict):
    """Slice -> (U, s, V, coords, X, X_mean, X_std) with X resident only in HBM.

    Row order = the reference's flatten order (variable-major, then level, latitude,
    longitude; ref slice_tools.py:311-336); embedding order k*m + s (ref :207-211)."""
    import time as _time

    import torch

    from . import svd as dsvd
    from .kernels import default_kernels

    kern = default_kernels()
    device = torch.device("cuda", torch.cuda.current_device())
    d = parsed_config["delay_embedding"]
    center, scale = parsed_config["mean_center"], parsed_config["scale"]
    names = list(ds.data_vars)
    level_idx, levels, take, time = plan_selection(ds, parsed_config["levels"], parsed_config["delta_time"])
    log_and_print(logger, f"Dataset slicing completed successfully using levels {list(levels)}")
    log_and_print(logger, f"Resampled the dataset with time delta: {parsed_config['delta_time']}")
    one = space_labels(levels, ds.coords["latitude"].values, ds.coords["longitude"].values)
    m_v = one.shape[0]

    t0 = _time.perf_counter()
    blocks, stats, total = [], {"mean": [], "std": []}, 0
    for name in names:
        vb, mv, nbytes = _upload_variable(ds[name], level_idx, take, device, kern, center, scale, stats)
        assert mv == m_v
        blocks.extend(vb)
        total += nbytes
    torch.cuda.synchronize()
    dt = _time.perf_counter() - t0
    log_and_print(logger, f"Ingest: {total / 1e9:.3f} GB to HBM in {dt:.2f} s ({total / 1e9 / max(dt, 1e-9):.2f} GB/s, "
                          f"{len(blocks)} row blocks, centre/scale on device)")
    k = parsed_config["n_components"]
    t0 = _time.perf_counter()
    if parsed_config["svd_type"] == "standard":
        log_and_print(logger, "Performing standard SVD...")
        res = dsvd.svd_snapshots(blocks, k, delay=d)
        if res.info.get("mean_deflated"):
            log_and_print(logger, "Un-centred data: SVD of the centred matrix + rank-one update for the time mean.")
        if res.info.get("warning"):
            log_and_print(logger, "WARNING: " + res.info["warning"])
        log_and_print(logger, "Standard SVD complete.")
    else:
        log_and_print(logger, "Performing randomized SVD...")
        res = dsvd.svd_randomized(blocks, k, delay=d, **_engine_opts(parsed_config))
        log_and_print(logger, "Randomized SVD complete.")
    torch.cuda.synchronize()
    dt = _time.perf_counter() - t0
    log_and_print(logger, f"SVD stage: {dt:.3f} s ({total / 1e9 / max(dt, 1e-9):.1f} GB/s of X)")
    src_dtype = ds[names[0]].dtype
    out_dtype = src_dtype if src_dtype in (np.float32, np.float64) else np.float64
    U = res.Ut.cpu().numpy().T.astype(out_dtype, copy=False)
    s = res.s.cpu().numpy().astype(out_dtype, copy=False)
    V = res.Vh.cpu().numpy().astype(out_dtype, copy=False)

    coords = delay_coords(np.tile(one, (len(names), 1)), np.repeat(names, m_v), time, d)
    X = X_mean = X_std = None
    if center and d > 1:  # the reference keeps the mean / std only in this case (ref :400-414)
        mu = torch.cat(stats["mean"]).cpu().numpy().astype(out_dtype)
        X_mean = DataArray(np.tile(mu, d), ("space",), {k_: coords[k_] for k_ in ("space", "original_variable")})
        if scale:
            sd = torch.cat(stats["std"]).cpu().numpy().astype(out_dtype)
            X_std = DataArray(np.tile(sd, d), ("space",), {k_: coords[k_] for k_ in ("space", "original_variable")})
    if parsed_config["save_data_matrix"]:
        from .slice_tools import _apply_delay_embedding_np

        Xc = np.concatenate([b.cpu().numpy() for b in blocks], axis=1).T.astype(out_dtype, copy=False)
        X = DataArray(_apply_delay_embedding_np(np.asfortranarray(Xc), d), ("space", "time"), coords)
    return U, s, V, coords, X, X_mean, X_std


def main(config: dict | None = None, write_to_netcdf: bool = False, use_dvc: bool = False):
    """SVD of an ERA5 slice (ref :336-453).  Returns (results Dataset, added_to_dvc,
    retrieved_from_dvc) -- the two flags are always False here (no DVC)."""
    _no_dvc(use_dvc)
    if config is None:
        config = config_reader("era5-svd")
    parsed_config = config_parser(config, "era5-svd")
    for key in _ENGINE_KEYS:
        if key in config:
            parsed_config[key] = config[key]

    try:
        svd_results, _ = retrieve_svd_results(parsed_config, use_dvc)
    except Exception as e:
        msg = f"Error retrieving SVD results: {e}"
        log_and_print(logger, msg, "error")
        raise Exception(msg) from e
    if svd_results is not None:
        return svd_results, False, False

    try:
        ds, _ = retrieve_era5_slice(parsed_config, use_dvc)
        if ds is None:
            msg = ("\n                    Could not retrieve ERA5 slice from working directory.\n"
                   "                    Consider using DVC to retrieve the ERA5 slice, if available.\n"
                   "                    ")
            log_and_print(logger, msg, "error")
            raise FileNotFoundError(msg)
    except Exception as e:
        msg = f"Error retrieving ERA5 slice: {e}"
        log_and_print(logger, msg, "error")
        raise Exception(msg) from e

    try:
        ds = ds[parsed_config["variables"]]
        # slice_era5_dataset(levels=...) and resample_era5_dataset(...) happen inside, as index
        # selections applied while the slice is streamed to the device
        U, s, V, coords, X, X_mean, X_std = _device_pipeline(ds, parsed_config)
        svd_results = combine_svd_results(U, s, V, coords, X=X, X_mean=X_mean, X_std=X_std)
        svd_results = add_config_attributes(svd_results, parsed_config)
        svd_results = space_coord_to_level_lat_lon(svd_results)
    except Exception as e:
        msg = f"Error in the SVD on ERA5 process: {e}"
        log_and_print(logger, msg, "error")
        raise Exception(msg) from e

    if write_to_netcdf:
        try:
            log_and_print(logger, "Writing SVD results to NetCDF...")
            io_netcdf.to_netcdf(svd_results, parsed_config["save_path"])
            log_and_print(logger, f"SVD results written to {parsed_config['save_path']}")
        except Exception as e:
            msg = f"Error writing SVD results to NetCDF: {e}"
            log_and_print(logger, msg, "error")
            raise Exception(msg) from e
    return svd_results, False, False


# host-only reference pipeline (numpy pre-processing + svd_on_era5), kept for small inputs
# and as the readable statement of what _device_pipeline computes
def host_pipeline(ds: Dataset, parsed_config: dict):
    if parsed_config["mean_center"]:
        ds, ds_mean, ds_std = standardize_data(ds, scale=parsed_config["scale"])
    else:
        ds_mean = ds_std = None
    da = flatten_era5_variables(ds)
    da = apply_delay_embedding(da, parsed_config["delay_embedding"])
    U, s, V = svd_on_era5(da, parsed_config)
    return U, s, V, da


if __name__ == "__main__":
    log_and_print(logger, "Not a Data Version Control (DVC) repository. Will not use DVC.", level="warning")
    main(write_to_netcdf=True)
