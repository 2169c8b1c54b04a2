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
    from .engine import FP64_MAX_BYTES

    if getattr(X, "dtype", None) == np.float64 and X.nbytes > FP64_MAX_BYTES:
        log_and_print(logger, "Input is float64 and larger than the fp64 path takes: the engine computes it in "
                              "fp32 on the MFMA units (the reference's ERA5 slices are float32) and returns "
                              "float64 arrays that carry ~1e-6 relative accuracy, not LAPACK's 1e-12.",
                      level="warning")
    U, s, V = svd_numpy(X, svd_type, n_components, **_engine_opts(parsed_config))
    if s.size and float(s[0]) > 0 and float(s[-1]) <= 1e-7 * float(s[0]):
        log_and_print(logger, "Singular values below 1e-7 s_1 are under the fp32 resolution of the data: their "
                              "columns of U are an arbitrary orthonormal completion (as LAPACK's are for zero "
                              "singular values), not directions of the data.", level="warning")
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


def _resident_reserve_bytes(rows: int, n: int, d: int, k: int, svd_type: str, kern) -> int:
    """HBM the resident path needs NEXT TO the snapshot matrix, from the problem's own sizes (a
    fixed 12 GB used to be added: it refused small slices on small cards and under-reserved cfg3's
    rank 200): the fp64 Gram and the dense pieces of the eigen stage (standard), the partial-tile
    workspace of the batched Gram / product launch as the library itself sizes it, the m x l basis
    and the m x k result (fp32), the two pinned-slab-sized device staging buffers of the ingest,
    and 1 GiB of slack for the allocator."""
    nd = max(1, n - d + 1)
    l = min(nd, k + max(8, k // 4)) if svd_type == "standard" else min(nd, k + 20)
    out = 2 * SLAB_BYTES + (1 << 30)
    out += 4 * rows * d * (l + k)                                   # U' / Y and U
    if svd_type == "standard":
        out += 3 * 8 * n * n                                        # G, the deflated / shifted copy, K8 partials
        out += 8 * 8 * nd * 2 * l                                   # blocks of the eigen stage
    else:
        out += 4 * rows * d * l                                     # Q next to Y in the range finder
    try:
        import ctypes as C

        from .svd import split_rows

        ms = [b - a for a, b in split_rows(rows)][:16]
        arr = (C.c_int64 * len(ms))(*ms)
        if svd_type == "standard":
            out += int(kern._lib.dmdx_syrk_blocks_workspace_bytes(arr, len(ms), n))
        else:
            out += int(kern._lib.dmdx_gemm_tn_blocks_workspace_bytes(arr, len(ms), n, l))
    except Exception:
        out += 8 << 30                                              # a provider without the C ABI (tests)
    return int(out)


def lat_band(nlat: int, rank: int, world: int) -> tuple[int, int]:
    """Latitude rows [i0, i1) of rank ``rank``: the space points are sharded over the GPUs by
    latitude band (every level and variable of the band lives on the same rank), bands of
    nearly equal height."""
    if world > nlat:
        raise ValueError(f"{world} ranks for {nlat} latitude rows: use at most one rank per latitude row")
    return rank * nlat // world, (rank + 1) * nlat // world


def _upload_variable(da: DataArray, level_idx, take, device, kern, center, scale, stats, band=None, pad4=False):
    """One variable (time, level, lat, lon) -> centred/scaled row blocks (time, rows) in HBM.

    ``pad4``: a variable whose number of space points is not a multiple of 4 (a grid with an odd
    number of longitudes and latitudes) gets its last row block widened by 1-3 all-zero space points,
    so that every block keeps the 16-byte-aligned leading dimension the LDS-DMA / 16-byte-load bodies of
    K1 / K2 / K3 need (the register-staged bodies they otherwise fall to are 1.3-1.4x slower).  Zero
    rows change neither X^T X nor the singular values, and their rows of U are exactly zero; the
    caller drops them (returned widths are the padded ones, ``m_v`` the true count).

    Streams time slabs: file/host -> pinned staging -> device slab -> strided device copy
    into each row block.  Row order inside the variable: level slowest, longitude fastest.
    ``band`` = (i0, i1): only these latitude rows (this rank's shard); file-backed variables
    are then read as hyperslabs, so a rank touches only its own bytes of the file."""
    import torch

    from . import svd as dsvd

    order = [da.dims.index(x) for x in ("time", "level", "latitude", "longitude")]
    if order != [0, 1, 2, 3]:
        raise ValueError(f"variable {da.name}: expected dims (time, level, latitude, longitude), got {da.dims}")
    n = len(take)
    nlev_all, nlat_all, nlon = da.shape[1:]
    i0, i1 = band if band is not None else (0, nlat_all)
    nlat = i1 - i0
    whole = (i0, i1) == (0, nlat_all)
    m_v = len(level_idx) * nlat * nlon
    ranges = dsvd.split_rows(m_v)
    blocks = [torch.zeros((n, b - a + (-(b - a)) % 4), dtype=torch.float32, device=device) if pad4 and (b - a) % 4
              else torch.empty((n, b - a), dtype=torch.float32, device=device) for a, b in ranges]
    lazy = da.lazy
    host = None if lazy is not None else da.values
    rows = max(1, SLAB_BYTES // max(1, nlev_all * max(nlat, 1) * nlon * 4))
    contiguous = np.array_equal(take, np.arange(take[0], take[0] + n)) if n else True
    all_levels = len(level_idx) == nlev_all and np.array_equal(level_idx, np.arange(nlev_all))
    nbytes = 0
    # fast path: fp32 file-backed variable, contiguous snapshots, every level: the slab is read
    # straight into one of two pinned staging buffers and copied to the device asynchronously,
    # so the next read overlaps the previous host->device copy
    direct = lazy is not None and contiguous and all_levels and lazy.dtype == np.float32
    on_gpu = device.type == "cuda"
    pinned = None
    if direct:
        pinned = [torch.empty((rows, m_v), dtype=torch.float32) for _ in range(2)]
        if on_gpu:
            pinned = [b.pin_memory() for b in pinned]
    events = [None, None]
    for it, j0 in enumerate(range(0, n, rows)):
        j1 = min(n, j0 + rows)
        if direct:
            buf = pinned[it & 1]
            if events[it & 1] is not None:
                events[it & 1].synchronize()        # the copy that last used this buffer is done
            view = buf[: j1 - j0].numpy().reshape((j1 - j0, nlev_all, nlat, nlon))
            if whole:
                lazy.read_slab(int(take[j0]), int(take[j1 - 1]) + 1, view)
            else:
                lazy.read_box((int(take[j0]), 0, i0, 0), (j1 - j0, nlev_all, nlat, nlon), view)
            dev = buf[: j1 - j0].to(device, non_blocking=True)
            nbytes += (j1 - j0) * m_v * 4
        else:
            def read(lo, hi):
                if lazy is None:
                    return host[lo:hi] if whole else host[lo:hi, :, i0:i1]
                if whole:
                    return lazy.read_slab(lo, hi)
                return lazy.read_box((lo, 0, i0, 0), (hi - lo, nlev_all, nlat, nlon))

            if contiguous:
                slab = read(int(take[j0]), int(take[j1 - 1]) + 1)
            else:  # resampled: gather the selected snapshots
                idx = take[j0:j1]
                lo = int(idx.min())
                slab = read(lo, int(idx.max()) + 1)[idx - lo]
            if not all_levels:
                slab = slab[:, level_idx]
            slab = np.ascontiguousarray(slab.reshape(j1 - j0, m_v), dtype=np.float32)
            nbytes += slab.nbytes
            dev = torch.from_numpy(slab).to(device, non_blocking=False)
        for (a, b), Xb in zip(ranges, blocks):
            Xb[j0:j1, : b - a].copy_(dev[:, a:b])
        if direct and on_gpu:
            events[it & 1] = torch.cuda.Event()
            events[it & 1].record()
        del dev
    for (a, b), Xb in zip(ranges, blocks):
        if center:
            mu, sd = kern.row_center_scale_(Xb[:, : b - a], bool(scale))   # (the zero rows stay zero: 0 / 0 otherwise)
            stats["mean"].append(mu)
            if scale:
                stats["std"].append(sd)
    return blocks, m_v, nbytes


def _device_pipeline(ds: Dataset, parsed_config: dict, comm=None, kern=None, device=None):
    """Slice -> (U, s, V, coords, X, X_mean, X_std) with X resident only in HBM.

    Row order = the reference's flatten order (variable-major, then level, latitude,
    longitude; ref slice_tools.py:311-336); embedding order k*m + s (ref :207-211).

    ``comm`` (svd.TorchDistComm, one process per GPU): the space points are sharded by latitude
    band (``lat_band``); each rank reads, centres and decomposes only its rows, the exchanges
    are the small all-reduces inside the SVD, and the row-sharded results (U, the row means,
    X if asked for) are gathered to rank 0's HOST memory in the global row order for the
    NetCDF write.  Ranks other than 0 return U = X = X_mean = X_std = None.

    ``kern`` / ``device``: the kernel provider and where the row blocks live -- the HIP kernels
    on the current GPU unless a test injects its CPU double (main() never does: without
    libdmdx.so or a GPU ``default_kernels()`` raises)."""
    import time as _time

    import torch

    from . import svd as dsvd
    from .kernels import default_kernels

    comm = comm or dsvd.Comm()
    root = comm.rank == 0
    if kern is None:
        kern = default_kernels()
        device = torch.device("cuda", torch.cuda.current_device())
    sync = torch.cuda.synchronize if device.type == "cuda" else (lambda: None)
    d = parsed_config["delay_embedding"]
    center, scale = parsed_config["mean_center"], parsed_config["scale"]
    names = list(ds.data_vars)
    level_idx, levels, take, time = plan_selection(ds, parsed_config["levels"], parsed_config["delta_time"])
    log_and_print(logger, f"Dataset slicing completed successfully using levels {list(levels)}")
    log_and_print(logger, f"Resampled the dataset with time delta: {parsed_config['delta_time']}")
    lats, lons = ds.coords["latitude"].values, ds.coords["longitude"].values
    nlev, nlat, nlon = len(levels), len(lats), len(lons)
    band = lat_band(nlat, comm.rank, comm.world_size) if comm.world_size > 1 else None

    k = parsed_config["n_components"]
    rows_global = d * len(names) * nlev * nlat * nlon
    wide = rows_global < len(take) - d + 1
    shard = f" (rank {comm.rank} of {comm.world_size}: latitude rows {band[0]}:{band[1]})" if band else ""
    stream_bytes = int(os.environ.get("DMDX_STREAM_BYTES", "0"))      # > 0 forces the streaming path (tests)
    # what can be streamed from the file in passes when X does not fit: anything tall whose data
    # matrix is not asked for (un-centred standard SVD included: the exact mean deflation centres
    # the pieces as they pass; a spectrum steeper than the Gram resolves costs one more pass)
    can_stream = not wide and not parsed_config["save_data_matrix"]
    if device.type == "cuda" and not stream_bytes:
        # X should be resident: find out before the allocator does
        rows = len(names) * nlev * ((band[1] - band[0]) if band else nlat) * nlon
        reserve = _resident_reserve_bytes(rows, len(take), d, k, parsed_config["svd_type"], kern)
        need = 4 * rows * len(take) + reserve
        # free HBM = what the driver reports + what torch's allocator holds cached but unused
        free = torch.cuda.mem_get_info(device)[0] + max(
            0, torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
        fits = need <= free
        if comm.exchanges:   # one decision for all ranks: the resident and the streaming path exchange differently
            flags = comm.allgather(torch.tensor([1 if fits else 0], dtype=torch.int64, device=device))
            fits = all(int(f.item()) for f in flags)
        if not fits:
            if can_stream:
                stream_bytes = max(1 << 30, min(free // 3, 32 << 30))
            else:
                ranks = -(-4 * len(names) * nlev * nlat * nlon * len(take) // max(free - reserve, 1 << 30))
                raise MemoryError(
                    f"the snapshot matrix of this rank ({rows} x {len(take)} fp32 = {4 * rows * len(take) / 1e9:.1f} GB) "
                    f"does not fit the {free / 1e9:.1f} GB of free HBM; shard the space points over more GPUs: "
                    f"python -m torch.distributed.run --nproc-per-node N -m dmd_era5_amd.era5_svd with N >= "
                    f"{max(ranks, comm.world_size + 1)} (without save_data_matrix such a slice is streamed from the "
                    "file in passes instead)")

    blocks = []
    if stream_bytes:
        # X does not fit: the Gram is accumulated and the projection made while latitude sub-bands of
        # the variables pass through the HBM, two or three times (svd.svd_snapshots_streaming)
        if not can_stream:
            raise ValueError("the streaming path needs save_data_matrix = False and a tall problem")
        i0, i1 = band if band else (0, nlat)
        h = max(1, stream_bytes // max(1, 4 * len(take) * nlev * nlon))       # latitude rows per piece
        sub = [(j, min(i1, j + h)) for j in range(i0, i1, h)]
        means, stds, moved = {}, {}, [0]

        def pieces():
            for vi, name in enumerate(names):
                for j0, j1 in sub:
                    st = {"mean": [], "std": []}
                    vb, _, nbytes = _upload_variable(ds[name], level_idx, take, device, kern, center, scale, st, (j0, j1))
                    if center:
                        means[(vi, j0)] = torch.cat(st["mean"])
                    if scale:
                        stds[(vi, j0)] = torch.cat(st["std"])
                    moved[0] += nbytes
                    yield vb

        t0 = _time.perf_counter()
        log_and_print(logger, f"Snapshot matrix larger than the HBM: streaming it in {len(names) * len(sub)} pieces of "
                              f"<= {h} latitude rows{shard}")
        log_and_print(logger, f"Performing {parsed_config['svd_type']} SVD...")
        if parsed_config["svd_type"] == "standard":
            Ub, s_, Vh_, sinfo = dsvd.svd_snapshots_streaming(pieces, k, rows_global, delay=d, comm=comm, kern=kern)
        else:
            Ub, s_, Vh_, sinfo = dsvd.svd_randomized_streaming(pieces, k, rows_global, len(take), delay=d, comm=comm,
                                                               kern=kern, **_engine_opts(parsed_config))
        if sinfo.get("warning"):
            log_and_print(logger, "WARNING: " + sinfo["warning"])
        kk = int(s_.numel())
        Ul = torch.empty((kk, d, len(names), nlev, i1 - i0, nlon), dtype=torch.float32, device=device)
        mean_l = torch.empty((len(names), nlev, i1 - i0, nlon), dtype=torch.float32, device=device)
        std_l = torch.empty_like(mean_l) if scale else None
        idx = 0
        for vi in range(len(names)):
            for j0, j1 in sub:      # a piece's rows: (level, its latitude rows, longitude), embedded as k_delay * m_piece + s
                Ul[:, :, vi, :, j0 - i0:j1 - i0, :] = dsvd._assemble_rows(Ub[idx], d).reshape(kk, d, nlev, j1 - j0, nlon)
                if center:
                    mean_l[vi, :, j0 - i0:j1 - i0, :] = means[(vi, j0)].reshape(nlev, j1 - j0, nlon)
                if scale:
                    std_l[vi, :, j0 - i0:j1 - i0, :] = stds[(vi, j0)].reshape(nlev, j1 - j0, nlon)
                Ub[idx] = None
                idx += 1
        res = dsvd.SvdResult(Ut=Ul.reshape(kk, -1), s=s_, Vh=Vh_, info=sinfo)
        stats = {"mean": [mean_l.reshape(-1)], "std": [std_l.reshape(-1)] if scale else []}
        npass = int(sinfo.get("passes_over_X", 2))
        total = moved[0] // npass
        sync()
        dt = _time.perf_counter() - t0
        log_and_print(logger, f"{parsed_config['svd_type'].capitalize()} SVD complete.")
        log_and_print(logger, f"Ingest + SVD ({npass} passes over the file): {dt:.2f} s "
                              f"({npass * total / 1e9 / max(dt, 1e-9):.1f} GB/s of file data)")
        t0 = _time.perf_counter()
    else:
        t0 = _time.perf_counter()
        # `python -m dmd_era5.era5_svd.era5_svd` is one process per slice: every run is a "first
        # call".  While the slice crosses PCIe (the GPU and most host threads are idle) a side
        # thread takes the SVD path once on a toy matrix, so that the dense libraries' handles and
        # the code objects of every kernel are loaded when the real matrix is resident.  Only for
        # slices of >= 1 GiB: the reference's default config (0.4 GB, scripts/bench_default_config.py)
        # ingests in 0.1-0.4 s, less than the toy run takes -- 2.62 s primed against 1.99 s.
        # (DMDX_PRIME_MIN_BYTES: tests force the primer onto small slices)
        prime_min = int(os.environ.get("DMDX_PRIME_MIN_BYTES", str(1 << 30)))
        primer = _prime_async(device, parsed_config["svd_type"]) if device.type == "cuda" and \
            4 * rows * len(take) >= prime_min else None
        stats, total, true_rows = {"mean": [], "std": []}, 0, []
        for name in names:
            # (a tall problem on a grid whose space-point count is not a multiple of 4: 1-3 zero rows per variable)
            vb, m_v, nbytes = _upload_variable(ds[name], level_idx, take, device, kern, center, scale, stats, band,
                                               pad4=not wide)
            blocks.extend(vb)
            true_rows.extend(b - a for a, b in dsvd.split_rows(m_v))
            total += nbytes
        if primer is not None:
            primer.join()
        sync()
        dt = _time.perf_counter() - t0
        log_and_print(logger, f"Ingest: {total / 1e9:.3f} GB to HBM in {dt:.2f} s ({total / 1e9 / max(dt, 1e-9):.2f} GB/s, "
                              f"{len(blocks)} row blocks, centre/scale on device){shard}")
        t0 = _time.perf_counter()
    if stream_bytes:
        pass          # res, stats are ready
    elif wide:
        # WIDE problem (fewer space rows than snapshots: a coarse mock grid over a long period).
        # sklearn transposes wide inputs (extmath.py:562-566) and LAPACK does not care; here the
        # tall algorithms run on E^T -- the embedded matrix is small by definition (< n^2 floats),
        # so it is materialised -- and the factors swap roles.
        if comm.world_size > 1:
            raise ValueError(f"{rows_global} space rows < {len(take) - d + 1} snapshots: a wide problem is small, "
                             "run it as a single process (no torch.distributed)")
        nd = len(take) - d + 1
        Xall = torch.cat(blocks, dim=1)                                            # (n, M)
        E = torch.cat([Xall[kd:kd + nd].T for kd in range(d)], dim=0).contiguous()  # (d M, nd): row kd*M + s
        del Xall
        log_and_print(logger, f"Performing {parsed_config['svd_type']} SVD...")
        if parsed_config["svd_type"] == "standard":
            rt = dsvd.svd_snapshots(E, k, flip_sign=False, kern=kern)
        else:
            rt = dsvd.svd_randomized(E, k, flip_sign=False, kern=kern, **_engine_opts(parsed_config))
        Ut = rt.Vh.to(torch.float32)                   # (k, d M): left singular vectors of E
        Vh = rt.Ut.to(torch.float64)                   # (k, nd)
        big = Ut.abs().argmax(dim=1, keepdim=True)     # u-based sign convention (svd_flip)
        sign = torch.where(Ut.gather(1, big) < 0, -1.0, 1.0)
        res = dsvd.SvdResult(Ut=Ut * sign, s=rt.s, Vh=Vh * sign.to(torch.float64), info=dict(rt.info, wide=True))
        log_and_print(logger, f"{parsed_config['svd_type'].capitalize()} SVD complete.")
    elif parsed_config["svd_type"] == "standard":
        log_and_print(logger, "Performing standard SVD...")
        res = dsvd.svd_snapshots(blocks, k, delay=d, comm=comm, kern=kern)
        if res.info.get("mean_deflated"):
            log_and_print(logger, "Un-centred data: SVD of the centred matrix + rank-one update for the time mean.")
        if res.info.get("warning"):
            log_and_print(logger, "WARNING: " + res.info["warning"])
        log_and_print(logger, "Standard SVD complete.")
    else:
        log_and_print(logger, "Performing randomized SVD...")
        res = dsvd.svd_randomized(blocks, k, delay=d, comm=comm, kern=kern, **_engine_opts(parsed_config))
        log_and_print(logger, "Randomized SVD complete.")
    if not stream_bytes and any(int(b.shape[1]) != r for b, r in zip(blocks, true_rows)):
        # drop the zero rows the ingest appended (their rows of U are exactly zero); from here on the
        # blocks are the narrow views again
        Mp = sum(int(b.shape[1]) for b in blocks)
        keep, off = [], 0
        for b, r in zip(blocks, true_rows):
            keep.append(torch.arange(off, off + r, device=device))
            off += int(b.shape[1])
        kk_ = res.Ut.shape[0]
        Ut = res.Ut.reshape(kk_, d, Mp).index_select(2, torch.cat(keep)).reshape(kk_, -1)
        res = dsvd.SvdResult(Ut=Ut, s=res.s, Vh=res.Vh, info=res.info)
        blocks = [b[:, :r] for b, r in zip(blocks, true_rows)]
    sync()
    dt = _time.perf_counter() - t0
    if not stream_bytes:
        log_and_print(logger, f"SVD stage: {dt:.3f} s ({total / 1e9 / max(dt, 1e-9):.1f} GB/s of X)")
    src_dtype = ds[names[0]].dtype
    out_dtype = src_dtype if src_dtype in (np.float32, np.float64) else np.float64

    def assemble(t: "torch.Tensor", lead: tuple) -> np.ndarray | None:
        """Rank-local (*lead, rows of this rank in (variable, level, band, lon) order) pieces ->
        the global (*lead, variable, level, latitude, longitude) host array on rank 0."""
        parts = comm.gather_to_root(t)
        if parts is None:
            return None
        if len(parts) == 1:
            return parts[0].numpy().reshape(lead + (len(names), nlev, nlat, nlon))
        out = np.empty(lead + (len(names), nlev, nlat, nlon), dtype=parts[0].numpy().dtype)
        for r, part in enumerate(parts):
            j0, j1 = lat_band(nlat, r, comm.world_size)
            out[..., j0:j1, :] = part.numpy().reshape(lead + (len(names), nlev, j1 - j0, nlon))
        return out

    kk = res.Ut.shape[0]
    Ug = assemble(res.Ut, (kk, d))                       # local row order k_delay*m_loc + s_loc
    U = None if Ug is None else Ug.reshape(kk, -1).T.astype(out_dtype, copy=False)
    s = res.s.cpu().numpy().astype(out_dtype, copy=False)
    V = res.Vh.cpu().numpy().astype(out_dtype, copy=False)

    coords = X = X_mean = X_std = None
    if root:
        one = space_labels(levels, lats, lons)
        coords = delay_coords(np.tile(one, (len(names), 1)), np.repeat(names, one.shape[0]), time, d)
    if center and d > 1:  # the reference keeps the mean / std only in this case (ref :400-414)
        mu = assemble(torch.cat(stats["mean"]), ())
        if root:
            X_mean = DataArray(np.tile(mu.reshape(-1).astype(out_dtype), d), ("space",),
                               {k_: coords[k_] for k_ in ("space", "original_variable")})
        if scale:
            sd = assemble(torch.cat(stats["std"]), ())
            if root:
                X_std = DataArray(np.tile(sd.reshape(-1).astype(out_dtype), d), ("space",),
                                  {k_: coords[k_] for k_ in ("space", "original_variable")})
    if parsed_config["save_data_matrix"]:
        from .slice_tools import _apply_delay_embedding_np

        n = blocks[0].shape[0]
        nt, M = n - d + 1, sum(int(b.shape[1]) for b in blocks)
        on_device = False
        if not comm.exchanges and device.type == "cuda" and nt > 0:
            free = torch.cuda.mem_get_info(device)[0] + max(
                0, torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device))
            on_device = 4 * d * M * nt + (1 << 30) <= free
        if on_device:
            # one rank and room for it: the embedded matrix is laid out on the device as the file
            # wants it -- (space = k_delay * M + s, time) row-major, E[k M + s, t] = X[t + k, s] -- and
            # leaves in one copy.  (The host route below gathers the time-major blocks, embeds them
            # column-major as the reference does and lets the writer transpose 0.8 GB: 0.27 s of the
            # 0.73 s a warm main() takes on the reference's default config; this takes 0.06 s.)
            Xe = torch.empty((d, M, nt), dtype=torch.float32, device=device)
            off = 0
            for b in blocks:
                mb = int(b.shape[1])
                for kd in range(d):
                    Xe[kd, off:off + mb].copy_(b[kd:kd + nt].T)
                off += mb
            X = DataArray(Xe.reshape(d * M, nt).cpu().numpy().astype(out_dtype, copy=False), ("space", "time"), coords)
            del Xe
        else:
            # several ranks, or no room: concatenate on the host (a device-side cat would hold X twice in HBM)
            Xall = torch.cat(blocks, dim=1) if comm.exchanges else torch.cat([b.cpu() for b in blocks], dim=1)
            Xg = assemble(Xall, (n,))                        # (time, variable, level, lat, lon) on rank 0
            del Xall
            if root:
                Xc = Xg.reshape(n, -1).T.astype(out_dtype, copy=False)
                X = DataArray(_apply_delay_embedding_np(np.asfortranarray(Xc), d), ("space", "time"), coords)
    return U, s, V, coords, X, X_mean, X_std


_PRIMED: set = set()


def _prime_async(device, svd_type: str):
    """Start (once per process, device and svd_type) a thread that runs the SVD path on a toy
    matrix with a kernel provider of its own: first-use costs of rocBLAS / rocSOLVER / libdmdx
    (0.3-0.8 s at cfg2 scale) overlap the ingest instead of following it.  Returns the thread, or
    None when there is nothing left to prime or DMDX_NO_PRIME=1."""
    import os
    import threading

    import torch

    key = (str(device), svd_type)
    if key in _PRIMED or os.environ.get("DMDX_NO_PRIME") == "1":
        return None
    _PRIMED.add(key)

    def work():
        try:
            from . import svd as dsvd
            from .kernels import HipKernels

            torch.cuda.set_device(device)
            k2 = HipKernels()
            # a stream of its own: the toy run (it includes the grid-barrier kernels K7L / K10) must not
            # queue in front of, or between, the ingest's copies and K5 launches on the default stream;
            # its workspaces (a few MB) belong to this provider and stream and are released below
            side = torch.cuda.Stream(device=device)
            with torch.cuda.stream(side):
                g = torch.Generator(device=device)
                g.manual_seed(0)
                toy = torch.randn((1024, 8192), device=device, dtype=torch.float32, generator=g)
                halves = [toy[:, :4096].contiguous(), toy[:, 4096:].contiguous()]
                if svd_type == "standard":
                    dsvd.svd_snapshots(halves, 12, delay=2, kern=k2)
                else:
                    dsvd.svd_randomized(halves, 12, delay=2, random_state=0, kern=k2)
                side.synchronize()
            k2.release_workspace()
        except Exception as e:      # priming is an optimisation: never fail the run for it
            logger.debug(f"library priming failed: {e}")

    t = threading.Thread(target=work, name="dmdx-prime", daemon=True)
    t.start()
    return t


def _small_float64_slice(ds: Dataset, parsed_config: dict) -> bool:
    """float64 variables whose embedded snapshot matrix is small enough for the fp64 engine path
    (the reference's mock slices: a few MB).  Real ERA5 slices are float32 and go to the device
    pipeline."""
    from .engine import FP64_MAX_BYTES

    names = list(ds.data_vars)
    if not names or any(np.dtype(ds[v].dtype) != np.float64 for v in names):
        return False
    cells = sum(int(np.prod(ds[v].shape)) for v in names)
    return 8 * cells * int(parsed_config["delay_embedding"]) <= FP64_MAX_BYTES


def _host_pipeline_fp64(ds: Dataset, parsed_config: dict):
    """The reference's own sequence (ref era5_svd.py:384-415) on the mirrored slice tools, for
    small float64 slices: slice -> resample -> standardize -> flatten -> delay-embed on the host
    in float64 (a few MB: not a device job), then ``svd_on_era5``, whose float64 input takes the
    engine's fp64 path (K9 Gram, fp64 eigen stage) -- so that the reference's float64 mock data
    come back with float64 accuracy (1e-12 against numpy), not fp32's 1e-6.  Returns what
    ``_device_pipeline`` returns."""
    from .slice_tools import (apply_delay_embedding, flatten_era5_variables, resample_era5_dataset,
                              slice_era5_dataset, standardize_data)

    d = parsed_config["delay_embedding"]
    ds = slice_era5_dataset(ds, levels=parsed_config["levels"])
    ds = resample_era5_dataset(ds, parsed_config["delta_time"])
    ds_mean = ds_std = None
    if parsed_config["mean_center"] and parsed_config["scale"]:
        ds, ds_mean, ds_std = standardize_data(ds)
    elif parsed_config["mean_center"]:
        ds, ds_mean, ds_std = standardize_data(ds, scale=False)
    da = apply_delay_embedding(flatten_era5_variables(ds), d)
    row = {c: da.coords[c] for c in ("space", "original_variable", "delay") if c in da.coords}
    da_mean = da_std = None
    if ds_mean is not None and d > 1:                                  # (the reference's d == 1 quirk: no X_mean)
        da_mean = DataArray(np.tile(flatten_era5_variables(ds_mean).values, d), ("space",), row)
        if ds_std is not None:
            da_std = DataArray(np.tile(flatten_era5_variables(ds_std).values, d), ("space",), row)
    U, s, V = svd_on_era5(da, parsed_config)
    return U, s, V, da.coords, (da if parsed_config["save_data_matrix"] else None), da_mean, da_std


def _dist_comm():
    """The communicator of this process: single-rank unless it runs under
    ``python -m torch.distributed.run --nproc-per-node N -m dmd_era5_amd.era5_svd`` (one process
    per GPU, RCCL) or inside an already initialised ``torch.distributed`` job.  Returns
    (comm, created) -- ``created``: the process group was made here and is torn down by main.
    DMDX_DIST_BACKEND=gloo / DMDX_DEVICE=i: rehearsal knobs (several ranks on one GPU);
    DMDX_COMM_FORCE=1: a one-rank process group whose collectives are all issued (RCCL itself on
    a one-GPU box)."""
    import os

    import torch
    import torch.distributed as dist

    from . import svd as dsvd

    force = os.environ.get("DMDX_COMM_FORCE") == "1"
    if dist.is_available() and dist.is_initialized():
        return (dsvd.TorchDistComm() if dist.get_world_size() > 1 or force else dsvd.Comm()), False
    if int(os.environ.get("WORLD_SIZE", "1")) <= 1 and not force:
        return dsvd.Comm(), False
    if force:
        for key, val in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(key, val)
    dev = int(os.environ.get("DMDX_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = os.environ.get("DMDX_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    return dsvd.TorchDistComm(), True


def main(config: dict | None = None, write_to_netcdf: bool = False, use_dvc: bool = False):
    """SVD of an ERA5 slice (ref :336-453).  Returns (results Dataset, added_to_dvc,
    retrieved_from_dvc) -- the two flags are always False here (no DVC).

    Under torch.distributed (one process per GPU) the space points are sharded over the ranks
    (``_device_pipeline``); rank 0 assembles, returns and writes the results, the other ranks
    return ``(None, False, False)`` -- unless the results were already on disk, which every
    rank finds for itself."""
    import time

    _no_dvc(use_dvc)
    t_start = time.perf_counter()
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

    t_open = time.perf_counter()
    comm, created = _dist_comm()
    try:
        ds = ds[parsed_config["variables"]]
        # slice_era5_dataset(levels=...) and resample_era5_dataset(...) happen inside, as index
        # selections applied while the slice is streamed to the device
        if comm.world_size == 1 and _small_float64_slice(ds, parsed_config):
            U, s, V, coords, X, X_mean, X_std = _host_pipeline_fp64(ds, parsed_config)
        else:
            U, s, V, coords, X, X_mean, X_std = _device_pipeline(ds, parsed_config, comm)
        svd_results = None
        t_pipe = time.perf_counter()
        if comm.rank == 0:
            svd_results = combine_svd_results(U, s, V, coords, X=X, X_mean=X_mean, X_std=X_std)
            svd_results = add_config_attributes(svd_results, parsed_config)
            svd_results = space_coord_to_level_lat_lon(svd_results)
    except Exception as e:
        msg = f"Error in the SVD on ERA5 process: {e}"
        log_and_print(logger, msg, "error")
        raise Exception(msg) from e
    finally:
        from .kernels import release_cached_workspaces

        release_cached_workspaces()
        if created:
            import torch.distributed as dist

            dist.destroy_process_group()

    t_done = time.perf_counter()
    if comm.rank == 0:
        log_and_print(logger, f"main(): slice found and opened in {t_open - t_start:.2f} s, ingest + SVD + gather "
                              f"{t_pipe - t_open:.2f} s, result Dataset {t_done - t_pipe:.2f} s")
    if write_to_netcdf and svd_results is not None:
        try:
            log_and_print(logger, "Writing SVD results to NetCDF...")
            io_netcdf.to_netcdf(svd_results, parsed_config["save_path"])
            log_and_print(logger, f"SVD results written to {parsed_config['save_path']} "
                                  f"({time.perf_counter() - t_done:.2f} s)")
        except Exception as e:
            msg = f"Error writing SVD results to NetCDF: {e}"
            log_and_print(logger, msg, "error")
            raise Exception(msg) from e
    return svd_results, False, False


if __name__ == "__main__":
    log_and_print(logger, "Not a Data Version Control (DVC) repository. Will not use DVC.", level="warning")
    main(write_to_netcdf=True)
