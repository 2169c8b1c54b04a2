"""Synthetic ERA5-like slices for tests and BASELINE config 1 (mirror of the reference's
src/dmd_era5/create_mock_data/create_mock_data.py:26-155, with a seed: the reference draws
from the unseeded global numpy state)."""
from __future__ import annotations

from datetime import datetime

import numpy as np

from .labeled import Coord, DataArray, Dataset

DIMS = ("time", "level", "latitude", "longitude")


def _hourly(start, end) -> np.ndarray:
    t0 = np.datetime64(datetime.fromisoformat(start) if isinstance(start, str) else start, "ns")
    t1 = np.datetime64(datetime.fromisoformat(end) if isinstance(end, str) else end, "ns")
    n = int((t1 - t0) / np.timedelta64(1, "h")) + 1
    return t0 + np.arange(n) * np.timedelta64(1, "h")


def create_mock_era5(start_datetime, end_datetime, variables, levels, seed=None,
                     dtype=np.float64) -> Dataset:
    """Hourly fields on a 5-degree grid (36 x 72): temperature = rand*30+250, cooled with
    height and scaled by cos(latitude); winds = rand*20-10; anything else rand*100."""
    rs = np.random.RandomState(seed)
    times = _hourly(start_datetime, end_datetime)
    lats = np.arange(90, -90, -5.0)
    lons = np.arange(-180, 180, 5.0)
    shape = (len(times), len(levels), len(lats), len(lons))
    cds = {"time": Coord("time", times), "level": Coord("level", np.asarray(levels)),
           "latitude": Coord("latitude", lats), "longitude": Coord("longitude", lons)}
    ds = Dataset(coords=cds, attrs={"Conventions": "CF-1.6", "history": "Mock ERA5 data created for testing",
                                    "source": "Generated mock data"})
    for var in variables:
        if var == "temperature":
            data = rs.rand(*shape) * 30 + 250
            for i, level in enumerate(levels):
                data[:, i] -= (1000 - level) / 100
            data = data * np.cos(np.radians(lats))[None, None, :, None]
            units = "K"
        elif "wind" in var:
            data, units = rs.rand(*shape) * 20 - 10, "m/s"
        else:
            data, units = rs.rand(*shape) * 100, "unknown"
        ds[var] = DataArray(data.astype(dtype), DIMS, cds, {"units": units})
    return ds


def add_download_attributes(ds: Dataset, parsed_config: dict) -> Dataset:
    """Attributes the reference's downloader stamps on a slice (era5_download.py:36-42);
    ``retrieve_era5_slice`` keys on source_path / variables / levels."""
    ds.attrs["source_path"] = parsed_config["source_path"]
    ds.attrs["start_datetime"] = parsed_config["start_datetime"].isoformat()
    ds.attrs["end_datetime"] = parsed_config["end_datetime"].isoformat()
    ds.attrs["hours_delta_time"] = parsed_config["delta_time"].total_seconds() / 3600
    ds.attrs["variables"] = parsed_config["variables"]
    ds.attrs["levels"] = parsed_config["levels"]
    ds.attrs["date_downloaded"] = datetime.now().isoformat()
    return ds


def create_mock_era5_svd(start_datetime="2020-01-01", end_datetime="2020-01-02", variables=None, levels=None,
                         mean_center: bool = True, scale: bool = False, delay_embedding: int = 2,
                         n_components: int = 6, seed=None):
    """Mock SVD results for tests of the result schema (mirror of the reference's helper,
    create_mock_data.py:158-221): mock slice -> standardize -> flatten -> delay embedding ->
    rank-``n_components`` SVD.  Returns ``(U, s, V, coords, X)`` with ``X`` the embedded
    (space, time) array.  The reference calls ``np.linalg.svd`` here; this one goes through
    ``svd_on_era5`` like everything else, i.e. it needs the GPU."""
    from .era5_svd import svd_on_era5
    from .slice_tools import apply_delay_embedding, flatten_era5_variables, standardize_data

    ds = create_mock_era5(start_datetime, end_datetime, variables or ["temperature"], levels or [1000], seed=seed)
    if mean_center:
        ds, _, _ = standardize_data(ds, scale=scale)
    da = apply_delay_embedding(flatten_era5_variables(ds), delay_embedding)
    U, s, V = svd_on_era5(da, {"svd_type": "standard", "n_components": n_components})
    return U, s, V, da.coords, da
