"""Scratch: main() on a WIDE problem (1 variable, 1 level, 5-degree mock grid, one year hourly:
d*m = 5184 space rows < n = 8759 snapshots) against numpy on the oracle's pre-processing."""
import os, sys, tempfile, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["DMD_ERA5_ROOT"] = tempfile.mkdtemp(prefix="dmdx_wide_")
os.environ["DMDX_NETCDF_BACKEND"] = "hdf5"
from dmd_era5_amd import io_netcdf
from dmd_era5_amd.config_parser import config_parser
from dmd_era5_amd.create_mock_data import add_download_attributes, create_mock_era5
from dmd_era5_amd.era5_svd import main
for typ in ("standard", "randomized"):
    cfg = {"source_path": "mock", "variables": "temperature", "levels": "1000", "svd_type": typ, "delay_embedding": 2,
           "mean_center": True, "scale": False, "start_datetime": "2019-01-01T00", "end_datetime": "2019-12-31T23",
           "delta_time": "1h", "n_components": 10, "save_data_matrix": True, "svd_seed": 0}
    p = config_parser(cfg, "era5-svd")
    if not os.path.exists(p["era5_slice_path"]):
        ds = add_download_attributes(create_mock_era5(cfg["start_datetime"], cfg["end_datetime"], p["variables"], p["levels"], seed=1, dtype=np.float32), p)
        io_netcdf.to_netcdf(ds, p["era5_slice_path"])
    if os.path.exists(p["save_path"]): os.remove(p["save_path"])
    t0 = time.perf_counter(); res, _, _ = main(cfg, write_to_netcdf=True); dt = time.perf_counter() - t0
    X = res["X"].values.astype(np.float64)
    U, s, V = res["U"].values.astype(np.float64), res["s"].values.astype(np.float64), res["V"].values.astype(np.float64)
    sref = np.linalg.svd(X, compute_uv=False)[:10]
    print(f"{typ}: X {X.shape} main() {dt:.2f} s; max |ds|/s1 {np.abs(s - sref).max() / sref[0]:.1e}; U^T U - I {np.abs(U.T @ U - np.eye(10)).max():.1e}; "
          f"V V^T - I {np.abs(V @ V.T - np.eye(10)).max():.1e}; |X^T u - s v| / s {np.max(np.linalg.norm(X.T @ U - V.T * s, axis=0) / s):.1e}", flush=True)
