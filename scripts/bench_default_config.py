"""Scratch: main() on the reference's DEFAULT config.ini [era5-svd] at the real grid: 4 days of hourly
0.25-degree temperature at 1000 hPa (97 snapshots x 1 038 240 points), randomized, n_components 10,
delay embedding 2, mean-centred, save_data_matrix = True (the 0.8 GB embedded X goes into the file)."""
import os, sys, time, tempfile, shutil
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = tempfile.mkdtemp(prefix="dmdx_default_")
os.environ["DMD_ERA5_ROOT"] = root
from dmd_era5_amd import io_netcdf, hdf5_lite
from dmd_era5_amd.config_parser import config_parser
from dmd_era5_amd.era5_svd import main
cfg = {"source_path": "synthetic", "variables": "temperature", "levels": "1000", "svd_type": "randomized",
       "delay_embedding": 2, "mean_center": True, "scale": False, "start_datetime": "2019-01-01T00",
       "end_datetime": "2019-01-05T00", "delta_time": "1h", "n_components": 10, "save_data_matrix": True}
p = config_parser(cfg, "era5-svd")
n, nlat, nlon = 97, 721, 1440
os.makedirs(os.path.dirname(p["era5_slice_path"]), exist_ok=True)
rs = np.random.RandomState(0)
with hdf5_lite.Writer(p["era5_slice_path"]) as w:
    times = np.datetime64("2019-01-01T00", "ns") + np.arange(n) * np.timedelta64(1, "h")
    hours = ((times - np.datetime64("1970-01-01T00", "ns")) / np.timedelta64(1, "h")).astype(np.int64)
    w.dataset("time", hours, ("time",), {"units": io_netcdf.TIME_UNITS, "calendar": "proleptic_gregorian"})
    w.dataset("level", np.array([1000], dtype=np.int64), ("level",))
    w.dataset("latitude", np.linspace(90, -90, nlat), ("latitude",))
    w.dataset("longitude", np.linspace(0, 359.75, nlon), ("longitude",))
    base = rs.standard_normal((8, nlat * nlon)).astype(np.float32)
    coef = rs.standard_normal((n, 8)).astype(np.float32) * (0.8 ** np.arange(8, dtype=np.float32))
    w.dataset("temperature", (coef @ base + 280).reshape(n, 1, nlat, nlon), ("time", "level", "latitude", "longitude"))
    w.attrs(None, {"source_path": "synthetic", "variables": ["temperature"], "levels": [1000]})
import cProfile, pstats
for rep in range(3):
    pr = cProfile.Profile() if (rep == 2 and len(sys.argv) > 1 and sys.argv[1] == "profile") else None
    t0 = time.perf_counter()
    if pr: pr.enable()
    res, _, _ = main(cfg, write_to_netcdf=True)
    if pr:
        pr.disable(); pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
    dt = time.perf_counter() - t0
    sz = os.path.getsize(p["save_path"]) / 1e9
    print(f"main() call {rep}: {dt:.2f} s total; result file {sz:.2f} GB; s head {res['s'].values[:3]}", flush=True)
    os.remove(p["save_path"])
shutil.rmtree(root, ignore_errors=True)
