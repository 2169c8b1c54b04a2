"""Scratch: file -> HBM ingest rate of main() (HDF5/NETCDF4 slice, lazy time slabs, K5 on device),
i.e. the PCIe-inclusive side of the path that bench.py's `value` leaves out."""
import os, sys, time, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = tempfile.mkdtemp(prefix="dmdx_ingest_")
os.environ["DMD_ERA5_ROOT"] = root
os.environ["DMDX_NETCDF_BACKEND"] = "hdf5"
from dmd_era5_amd import io_netcdf, hdf5_lite
from dmd_era5_amd.config_parser import config_parser
from dmd_era5_amd.era5_svd import main

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
end = str((np.datetime64("2019-01-01T00") + np.timedelta64(n - 1, "h")).astype("datetime64[h]"))
cfg = {"source_path": "synthetic", "variables": "temperature", "levels": "1000",
       "svd_type": sys.argv[2] if len(sys.argv) > 2 else "standard", "svd_seed": 0, "delay_embedding": 2, "mean_center": True, "scale": False,
       "start_datetime": "2019-01-01T00", "end_datetime": end, "delta_time": "1h",
       "n_components": 20, "save_data_matrix": False}
p = config_parser(cfg, "era5-svd")
nlat, nlon = 721, 1440                     # n hourly 0.25-degree fields: 1000 = 4.15 GB
os.makedirs(os.path.dirname(p["era5_slice_path"]), exist_ok=True)
rs = np.random.RandomState(0)
t0 = time.perf_counter()
with hdf5_lite.Writer(p["era5_slice_path"]) as w:
    times = (np.datetime64("2019-01-01T00", "ns") + np.arange(n) * np.timedelta64(1, "h"))
    hours = ((times - np.datetime64("1970-01-01T00", "ns")) / np.timedelta64(1, "h")).astype(np.int64)
    w.dataset("time", hours, ("time",), {"units": io_netcdf.TIME_UNITS, "calendar": "proleptic_gregorian"})
    w.dataset("level", np.array([1000], dtype=np.int64), ("level",))
    w.dataset("latitude", np.linspace(90, -90, nlat), ("latitude",))
    w.dataset("longitude", np.linspace(0, 359.75, nlon), ("longitude",))
    base = rs.standard_normal((8, nlat * nlon)).astype(np.float32)
    coef = rs.standard_normal((n, 8)).astype(np.float32) * (0.8 ** np.arange(8, dtype=np.float32))
    field = (coef @ base + 280).reshape(n, 1, nlat, nlon)
    w.dataset("temperature", field, ("time", "level", "latitude", "longitude"))
    w.attrs(None, {"source_path": "synthetic", "variables": ["temperature"], "levels": [1000]})
print(f"wrote {os.path.getsize(p['era5_slice_path'])/1e9:.2f} GB slice in {time.perf_counter()-t0:.1f} s", flush=True)
del field
t0 = time.perf_counter()
res, _, _ = main(cfg, write_to_netcdf=True)
print(f"main(): {time.perf_counter()-t0:.2f} s total; s head {res['s'].values[:3]}", flush=True)
print("result file:", os.path.getsize(p["save_path"]) / 1e6, "MB")
# the same slice as if it did not fit the HBM: streamed from the file in two passes (4 GiB pieces)
os.remove(p["save_path"])
os.environ["DMDX_STREAM_BYTES"] = str(4 << 30)
t0 = time.perf_counter()
res2, _, _ = main(cfg, write_to_netcdf=False)
sa, sb = res["s"].values.astype(np.float64), res2["s"].values.astype(np.float64)
print(f"main() streamed in 4 GiB pieces: {time.perf_counter()-t0:.2f} s total; max |ds| / s_1 vs the resident run "
      f"{float(np.max(np.abs(sb - sa)) / sa[0]):.1e}; |<u_j, u_j'>| on the leading 8: "
      f"{float(np.min(np.abs(np.sum(res2['U'].values[:, :8].astype(np.float64) * res['U'].values[:, :8], axis=0)))):.6f}", flush=True)
del os.environ["DMDX_STREAM_BYTES"]
main(cfg, write_to_netcdf=True)          # leave a result file for the steps below
if os.environ.get("DMDX_PROFILE_MAIN"):
    import cProfile, pstats
    os.remove(p["save_path"])
    pr = cProfile.Profile(); pr.enable()
    main(cfg, write_to_netcdf=True)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
# one rank's share when the space points are sharded over 8 GPUs: the latitude band read as hyperslabs
import torch
from dmd_era5_amd import era5_svd
from dmd_era5_amd.kernels import default_kernels
ds = io_netcdf.open_dataset(p["era5_slice_path"])
lvl, _, take, _ = era5_svd.plan_selection(ds, None, p["delta_time"])
for rank in (0, 3):
    band = era5_svd.lat_band(nlat, rank, 8)
    for rep in range(2):
        stats = {"mean": [], "std": []}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        blocks, m_v, nbytes = era5_svd._upload_variable(ds["temperature"], lvl, take, torch.device("cuda", 0), default_kernels(),
                                                       True, False, stats, band)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"band {band} of rank {rank}/8, rep {rep}: {nbytes/1e9:.2f} GB in {dt*1e3:.0f} ms = {nbytes/dt/1e9:.1f} GB/s", flush=True)
    ref = torch.from_numpy(np.ascontiguousarray(ds["temperature"].lazy.read_slab(5, 6)[0, 0, band[0]:band[1]].reshape(-1))).cuda()
    got = torch.cat(blocks, dim=1)[5] + torch.cat(stats["mean"])
    print("   row check:", float((got - ref).abs().max()))
    del blocks
import shutil
shutil.rmtree(root, ignore_errors=True)
