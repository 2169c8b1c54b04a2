set -e
export DMD_ERA5_ROOT=$(mktemp -d)
cp config.ini $DMD_ERA5_ROOT/config.ini
python - <<'PY'
import os, numpy as np
from dmd_era5_amd import io_netcdf
from dmd_era5_amd.config_reader import config_reader
from dmd_era5_amd.config_parser import config_parser
from dmd_era5_amd.create_mock_data import add_download_attributes, create_mock_era5
cfg = config_reader("era5-svd")
p = config_parser(cfg, "era5-svd")
ds = add_download_attributes(create_mock_era5(cfg["start_datetime"], cfg["end_datetime"], p["variables"], p["levels"], seed=1, dtype=np.float32), p)
os.makedirs(os.path.dirname(p["era5_slice_path"]), exist_ok=True)
print("slice ->", io_netcdf.to_netcdf(ds, p["era5_slice_path"]), p["era5_slice_path"])
PY
python -m dmd_era5.era5_svd.era5_svd | tail -4
ls -la $DMD_ERA5_ROOT/data/era5_svd/
python - <<'PY'
import os, glob
from dmd_era5_amd import io_netcdf
f = glob.glob(os.path.join(os.environ["DMD_ERA5_ROOT"], "data", "era5_svd", "*.nc"))[0]
ds = io_netcdf.open_dataset(f)
print(sorted(ds.data_vars), {k: ds[k].shape for k in ds.data_vars}, ds.attrs.get("svd_type"), ds.attrs.get("n_components"))
PY
# the same entry point, one process per rank (two ranks share the GPU of a one-GPU box over gloo;
# on a multi-GPU node drop the two DMDX_ variables and RCCL is used)
ONE=$DMD_ERA5_ROOT
export DMD_ERA5_ROOT=$(mktemp -d)
cp config.ini $DMD_ERA5_ROOT/config.ini
mkdir -p $DMD_ERA5_ROOT/data/era5_download && cp $ONE/data/era5_download/*.nc $DMD_ERA5_ROOT/data/era5_download/
DMDX_DIST_BACKEND=gloo DMDX_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 -m dmd_era5.era5_svd.era5_svd | grep -i "ingest\|written"
python - "$ONE" <<'PY'
import os, sys, glob, numpy as np
from dmd_era5_amd import io_netcdf
a = io_netcdf.open_dataset(glob.glob(os.path.join(sys.argv[1], "data", "era5_svd", "*.nc"))[0])
b = io_netcdf.open_dataset(glob.glob(os.path.join(os.environ["DMD_ERA5_ROOT"], "data", "era5_svd", "*.nc"))[0])
print("two ranks vs one (randomized, unseeded Omega as in the reference: the runs differ by the sketch): max rel diff of s", float(np.max(np.abs(a["s"].values - b["s"].values) / a["s"].values)),
      "U shapes", a["U"].shape, b["U"].shape)
PY
