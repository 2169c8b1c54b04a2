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
