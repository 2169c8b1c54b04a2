"""a12 / f-2 / f-1 against an HDF5 implementation that is not this repository's: h5py (on the
HDF5 library's own H5DS dimension-scale API -- the layer netCDF-C builds NETCDF4 files on), run
under the image's second interpreter (/opt/conda/bin/python3.9; nothing is installed or
downloaded).  Two directions:

  * the result file `to_netcdf` writes, read by h5py: every variable's dimensions resolve to the
    right scales through DIMENSION_LIST / REFERENCE_LIST, attributes and values arrive unchanged;
  * an ERA5-slice-shaped file made by h5py (`make_scale` / `attach_scale`), read by
    `io_netcdf.open_dataset` and taken through `main()`'s host-side steps.

This is still not netCDF-C or xarray (neither exists here): what THEY make of the files stays
"interop parity unpinned"; what is pinned is that the files are what the HDF5 dimension-scale API
says they are, and that the reader does not depend on its own writer's habits."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from dmd_era5_amd import io_netcdf
from dmd_era5_amd import slice_tools as st
from dmd_era5_amd.config_parser import config_parser
from dmd_era5_amd.create_mock_data import create_mock_era5

PEER_PY = os.environ.get("DMDX_H5PY_PYTHON", "/opt/conda/bin/python3.9")
PEER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "h5py_peer.py")


def _peer(*args):
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    env.pop("LD_LIBRARY_PATH", None)          # the peer brings its own libhdf5
    return subprocess.run([PEER_PY, PEER, *args], capture_output=True, text=True, timeout=120, env=env)


@pytest.fixture(scope="module")
def peer_ok():
    if not os.path.exists(PEER_PY):
        pytest.skip(f"{PEER_PY} not present")
    r = subprocess.run([PEER_PY, "-c", "import h5py"], capture_output=True, text=True, timeout=120)
    if r.returncode != 0:
        pytest.skip("no h5py under the peer interpreter")
    from dmd_era5_amd import hdf5_lite

    if not hdf5_lite.available():
        pytest.skip("libhdf5 not found")


def _sha(a):
    if a.dtype.kind in "OUS":
        return hashlib.sha256("\x00".join(str(x) for x in a.reshape(-1).tolist()).encode()).hexdigest()
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_result_file_read_by_h5py(peer_ok, svd_base_config, project_root, monkeypatch):
    from dmd_era5_amd.era5_svd import add_config_attributes, combine_svd_results
    from dmd_era5_amd.labeled import DataArray

    monkeypatch.setenv("DMDX_NETCDF_BACKEND", "hdf5")
    p = config_parser(dict(svd_base_config, start_datetime="2019-01-01T00", end_datetime="2019-01-02T00",
                           variables="temperature,u_component_of_wind", levels="1000,850", n_components=5,
                           mean_center=True, scale=True), "era5-svd")
    ds = create_mock_era5("2019-01-01", "2019-01-02", ["temperature", "u_component_of_wind"], [1000, 850],
                          seed=5, dtype=np.float32)
    c, mu, sd = st.standardize_data(ds, scale=True)
    da = st.apply_delay_embedding(st.flatten_era5_variables(c), 2)
    U, s, V = np.linalg.svd(da.values, full_matrices=False)
    row = {k: da.coords[k] for k in ("space", "original_variable", "delay")}
    Xm = DataArray(np.tile(st.flatten_era5_variables(mu).values, 2).astype(np.float32), ("space",), row)
    Xs = DataArray(np.tile(st.flatten_era5_variables(sd).values, 2).astype(np.float32), ("space",), row)
    res = combine_svd_results(U[:, :5].astype(np.float32), s[:5].astype(np.float32), V[:5].astype(np.float32),
                              da.coords, X=da, X_mean=Xm, X_std=Xs)
    res = st.space_coord_to_level_lat_lon(add_config_attributes(res, p))
    assert io_netcdf.to_netcdf(res, p["save_path"]) == "hdf5-lite"

    r = _peer("dump", p["save_path"])
    assert r.returncode == 0, r.stderr[-2000:]
    got = json.loads(r.stdout)
    g, dsets = got["root_attrs"], got["datasets"]
    assert g["_NCProperties"].startswith("version=2,")
    # numeric attributes are 1-D of length 1, text is a scalar fixed-length string: netCDF-C's own layout
    # (nc4hdf.c writes every non-text attribute with H5Screate_simple(1, ...))
    assert g["svd_type"] == "randomized" and g["n_components"] == [5] and g["variables"] == ["temperature", "u_component_of_wind"]
    # the dimension scales, as the H5DS API sees them
    for d in ("space", "components", "time"):
        assert dsets[d]["is_scale"] and dsets[d]["attrs"]["NAME"] == d and dsets[d]["attrs"]["CLASS"] == "DIMENSION_SCALE"
    assert sorted(dsets[d]["attrs"]["_Netcdf4Dimid"] for d in ("space", "components", "time")) == [0, 1, 2]   # scalars
    # every variable: its dimensions resolve, through the library, to exactly these scales
    want = {"U": ["space", "components"], "s": ["components"], "V": ["components", "time"], "X": ["space", "time"],
            "X_mean": ["space"], "X_std": ["space"], "level": ["space"], "latitude": ["space"], "longitude": ["space"],
            "delay": ["space"], "original_variable": ["space"]}
    for v, dims in want.items():
        assert not dsets[v]["is_scale"], v
        assert dsets[v]["scales"] == [[d] for d in dims], (v, dsets[v]["scales"])
    # values and dtypes arrive unchanged
    for v in ("U", "s", "V", "X", "X_mean", "X_std"):
        a = res[v].values
        assert dsets[v]["dtype"] == "float32" and dsets[v]["shape"] == list(a.shape) and dsets[v]["sha"] == _sha(a), v
    for cname, kind in {"components": "i", "space": "i", "delay": "i", "level": "i", "latitude": "f", "longitude": "f"}.items():
        a = res.coords[cname].values
        assert dsets[cname]["kind"] == kind and dsets[cname]["dtype"].endswith("64") and dsets[cname]["sha"] == _sha(a), cname
    ov = res.coords["original_variable"].values
    assert dsets["original_variable"]["edge"] == [str(ov[0]), str(ov[-1])] and dsets["original_variable"]["sha"] == _sha(ov)
    assert dsets["U"]["attrs"]["coordinates"] == "delay latitude level longitude original_variable"
    assert " since " in dsets["time"]["attrs"]["units"]


def test_slice_written_by_h5py_goes_through_the_host_pipeline(peer_ok, tmp_path):
    path = str(tmp_path / "peer_slice.nc")
    r = _peer("write", path)
    assert r.returncode == 0, r.stderr[-2000:]
    ds = io_netcdf.open_dataset(path)
    assert list(ds.data_vars) == ["temperature", "u_component_of_wind"]
    t = ds["temperature"]
    assert t.dims == ("time", "level", "latitude", "longitude") and t.shape == (30, 2, 5, 8)
    assert t.values.dtype == np.float32
    rs = np.random.RandomState(11)
    assert np.array_equal(np.asarray(t.values), rs.standard_normal((30, 2, 5, 8)).astype(np.float32))
    assert np.array_equal(np.asarray(ds["u_component_of_wind"].values), rs.standard_normal((30, 2, 5, 8)).astype(np.float32))
    # the ingest's time-slab reads (raw preadv for the contiguous variable, H5Dread hyperslabs for the
    # chunked + deflated one)
    ds2 = io_netcdf.open_dataset(path)
    for name, ref in (("temperature", t.values), ("u_component_of_wind", ds["u_component_of_wind"].values)):
        lazy = getattr(ds2[name], "lazy", None)
        if lazy is not None:
            assert np.array_equal(lazy.read_slab(5, 19), np.asarray(ref)[5:19]), name
    tt = ds.coords["time"].values
    assert tt.dtype.kind == "M" and tt[0] == np.datetime64("2019-01-01T00") and tt[-1] == np.datetime64("2019-01-02T05")
    assert list(ds.coords["level"].values) == [1000, 850]
    assert np.allclose(ds.coords["latitude"].values, np.linspace(90, -90, 5))
    assert ds.attrs["source_path"] == "peer" and list(ds.attrs["levels"]) == [1000, 850]
    # ... and through the reference's host-side steps (slice, resample, standardize, flatten, embed)
    sl = st.slice_era5_dataset(ds, "2019-01-01T02", "2019-01-02T01", [850])
    sl = st.resample_era5_dataset(sl, __import__("datetime").timedelta(hours=2))
    c, mu, _ = st.standardize_data(sl, scale=False)
    da = st.apply_delay_embedding(st.flatten_era5_variables(c), 2)
    assert da.shape == (2 * 2 * 5 * 8, 12 - 1)
    assert np.abs(np.asarray(c["temperature"].values).mean(axis=0)).max() < 1e-6
