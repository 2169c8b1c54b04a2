"""CPU tests of the mirrored reference interface (config, slice tools, result packaging,
NetCDF round trip).  Each test states the reference test whose behaviour it re-checks
(/root/reference/tests/...); data come from the seeded mock generator."""
import os
from datetime import datetime, timedelta

import numpy as np
import pytest

from dmd_era5_amd import io_netcdf, slice_tools as st
from dmd_era5_amd.config_parser import config_parser
from dmd_era5_amd.config_reader import config_reader
from dmd_era5_amd.create_mock_data import add_download_attributes, create_mock_era5
from dmd_era5_amd.labeled import DataArray, Dataset


# ---- config parser: reference tests/test_00_config_parser.py, test_03_era5_svd.py:71-150
def test_config_parser_basic(svd_base_config, project_root):
    p = config_parser(svd_base_config, section="era5-svd")
    assert p["start_datetime"] == datetime(2019, 1, 1, 6, 0)
    assert p["end_datetime"] == datetime(2020, 1, 1, 12, 0)
    assert p["delta_time"] == timedelta(hours=1)
    assert p["save_name"] == "2019-01-01T06_2020-01-01T12_1h.nc"
    assert p["save_path"] == os.path.join(str(project_root), "data", "era5_svd", p["save_name"])
    assert p["era5_slice_path"] == os.path.join(str(project_root), "data", "era5_download", p["save_name"])
    assert p["era5_svd_path"] == p["save_path"]
    assert p["variables"] == ["temperature"] and p["levels"] == [1000]


@pytest.mark.parametrize("field", ["source_path", "variables", "levels", "svd_type", "delay_embedding",
                                   "mean_center", "scale", "start_datetime", "end_datetime",
                                   "delta_time", "n_components", "save_data_matrix"])
def test_config_parser_missing_field(svd_base_config, field):
    del svd_base_config[field]
    with pytest.raises(ValueError, match=f"Missing required field in config: {field}"):
        config_parser(svd_base_config, section="era5-svd")


def test_config_parser_invalid_values(svd_base_config):
    for key, bad, msg in [("svd_type", "invalid", "Invalid SVD type in config"),
                          ("delay_embedding", 0, "Invalid delay embedding in config"),
                          ("delay_embedding", 1.2, "Invalid delay embedding in config"),
                          ("delay_embedding", "invalid", "Invalid delay embedding in config"),
                          ("n_components", 0, "Invalid number of components in config"),
                          ("n_components", 1.2, "Invalid number of components in config"),
                          ("n_components", "invalid", "Invalid number of components in config"),
                          ("start_datetime", "2019-13-01", "Invalid datetime"),
                          ("delta_time", "1x", "Error parsing delta_time"),
                          ("delta_time", "h", "Error parsing delta_time"),
                          ("variables", "2m_temperature", "Single level variables not currently supported"),
                          ("levels", "999", "Unsupported level in config")]:
        cfg = dict(svd_base_config)
        cfg[key] = bad
        with pytest.raises(ValueError, match=msg):
            config_parser(cfg, section="era5-svd")
    with pytest.raises(ValueError, match="is not currently supported"):
        config_parser(svd_base_config, section="nope")


@pytest.mark.parametrize("text,expected", [("1h", timedelta(hours=1)), ("24h", timedelta(hours=24)),
                                           ("1d", timedelta(days=1)), ("7d", timedelta(days=7)),
                                           ("2w", timedelta(weeks=2)), ("1m", timedelta(days=30)),
                                           ("1y", timedelta(days=365))])
def test_delta_time_units(svd_base_config, text, expected):
    cfg = dict(svd_base_config, delta_time=text)
    assert config_parser(cfg, "era5-svd")["delta_time"] == expected


def test_time_validation(svd_base_config):
    with pytest.raises(ValueError, match="End datetime must be after start datetime"):
        config_parser(dict(svd_base_config, end_datetime="2019-01-01T06"), "era5-svd")
    with pytest.raises(ValueError, match="Time range must be at least as long as delta_time"):
        config_parser(dict(svd_base_config, end_datetime="2019-01-01T07", delta_time="1d"), "era5-svd")
    with pytest.raises(ValueError, match="Start date cannot be in the future"):
        config_parser(dict(svd_base_config, start_datetime="2999-01-01T00", end_datetime="2999-02-01T00"),
                      "era5-svd")


# ---- config reader: reference tests/test_00_config_reader.py
def test_config_reader_types_and_errors(tmp_path):
    ini = tmp_path / "config.ini"
    ini.write_text('[test-section-0]\nparam_0 = "value_0"\nparam_2 = "a,b"\n\n'
                   '[test-section-1]\nparam_3 = True\nparam_4 = 2\nparam_5 = 1.5\n')
    c0 = config_reader("test-section-0", str(ini))
    assert c0 == {"param_0": "value_0", "param_2": "a,b"}
    c1 = config_reader("test-section-1", str(ini))
    assert c1["param_3"] is True and c1["param_4"] == 2 and isinstance(c1["param_5"], float)
    with pytest.raises(Exception, match="Section nope not found"):
        config_reader("nope", str(ini))
    (tmp_path / "bad.ini").write_text("[s]\nx = not a literal\n")
    with pytest.raises((ValueError, SyntaxError)):
        config_reader("s", str(tmp_path / "bad.ini"))


def test_shipped_config_ini_has_the_reference_keys():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = config_reader("era5-svd", os.path.join(root, "config.ini"))
    assert set(cfg) >= {"source_path", "start_datetime", "end_datetime", "delta_time", "variables", "levels",
                        "svd_type", "delay_embedding", "mean_center", "scale", "n_components",
                        "save_data_matrix"}
    assert isinstance(cfg["delay_embedding"], int) and isinstance(cfg["mean_center"], bool)
    dl = config_reader("era5-download", os.path.join(root, "config.ini"))
    assert set(dl) == {"source_path", "start_datetime", "end_datetime", "delta_time", "variables", "levels"}


# ---- slice tools: reference tests/test_02_slice_tools.py
@pytest.fixture
def mock_tw():
    return create_mock_era5("2019-01-01", "2019-01-02", ["temperature", "u_component_of_wind"], [1000, 850],
                            seed=11)


def test_slice_time_and_levels(mock_tw):
    s = st.slice_era5_dataset(mock_tw, "2019-01-01T06", "2019-01-01T12", levels=[850])
    assert s.sizes["time"] == 7 and list(s.coords["level"].values) == [850]
    s = st.slice_era5_dataset(mock_tw, levels=[850, 1000])
    assert list(s.coords["level"].values) == [850, 1000]        # requested order
    assert np.array_equal(s["temperature"].values[:, 0], mock_tw["temperature"].values[:, 1])
    with pytest.raises(ValueError, match="outside dataset"):
        st.slice_era5_dataset(mock_tw, "2018-12-31T00", "2019-01-01T12")
    with pytest.raises(ValueError, match="Start datetime must be before end datetime"):
        st.slice_era5_dataset(mock_tw, "2019-01-01T12", "2019-01-01T06")
    with pytest.raises(ValueError, match="Requested level is not available"):
        st.slice_era5_dataset(mock_tw, levels=[500])


def test_resample_nearest_6h(mock_tw):
    r = st.resample_era5_dataset(mock_tw, timedelta(hours=6))
    assert r.sizes["time"] == 5, "Expected 5 time points"
    dt = np.diff(r.coords["time"].values).astype("timedelta64[h]").astype(int)
    assert np.all(dt == 6)
    assert np.array_equal(r["temperature"].values[1], mock_tw["temperature"].values[6])
    same = st.resample_era5_dataset(mock_tw, timedelta(hours=1))
    assert same.sizes["time"] == 25


def test_standardize_and_flatten_and_embed(mock_tw):
    c, mu, sd = st.standardize_data(mock_tw)
    assert np.allclose(c["temperature"].values.mean(axis=0), 0, atol=1e-6)
    assert np.allclose(c["u_component_of_wind"].values.std(axis=0), 1, atol=1e-6)
    c2, _, none = st.standardize_data(mock_tw, scale=False)
    assert none is None and not np.allclose(c2["temperature"].values.std(axis=0), 1, atol=1e-6)
    da = st.flatten_era5_variables(mock_tw)
    n_space = 2 * 36 * 72
    assert da.shape == (2 * n_space, 25) and da.dims == ("space", "time")
    assert sorted(da.coords) == ["original_variable", "space", "time"]
    assert da.attrs["original_variables"] == ["temperature", "u_component_of_wind"]
    lats, lons = mock_tw.coords["latitude"].values, mock_tw.coords["longitude"].values
    for level, lat, lon in [(1000, 40, 90), (1000, -20, -50), (850, 0, 0)]:
        rows = np.nonzero((da.coords["space"].values == [level, lat, lon]).all(axis=1))[0]
        assert len(rows) == 2
        li = [1000, 850].index(level)
        i, j = int(np.where(lats == lat)[0][0]), int(np.where(lons == lon)[0][0])
        assert np.allclose(da.values[rows[0]], mock_tw["temperature"].values[:, li, i, j])
        assert np.allclose(da.values[rows[1]], mock_tw["u_component_of_wind"].values[:, li, i, j])
    flat_mean = st.flatten_era5_variables(mu)
    assert flat_mean.dims == ("space",) and flat_mean.shape == (2 * n_space,)
    for d in (2, 3):
        e = st.apply_delay_embedding(da, d)
        assert e.shape == (da.shape[0] * d, da.shape[1] - d + 1)
        assert sorted(e.coords) == ["delay", "original_variable", "space", "time"]
        assert np.array_equal(np.unique(e.coords["delay"].values), np.arange(d))
        assert np.array_equal(e.coords["time"].values, da.coords["time"].values[d - 1:])
        m = da.shape[0]
        for k in range(d):                       # block k carries delay d-1-k and X[:, k:k+nt]
            assert np.all(e.coords["delay"].values[k * m:(k + 1) * m] == d - 1 - k)
            assert np.array_equal(e.values[k * m:(k + 1) * m], da.values[:, k:k + e.shape[1]])
    with pytest.raises(ValueError, match="Input data must be a xr.DataArray"):
        st.apply_delay_embedding(da.values, 2)


def test_space_coord_to_level_lat_lon(mock_tw):
    da = st.flatten_era5_variables(mock_tw[["temperature"]])
    ds = Dataset({"temperature": da})
    sp = da.coords["space"].values.copy()
    out = st.space_coord_to_level_lat_lon(ds)
    assert np.array_equal(out.coords["space"].values, np.arange(len(sp)))
    assert (sp[0] == [out.coords["level"].values[0], out.coords["latitude"].values[0],
                      out.coords["longitude"].values[0]]).all()
    assert (sp[-1] == [out.coords["level"].values[-1], out.coords["latitude"].values[-1],
                       out.coords["longitude"].values[-1]]).all()
    assert st.space_coord_to_level_lat_lon(out) is out    # second call is a no-op


# ---- result packaging + NetCDF round trip: reference tests/test_03_era5_svd.py:179-235
def test_combine_and_roundtrip(svd_base_config, project_root):
    from dmd_era5_amd.era5_svd import (add_config_attributes, combine_svd_results,
                                       retrieve_era5_slice, retrieve_svd_results)

    p = config_parser(dict(svd_base_config, start_datetime="2019-01-01T00", end_datetime="2019-01-02T00",
                           n_components=6), "era5-svd")
    assert retrieve_era5_slice(p, use_dvc=False) == (None, False)       # nothing on disk yet
    assert retrieve_svd_results(p) == (None, False)
    ds = create_mock_era5("2019-01-01", "2019-01-02", ["temperature"], [1000], seed=5)
    c, mu, _ = st.standardize_data(ds, scale=False)
    da = st.apply_delay_embedding(st.flatten_era5_variables(c), 2)
    U, s, V = np.linalg.svd(da.values, full_matrices=False)             # packaging only
    U, s, V = U[:, :6], s[:6], V[:6]
    res = combine_svd_results(U, s, V, da.coords, X=da)
    assert sorted(res.data_vars) == ["U", "V", "X", "s"]
    assert res["U"].dims == ("space", "components") and res["V"].dims == ("components", "time")
    assert sorted(res["U"].coords) == ["components", "delay", "original_variable", "space"]
    assert sorted(res["s"].coords) == ["components"] and sorted(res["V"].coords) == ["components", "time"]
    res = st.space_coord_to_level_lat_lon(add_config_attributes(res, p))
    assert res.attrs["mean_center"] == 0 and res.attrs["svd_type"] == "randomized"
    io_netcdf.to_netcdf(res, p["save_path"])
    back, _ = retrieve_svd_results(p)                                   # cache hit on matching attrs
    assert back is not None
    assert np.allclose(back["U"].values, U) and np.allclose(back["s"].values, s)
    assert back["X"].shape == da.shape
    assert list(np.unique(back.coords["original_variable"].values)) == ["temperature"]
    assert np.array_equal(back.coords["time"].values, da.coords["time"].values)
    other = config_parser(dict(svd_base_config, start_datetime="2019-01-01T00",
                               end_datetime="2019-01-02T00", n_components=7), "era5-svd")
    assert retrieve_svd_results(other) == (None, False)                 # attrs differ -> no hit


def test_era5_slice_roundtrip_and_match(svd_base_config, project_root):
    from dmd_era5_amd.era5_svd import retrieve_era5_slice

    cfg = dict(svd_base_config, start_datetime="2019-01-01T00", end_datetime="2019-01-02T00",
               variables="temperature,u_component_of_wind", levels="1000,850")
    p = config_parser(cfg, "era5-svd")
    ds = add_download_attributes(
        create_mock_era5("2019-01-01", "2019-01-02", p["variables"], p["levels"], seed=2, dtype=np.float32), p)
    io_netcdf.to_netcdf(ds, p["era5_slice_path"])
    got, _ = retrieve_era5_slice(p)
    assert got is not None and np.array_equal(got["temperature"].values, ds["temperature"].values)
    sub = config_parser(dict(cfg, variables="temperature", levels="850"), "era5-svd")
    assert retrieve_era5_slice(sub)[0] is not None                      # subset of the file -> accepted
    wrong = config_parser(dict(cfg, variables="v_component_of_wind"), "era5-svd")
    assert retrieve_era5_slice(wrong)[0] is None
    with pytest.raises(NotImplementedError):
        retrieve_era5_slice(p, use_dvc=True)


def test_main_reports_missing_slice_like_the_reference(svd_base_config, project_root):
    from dmd_era5_amd.era5_svd import main

    with pytest.raises(Exception, match="Error retrieving ERA5 slice"):
        main(svd_base_config)


# ---- NETCDF4 through the ctypes HDF5 binding ------------------------------------------
@pytest.mark.parametrize("backend", ["hdf5", "scipy"])
def test_netcdf_backends_roundtrip_the_same_dataset(backend, tmp_path, monkeypatch):
    from dmd_era5_amd import hdf5_lite

    if backend == "hdf5" and not hdf5_lite.available():
        pytest.skip("libhdf5 not found")
    monkeypatch.setenv("DMDX_NETCDF_BACKEND", backend)
    ds = create_mock_era5("2019-01-01", "2019-01-02", ["temperature", "u_component_of_wind"], [1000, 850],
                          seed=8, dtype=np.float32)
    ds.attrs.update(source_path="gs://x", variables=["temperature", "u_component_of_wind"], levels=[1000, 850],
                    hours_delta_time=1.0, mean_center=True)
    path = str(tmp_path / "slice.nc")
    used = io_netcdf.to_netcdf(ds, path)
    assert used == ("hdf5-lite" if backend == "hdf5" else "scipy-netcdf3")
    with open(path, "rb") as fh:
        assert fh.read(4) == (b"\x89HDF" if backend == "hdf5" else b"CDF\x02")
    back = io_netcdf.open_dataset(path)
    assert list(back.data_vars) == ["temperature", "u_component_of_wind"]
    assert back["temperature"].dims == ("time", "level", "latitude", "longitude")
    assert np.array_equal(back["temperature"].values, ds["temperature"].values)
    assert np.array_equal(back.coords["time"].values, ds.coords["time"].values)
    assert list(back.coords["level"].values) == [1000, 850]
    from dmd_era5_amd.era5_svd import _as_int_list, _as_str_list

    assert _as_str_list(back.attrs["variables"]) == ["temperature", "u_component_of_wind"]
    assert _as_int_list(back.attrs["levels"]) == [1000, 850]
    assert back.attrs["source_path"] == "gs://x" and back.attrs["mean_center"] == 1


def test_hdf5_lazy_variables_and_time_slabs(tmp_path, monkeypatch):
    from dmd_era5_amd import hdf5_lite

    if not hdf5_lite.available():
        pytest.skip("libhdf5 not found")
    monkeypatch.setenv("DMDX_NETCDF_BACKEND", "hdf5")
    monkeypatch.setattr(io_netcdf, "LAZY_BYTES", 1000)
    ds = create_mock_era5("2019-01-01", "2019-01-03", ["temperature"], [1000, 850, 500], seed=3, dtype=np.float32)
    path = str(tmp_path / "lazy.nc")
    io_netcdf.to_netcdf(ds, path)
    back = io_netcdf.open_dataset(path)
    lazy = back["temperature"].lazy
    assert lazy is not None and lazy.shape == (49, 3, 36, 72) and lazy.dtype == np.float32
    assert np.array_equal(lazy.read_slab(10, 17), ds["temperature"].values[10:17])
    assert back["temperature"].lazy is not None          # slab reads do not load the variable
    assert np.array_equal(back["temperature"].values, ds["temperature"].values)
    assert back["temperature"].lazy is None              # .values loaded it
    # dimension scales make the file a NETCDF4 file: every variable knows its dimension names
    with hdf5_lite.Reader(path) as r:
        assert r.variables["temperature"][2] == ("time", "level", "latitude", "longitude")
        assert r.attrs("time")["units"].startswith("hours since")


def test_hdf5_contiguous_datasets_are_read_by_parallel_preads(tmp_path, monkeypatch):
    """Contiguous little-endian numeric datasets take the raw pread path (several threads);
    it must return exactly what H5Dread returns, for whole reads and for time slabs."""
    from dmd_era5_amd import hdf5_lite

    if not hdf5_lite.available():
        pytest.skip("libhdf5 not found")
    rs = np.random.RandomState(0)
    a = rs.standard_normal((40, 3, 90, 180)).astype(np.float32)       # 7.8 MB: a single piece
    b = rs.standard_normal((70, 400, 400)).astype(np.float64)         # 89.6 MB: 11 pieces
    path = str(tmp_path / "raw.nc")
    with hdf5_lite.Writer(path) as w:
        w.dataset("time", np.arange(40, dtype=np.int64), ("time",))
        w.dataset("a", a, ("time", "level", "latitude", "longitude"))
        w.dataset("b", b, ("t2", "y", "x"))
    with hdf5_lite.Reader(path) as r:
        assert {"a", "b", "time"} <= set(r.raw_offset)
        assert np.array_equal(r.read("a"), a) and np.array_equal(r.read("b"), b)
        assert np.array_equal(r.read_slab("b", 13, 57), b[13:57])
        out = np.empty((5, 3, 90, 180), dtype=np.float32)
        assert r.read_slab("a", 35, 40, out) is out and np.array_equal(out, a[35:])
        monkeypatch.setattr(hdf5_lite, "RAW_READ_THREADS", 0)           # the H5Dread path
        assert np.array_equal(r.read_slab("b", 13, 57), b[13:57])
        with pytest.raises(ValueError):
            r.read_slab("a", 0, 5, np.empty((5, 3, 90, 180), dtype=np.float64))


def test_hdf5_read_box_is_a_hyperslab_on_both_read_paths(tmp_path, monkeypatch):
    """One rank's share of a time slab (a latitude band of every level) = a hyperslab: the
    pread runs and H5Sselect_hyperslab must both return the numpy slice."""
    from dmd_era5_amd import hdf5_lite
    from dmd_era5_amd.labeled import LazyArray

    if not hdf5_lite.available():
        pytest.skip("libhdf5 not found")
    rs = np.random.RandomState(1)
    a = rs.standard_normal((30, 3, 45, 64)).astype(np.float32)
    b = rs.standard_normal((9, 700, 700)).astype(np.float64)          # runs of 3.9 MB: several tasks
    path = str(tmp_path / "box.nc")
    with hdf5_lite.Writer(path) as w:
        w.dataset("a", a, ("time", "level", "latitude", "longitude"))
        w.dataset("b", b, ("t2", "y", "x"))
    boxes_a = [((4, 0, 10, 0), (11, 3, 17, 64)), ((0, 1, 0, 5), (30, 2, 45, 7)), ((29, 2, 44, 63), (1, 1, 1, 1)),
               ((0, 0, 0, 0), (30, 3, 45, 64)), ((3, 0, 0, 0), (0, 3, 45, 64))]
    with hdf5_lite.Reader(path) as r:
        for threads in (hdf5_lite.RAW_READ_THREADS or 8, 0):
            monkeypatch.setattr(hdf5_lite, "RAW_READ_THREADS", threads)
            for st, ct in boxes_a:
                ref = a[tuple(slice(x, x + c) for x, c in zip(st, ct))]
                assert np.array_equal(r.read_box("a", st, ct), ref)
            assert np.array_equal(r.read_box("b", (1, 100, 0), (8, 555, 700)), b[1:9, 100:655])
            out = np.empty((2, 3, 5, 64), dtype=np.float32)
            assert r.read_box("a", (7, 0, 40, 0), (2, 3, 5, 64), out) is out and np.array_equal(out, a[7:9, :, 40:45])
            with pytest.raises(IndexError):
                r.read_box("a", (0, 0, 40, 0), (1, 3, 6, 64))
            with pytest.raises(ValueError):
                r.read_box("a", (0, 0, 0), (1, 1, 1))
    # arrays that only know time slabs get the box by slicing the slab
    la = LazyArray(a.shape, a.dtype, lambda: a, lambda t0, t1, out=None: a[t0:t1])
    assert np.array_equal(la.read_box((4, 0, 10, 0), (11, 3, 17, 64)), a[4:15, :, 10:27])


def test_hdf5_label_coordinates_with_millions_of_strings(tmp_path):
    """`original_variable (space)`-type coordinates: run-length factorised on write, converted by
    the C helper on read (ASCII fast path, UTF-8 path, no-run fallback)."""
    from dmd_era5_amd import hdf5_lite

    if not hdf5_lite.available():
        pytest.skip("libhdf5 not found")
    runs = np.tile(np.repeat(np.array(["temperature", "u_component_of_wind", "v"]), 3000), 2)
    utf8 = np.repeat(np.array(["température", "vent_zonal"]), 5000)
    rs = np.random.RandomState(1)
    norun = np.array(["n%d" % i for i in rs.randint(0, 50, 6000)])
    path = str(tmp_path / "labels.nc")
    with hdf5_lite.Writer(path) as w:
        w.dataset("runs", runs, ("space",))
        w.dataset("utf8", utf8, ("s2",))
        w.dataset("norun", norun, ("s3",))
        w.dataset("few", np.array(["a", "bb", ""]), ("s4",))
    with hdf5_lite.Reader(path) as r:
        for name, ref in (("runs", runs), ("utf8", utf8), ("norun", norun), ("few", np.array(["a", "bb", ""]))):
            got = r.read(name)
            assert got.shape == ref.shape and np.array_equal(got, ref), name
    uq, inv = hdf5_lite._factorize(runs)
    assert np.array_equal(uq[inv], runs) and len(uq) == 3


def test_svd_result_file_has_the_hdf5_objects_netcdf_c_requires(svd_base_config, project_root, monkeypatch):
    """a12 / f-2: `svd_results.to_netcdf(save_path, format="NETCDF4")` (reference era5_svd.py:434).
    No netCDF library is importable in this image, so what netCDF-C / xarray make of the file is
    INTEROP PARITY UNPINNED; pinned instead is the HDF5 structure a NETCDF4 file consists of, walked
    object by object: the provenance attribute, one dimension scale per dimension (CLASS, NAME,
    _Netcdf4Dimid, back-references), the netCDF placeholder naming for dimensions without a
    coordinate variable, DIMENSION_LIST on every variable, fixed-length text attributes, the
    coordinate dtypes of the reference's result schema (media/svd_netcdf_contents.png: int64
    components / space / delay / level, float64 latitude / longitude, strings for
    original_variable, float32 data), and the CF `coordinates` attribute xarray adds -- from which
    (not from a fixed list) the reader then rebuilds the same coordinate set as the README's schema."""
    from dmd_era5_amd import hdf5_lite
    from dmd_era5_amd.era5_svd import add_config_attributes, combine_svd_results

    if not hdf5_lite.available():
        pytest.skip("libhdf5 not found")
    monkeypatch.setenv("DMDX_NETCDF_BACKEND", "hdf5")
    p = config_parser(dict(svd_base_config, start_datetime="2019-01-01T00", end_datetime="2019-01-02T00",
                           variables="temperature,u_component_of_wind", levels="1000,850", n_components=5,
                           mean_center=True, scale=True), "era5-svd")
    ds = create_mock_era5("2019-01-01", "2019-01-02", ["temperature", "u_component_of_wind"], [1000, 850],
                          seed=5, dtype=np.float32)
    c, mu, sd = st.standardize_data(ds, scale=True)
    da = st.apply_delay_embedding(st.flatten_era5_variables(c), 2)
    m = da.shape[0]
    U, s, V = np.linalg.svd(da.values, full_matrices=False)
    from dmd_era5_amd.labeled import DataArray, Dataset

    row = {k: da.coords[k] for k in ("space", "original_variable", "delay")}
    Xm = DataArray(np.tile(st.flatten_era5_variables(mu).values, 2).astype(np.float32), ("space",), row)
    Xs = DataArray(np.tile(st.flatten_era5_variables(sd).values, 2).astype(np.float32), ("space",), row)
    res = combine_svd_results(U[:, :5].astype(np.float32), s[:5].astype(np.float32), V[:5].astype(np.float32),
                              da.coords, X=da, X_mean=Xm, X_std=Xs)
    res = st.space_coord_to_level_lat_lon(add_config_attributes(res, p))
    assert io_netcdf.to_netcdf(res, p["save_path"]) == "hdf5-lite"

    with hdf5_lite.Reader(p["save_path"]) as r:
        g = r.attrs(None, raw=True)
        assert g["_NCProperties"].startswith("version=2,") and "hdf5=" in g["_NCProperties"]
        # global attributes: the eleven of add_config_attributes (era5_svd.py:55-65); text as
        # fixed-length strings, flags as integers, lists of str as 1-D string arrays
        assert g["svd_type"] == "randomized" and g["n_components"] == 5 and g["mean_center"] == 1 and g["scale"] == 1
        assert list(g["variables"]) == ["temperature", "u_component_of_wind"] and list(g["levels"]) == [1000, 850]
        assert {"source_path", "delay_embedding", "era5_slice_path", "date_processed", "save_data_matrix"} <= set(g)
        dims = {"space": m, "components": 5, "time": da.shape[1]}
        ids = {}
        for d, n in dims.items():                              # one dimension scale each
            shape, dt, vdims = r.variables[d]
            a = r.attrs(d, raw=True)
            assert shape == (n,) and vdims == (d,)
            assert a["CLASS"] == "DIMENSION_SCALE" and a["NAME"] == d      # a coordinate variable, not a placeholder
            assert "REFERENCE_LIST" in r.attr_names(d)                      # variables are attached to it
            ids[d] = a["_Netcdf4Dimid"]
        assert sorted(ids.values()) == [0, 1, 2]
        for v, vd in {"U": ("space", "components"), "s": ("components",), "V": ("components", "time"),
                      "X": ("space", "time"), "X_mean": ("space",), "X_std": ("space",)}.items():
            shape, dt, vdims = r.variables[v]
            assert vdims == vd and dt == np.float32 or v in ("X_mean", "X_std", "X")
            assert vdims == vd and "DIMENSION_LIST" in r.attr_names(v)
            a = r.attrs(v)
            if "space" in vd:
                assert a["coordinates"] == "delay latitude level longitude original_variable"
            else:
                assert "coordinates" not in a
        for cname, kind in {"components": "i", "space": "i", "delay": "i", "level": "i",
                            "latitude": "f", "longitude": "f"}.items():
            shape, dt, vdims = r.variables[cname]
            assert dt.kind == kind and dt.itemsize == 8, (cname, dt)
        assert r.variables["original_variable"][1] == "str" or isinstance(r.variables["original_variable"][1], str)
        assert r.variables["time"][1].kind in "if" and "units" in r.attrs("time") \
            and " since " in r.attrs("time")["units"]
        for cname in ("delay", "level", "latitude", "longitude", "original_variable"):
            assert r.variables[cname][2] == ("space",) and "DIMENSION_LIST" in r.attr_names(cname)

    # a dimension without a coordinate variable gets netCDF-C's placeholder scale
    bare = Dataset(coords={}, attrs={"title": "t"})
    bare["field"] = DataArray(np.zeros((3, 4), dtype=np.float32), ("a", "b"), {}, {})
    path = str(project_root / "bare.nc")
    io_netcdf.to_netcdf(bare, path)
    with hdf5_lite.Reader(path) as r:
        for d, n in (("a", 3), ("b", 4)):
            a = r.attrs(d, raw=True)
            assert a["CLASS"] == "DIMENSION_SCALE" and r.is_placeholder_dimension(d)
            assert a["NAME"] == f"This is a netCDF dimension but not a netCDF variable.{n:>10d}"
            assert "_Netcdf4Dimid" in a

    # the reader classifies coordinates from the `coordinates` attributes (as xarray's decoder
    # does), and arrives at the README's schema: coordinates vs data variables
    back = io_netcdf.open_dataset(p["save_path"])
    assert sorted(back.coords) == sorted(["components", "space", "time", "delay", "level", "latitude", "longitude",
                                          "original_variable"])
    assert sorted(back.data_vars) == ["U", "V", "X", "X_mean", "X_std", "s"]
    assert sorted(back["U"].coords) == sorted(["components", "space", "delay", "level", "latitude", "longitude",
                                               "original_variable"])
    assert "coordinates" not in back["U"].attrs
    # a variable NOT named in any `coordinates` attribute stays a data variable even if it is
    # called like a row label and lives on `space`
    odd = Dataset(coords={"space": res.coords["space"]}, attrs={})
    odd["delay"] = DataArray(np.arange(m, dtype=np.int64), ("space",), {}, {})
    odd["U"] = DataArray(np.zeros((m, 2), dtype=np.float32), ("space", "components"), {}, {"coordinates": "space"})
    path2 = str(project_root / "odd.nc")
    io_netcdf.to_netcdf(odd, path2)
    assert "delay" in io_netcdf.open_dataset(path2).data_vars


@pytest.mark.parametrize("start,step_h,delta_h,n", [("2019-01-01T00", 1, 1, 100), ("2019-03-05T06", 6, 6, 40),
                                                     ("2019-01-01T03", 3, 3, 17), ("2019-01-01T00", 24, 24, 9),
                                                     ("2019-01-01T00", 1, 1, 1),
                                                     # off the fast path: coarser delta, samples off the bins, a gap
                                                     ("2019-01-01T00", 1, 6, 100), ("2019-01-01T01", 2, 2, 30),
                                                     ("2019-01-01T00", 1, 1, -50)])
def test_resample_fast_path_equals_pandas(start, step_h, delta_h, n):
    """nearest_resample_index answers uniformly sampled, bin-aligned time axes without pandas:
    labels and indices must be what pandas' resample(...).nearest() gives (ref slice_tools.py:139)."""
    from datetime import timedelta

    from dmd_era5_amd.slice_tools import nearest_resample_index

    t = np.datetime64(start, "ns") + np.arange(abs(n)) * np.timedelta64(step_h, "h")
    if n < 0:
        t = np.delete(t, 7)                                   # a missing sample
    a = nearest_resample_index(t, timedelta(hours=delta_h))
    b = nearest_resample_index(t, timedelta(hours=delta_h), _force_pandas=True)
    assert a[0].dtype == b[0].dtype and np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
