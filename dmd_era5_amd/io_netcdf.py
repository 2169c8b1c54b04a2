"""NetCDF I/O for the two files on the path: the ERA5 slice written by the reference's
``era5_download`` (ref: src/dmd_era5/era5_download/era5_download.py:104-115 -- variables
``(time, level, latitude, longitude)``, attrs source_path / variables / levels / ...) and
the SVD result ``data/era5_svd/*.nc`` (ref: era5_svd.py:434, README.md:97-119).

Backends:
  1. :mod:`dmd_era5_amd.hdf5_lite` -- our ctypes binding of libhdf5 -> NETCDF4/HDF5, the
     reference's format (datasets + dimension scales with ``_Netcdf4Dimid`` + attributes +
     ``_NCProperties``: the objects netCDF-C looks for; CF ``coordinates`` attributes as xarray
     writes them), with lazy variables and time-slab reads for the streaming ingest.  The default
     and the tested one (libhdf5 1.10.6 under /opt/conda/lib in this image).  No netCDF library
     is importable here, so what netCDF-C / xarray make of these files is **interop parity
     unpinned**; tests/test_host_mirror.py pins the HDF5 structure instead;
  2. ``scipy.io.netcdf_file`` -> NetCDF-3 64-bit-offset, when no HDF5 library can be found.
     Same variables / dimensions / attributes; limits of the classic format apply (< 4 GiB per
     variable, no string variables: ``original_variable`` is stored as an index + flag_meanings);
  3. ``netCDF4`` -- only on request (``DMDX_NETCDF_BACKEND=netcdf4``): that path has never run
     (the module is not importable in this image).
``DMDX_NETCDF_BACKEND`` = hdf5 | scipy | netcdf4 forces one.
"""
from __future__ import annotations

import os

import numpy as np

from . import hdf5_lite
from .labeled import Coord, DataArray, Dataset, LazyArray

_EPOCH = np.datetime64("1970-01-01T00:00:00", "ns")
TIME_UNITS = "hours since 1970-01-01 00:00:00"


def _have_netcdf4() -> bool:
    try:
        import netCDF4  # noqa: F401

        return True
    except Exception:
        return False


def _encode_time(t: np.ndarray) -> np.ndarray:
    return ((t.astype("datetime64[ns]") - _EPOCH) / np.timedelta64(1, "h")).astype(np.float64)


def _decode_time(v: np.ndarray, units: str) -> np.ndarray:
    unit, _, ref = units.partition(" since ")
    step = {"hours": "h", "hour": "h", "days": "D", "day": "D", "minutes": "m", "seconds": "s"}[unit.strip()]
    ref64 = np.datetime64(ref.strip().replace(" ", "T"), "ns")
    secs = {"h": 3600.0, "D": 86400.0, "m": 60.0, "s": 1.0}[step]
    return ref64 + np.round(np.asarray(v, dtype=np.float64) * secs * 1e9).astype("timedelta64[ns]")


def _attr_out(v, classic: bool = False):
    """Attribute value as NetCDF can hold it (lists of strings -> comma separated, as the
    reference's own round trip through NetCDF yields; classic format has no int64)."""
    if isinstance(v, (bool, np.bool_)):
        v = int(v)
    if isinstance(v, (list, tuple)):
        if all(isinstance(x, str) for x in v):
            return ",".join(v) if len(v) != 1 else v[0]
        v = np.asarray(v)
    if classic:
        if isinstance(v, (int, np.integer)):
            return np.int32(v)
        if isinstance(v, np.ndarray) and v.dtype == np.int64:
            return v.astype(np.int32)
    return v


def _safe_netcdf_file():
    """scipy's netcdf_file mirrors global attributes into instance attributes, so a file
    with a global attribute called ``variables`` (every file on this path has one) would
    overwrite its own variable table while being read.  Keep them in ``_attributes`` only."""
    from scipy.io import netcdf_file

    class _NetcdfFile(netcdf_file):
        def _read_gatt_array(self):
            for k, v in self._read_att_array().items():
                self._attributes[k] = v

    return _NetcdfFile


# ----------------------------------------------------------------------------- write
def _backend() -> str:
    forced = os.environ.get("DMDX_NETCDF_BACKEND", "").lower()
    if forced in ("netcdf4", "hdf5", "scipy"):
        return forced
    return "hdf5" if hdf5_lite.available() else "scipy"


def _coordinates_attr(ds: Dataset, da) -> str | None:
    """The CF ``coordinates`` attribute xarray's ``to_netcdf`` (reference era5_svd.py:434) puts on
    a variable: the non-index coordinates that live on its dimensions (level, latitude,
    longitude, original_variable, delay for everything with a ``space`` dimension), sorted, space
    separated.  A reader (xarray's decoder, ours) takes exactly these for coordinates."""
    names = sorted(n for n, c in ds.coords.items()
                   if c.dims != (n,) and getattr(c.values, "ndim", 1) == 1 and set(c.dims) <= set(da.dims))
    return " ".join(names) if names else None


def to_netcdf(ds: Dataset, path: str) -> str:
    """Write ``ds``; returns the backend used ("netCDF4", "hdf5-lite" or "scipy-netcdf3")."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    be = _backend()
    if be == "netcdf4":
        _write_netcdf4(ds, path)
        return "netCDF4"
    if be == "hdf5":
        _write_hdf5(ds, path)
        return "hdf5-lite"
    _write_scipy(ds, path)
    return "scipy-netcdf3"


def _write_hdf5(ds: Dataset, path: str) -> None:
    with hdf5_lite.Writer(path) as w:
        for name, c in ds.coords.items():
            vals, attrs = c.values, {}
            if np.issubdtype(vals.dtype, np.datetime64):
                vals = np.round(_encode_time(vals)).astype(np.int64)
                attrs = {"units": TIME_UNITS, "calendar": "proleptic_gregorian"}
            elif vals.ndim != 1:
                continue  # (m, 3) space labels: stored as level / latitude / longitude
            w.dataset(name, vals, c.dims, attrs)
        for name, da in ds.data_vars.items():
            attrs = {k: v for k, v in da.attrs.items()}
            cattr = _coordinates_attr(ds, da)
            if cattr:
                attrs["coordinates"] = cattr
            w.dataset(name, np.asarray(da.values), da.dims, attrs)
        w.attrs(None, dict(ds.attrs))


def _prepared_vars(ds: Dataset):
    """(name, dims, array, attrs) for coordinates then data variables, NetCDF-storable."""
    out = []
    ov_names = None
    for name, c in ds.coords.items():
        vals, attrs = c.values, {}
        if name == "time" or np.issubdtype(vals.dtype, np.datetime64):
            vals, attrs = _encode_time(vals), {"units": TIME_UNITS, "calendar": "proleptic_gregorian"}
        elif vals.dtype.kind in "UOS":  # original_variable: strings -> index into an attribute
            ov_names = list(dict.fromkeys(vals.tolist()))
            lut = {v: i for i, v in enumerate(ov_names)}
            vals = np.array([lut[v] for v in vals.tolist()], dtype=np.int32)
            attrs = {"flag_meanings": " ".join(ov_names), "comment": "index into flag_meanings"}
        elif vals.ndim != 1:
            continue  # the (m, 3) space labels are stored as level/latitude/longitude instead
        out.append((name, c.dims, vals, attrs))
    for name, da in ds.data_vars.items():
        attrs = dict(da.attrs)
        cattr = _coordinates_attr(ds, da)
        if cattr:
            attrs["coordinates"] = cattr
        out.append((name, da.dims, np.asarray(da.values), attrs))
    return out


def _write_scipy(ds: Dataset, path: str) -> None:
    netcdf_file = _safe_netcdf_file()
    with netcdf_file(path, "w", version=2) as f:
        sizes = ds.sizes
        for d, nlen in sizes.items():
            f.createDimension(d, int(nlen))
        for name, dims, vals, attrs in _prepared_vars(ds):
            if vals.dtype == np.int64:
                vals = vals.astype(np.int32) if np.abs(vals).max(initial=0) < 2**31 else vals.astype(np.float64)
            if vals.dtype == np.bool_:
                vals = vals.astype(np.int8)
            var = f.createVariable(name, vals.dtype, dims)
            var[...] = vals
            # (attribute names such as "variables" would shadow scipy's own members if set
            # with setattr; its writer serialises the _attributes dicts)
            for k, v in attrs.items():
                var._attributes[k] = _attr_out(v, classic=True)
        for k, v in ds.attrs.items():
            f._attributes[k] = _attr_out(v, classic=True)
        f._attributes["dmdx_backend"] = "scipy-netcdf3"


def _write_netcdf4(ds: Dataset, path: str) -> None:
    import netCDF4

    with netCDF4.Dataset(path, "w", format="NETCDF4") as f:
        for d, nlen in ds.sizes.items():
            f.createDimension(d, int(nlen))
        for name, dims, vals, attrs in _prepared_vars(ds):
            var = f.createVariable(name, vals.dtype, dims)
            var[...] = vals
            for k, v in attrs.items():
                var.setncattr(k, _attr_out(v))
        for k, v in ds.attrs.items():
            f.setncattr(k, _attr_out(v))


# ----------------------------------------------------------------------------- read
def open_dataset(path: str) -> Dataset:
    """Read a NetCDF file into a :class:`Dataset` (coordinates = 1-D variables named like
    their dimension, plus per-row ``space`` coordinates of an SVD result)."""
    with open(path, "rb") as fh:
        magic = fh.read(4)
    if magic[:3] == b"CDF":
        return _read_scipy(path)
    be = _backend()
    if be == "netcdf4":
        return _read_netcdf4(path)
    if hdf5_lite.available():
        return _read_hdf5(path)
    raise RuntimeError(
        f"{path} is a NETCDF4/HDF5 file and neither the netCDF4 module nor libhdf5 was found "
        "(set DMDX_HDF5_LIB to the directory holding libhdf5.so).")


_ROW_COORDS = ("level", "latitude", "longitude", "original_variable", "delay")


def _assemble(raw: dict, dimsizes: dict, gattrs: dict) -> Dataset:
    """Variables -> Dataset.  Coordinates are the 1-D variables named like their dimension plus
    whatever the variables' CF ``coordinates`` attributes name (what xarray writes and decodes by);
    files without such attributes (older results of this package) fall back to the fixed list of
    per-row labels of an SVD result."""
    def _s(v):
        return v.decode() if isinstance(v, bytes) else str(v)

    listed = set()
    for _, (_, _, attrs) in raw.items():
        if "coordinates" in attrs:
            listed.update(_s(attrs["coordinates"]).split())
    if "coordinates" in gattrs:
        listed.update(_s(gattrs["coordinates"]).split())
    coords, data = {}, {}
    for name, (dims, vals, attrs) in raw.items():
        if "units" in attrs and " since " in str(attrs["units"]) and not isinstance(vals, LazyArray):
            vals = _decode_time(vals, attrs["units"].decode() if isinstance(attrs["units"], bytes) else attrs["units"])
        if name == "original_variable" and "flag_meanings" in attrs:
            fm = attrs["flag_meanings"]
            names = (fm.decode() if isinstance(fm, bytes) else fm).split(" ")
            vals = np.array([names[i] for i in vals])
        if listed:
            is_coord = (dims == (name,)) or name in listed
        else:
            is_coord = (dims == (name,)) or (name in _ROW_COORDS and dims == ("space",) and "space" in dimsizes
                                             and "U" in raw)
        attrs = {k: v for k, v in attrs.items() if k != "coordinates"}
        (coords if is_coord else data)[name] = (dims, vals, attrs)
    cds = {k: Coord(d, v) for k, (d, v, _) in coords.items()}
    ds = Dataset(coords=cds, attrs={k: v for k, v in gattrs.items() if k != "coordinates"})
    for k, (d, v, a) in data.items():
        ds[k] = DataArray(v, d, {c: cds[c] for c in cds if set(cds[c].dims) <= set(d)}, a)
    return ds


# variables above this size stay on disk until used (and can be streamed in time slabs)
LAZY_BYTES = 64 << 20


def _read_hdf5(path: str) -> Dataset:
    r = hdf5_lite.Reader(path)
    raw, sizes = {}, {}
    for name, (shape, dt, dims) in r.variables.items():
        if r.is_placeholder_dimension(name):
            sizes[name] = shape[0]
            continue
        attrs = r.attrs(name)
        nbytes = int(np.prod(shape)) * (dt.itemsize if hasattr(dt, "itemsize") else 8)
        if nbytes > LAZY_BYTES and not isinstance(dt, str) and len(shape) >= 2:
            vals = LazyArray(shape, dt, lambda n=name: r.read(n),
                             lambda a, b, out=None, n=name: r.read_slab(n, a, b, out),
                             lambda st, ct, out=None, n=name: r.read_box(n, st, ct, out))
        else:
            vals = r.read(name)
        raw[name] = (dims, vals, attrs)
        for d, n in zip(dims, shape):
            sizes[d] = n
    ds = _assemble(raw, sizes, r.attrs(None))
    ds._reader = r  # keeps the file open for the lazy variables
    return ds


def _clean_attrs(items) -> dict:
    out = {}
    for k, v in items:
        if isinstance(v, bytes):
            v = v.decode()
        elif isinstance(v, np.ndarray) and v.ndim == 0:
            v = v.item()
        out[k] = v
    return out


def _read_scipy(path: str) -> Dataset:
    netcdf_file = _safe_netcdf_file()
    with netcdf_file(path, "r", mmap=False) as f:
        raw = {}
        for name, var in f.variables.items():
            attrs = _clean_attrs((k, v) for k, v in var._attributes.items())
            raw[name] = (tuple(var.dimensions), np.array(var.data, copy=True).astype(var.data.dtype.newbyteorder("=")),
                         attrs)
        sizes = {k: int(v) if v is not None else 0 for k, v in f.dimensions.items()}
        g = _clean_attrs((k, v) for k, v in f._attributes.items())
    return _assemble(raw, sizes, g)


def _read_netcdf4(path: str) -> Dataset:
    import netCDF4

    with netCDF4.Dataset(path, "r") as f:
        f.set_auto_maskandscale(False)
        raw = {}
        for name, var in f.variables.items():
            attrs = {k: var.getncattr(k) for k in var.ncattrs()}
            raw[name] = (tuple(var.dimensions), np.asarray(var[...]), attrs)
        sizes = {k: len(v) for k, v in f.dimensions.items()}
        g = {k: f.getncattr(k) for k in f.ncattrs()}
    return _assemble(raw, sizes, g)
