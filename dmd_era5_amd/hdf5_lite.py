"""A small ctypes binding of the HDF5 C library -- just enough to read and write the
NETCDF4 files on the hot path without netCDF4 / h5py / xarray:

* read the ERA5 slice the reference's downloader writes with
  ``ds.to_netcdf(path, format="NETCDF4")`` (ref: src/dmd_era5/era5_download/era5_download.py:104-115):
  variables ``(time, level, latitude, longitude)``, coordinate variables, global attributes --
  whole arrays or **time slabs** (hyperslab reads, used by the streaming ingest);
* write ``data/era5_svd/*.nc`` (ref: src/dmd_era5/era5_svd/era5_svd.py:434) as an HDF5 file
  that netCDF-C / xarray open as NETCDF4: one dataset per variable, coordinate variables
  registered as HDF5 dimension scales and attached to every dimension that uses them,
  text attributes as fixed-length strings, string variables as variable-length strings.

libhdf5 (>= 1.10) and libhdf5_hl are looked up in ``$DMDX_HDF5_LIB``, the usual loader paths and
``/opt/conda/lib`` (where this image keeps 1.10.6).  ``available()`` tells whether they were found.
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import os
import sys

import numpy as np

# threads used to pread contiguous datasets (0: always go through H5Dread)
RAW_READ_THREADS = int(os.environ.get("DMDX_RAW_READ_THREADS", "8"))

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int

_H5F_ACC_RDONLY, _H5F_ACC_TRUNC = 0, 2
_H5P_DEFAULT = 0
_H5S_ALL = 0
_H5S_SELECT_SET = 0
_H5T_STRING, _H5T_INTEGER, _H5T_FLOAT = 3, 0, 1
_H5T_VARIABLE = C.c_size_t(-1).value
_H5_INDEX_NAME, _H5_ITER_INC = 0, 0

_lib = None
_hl = None


def _candidates(stem: str):
    env = os.environ.get("DMDX_HDF5_LIB")
    if env:
        d = env if os.path.isdir(env) else os.path.dirname(env)
        yield os.path.join(d, f"lib{stem}.so")
    found = ctypes.util.find_library(stem)
    if found:
        yield found
    for d in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib64", "/usr/local/lib"):
        yield os.path.join(d, f"lib{stem}.so")
        for suffix in ("103", "200", "310"):
            yield os.path.join(d, f"lib{stem}.so.{suffix}")


def _load():
    global _lib, _hl
    if _lib is not None:
        return _lib, _hl
    for path in _candidates("hdf5"):
        try:
            lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
        except OSError:
            continue
        hl = None
        for hp in [os.path.join(os.path.dirname(path), "libhdf5_hl.so")] + list(_candidates("hdf5_hl")):
            try:
                hl = C.CDLL(hp)
                break
            except OSError:
                continue
        if hl is None:
            continue
        _bind(lib, hl)
        lib.H5open()
        lib.H5Eset_auto2(hid_t(0), None, None)  # errors are reported through return codes
        _lib, _hl = lib, hl
        return _lib, _hl
    raise OSError("libhdf5 / libhdf5_hl not found (set DMDX_HDF5_LIB)")


def available() -> bool:
    try:
        _load()
        return True
    except OSError:
        return False


def _bind(lib, hl):
    def f(l, name, res, *args):
        fn = getattr(l, name)
        fn.restype, fn.argtypes = res, list(args)

    p_hs = C.POINTER(hsize_t)
    f(lib, "H5open", herr_t)
    f(lib, "H5get_libversion", herr_t, C.POINTER(C.c_uint), C.POINTER(C.c_uint), C.POINTER(C.c_uint))
    f(lib, "H5Eset_auto2", herr_t, hid_t, C.c_void_p, C.c_void_p)
    f(lib, "H5Fcreate", hid_t, C.c_char_p, C.c_uint, hid_t, hid_t)
    f(lib, "H5Fopen", hid_t, C.c_char_p, C.c_uint, hid_t)
    f(lib, "H5Fclose", herr_t, hid_t)
    f(lib, "H5Screate_simple", hid_t, C.c_int, p_hs, p_hs)
    f(lib, "H5Screate", hid_t, C.c_int)
    f(lib, "H5Sclose", herr_t, hid_t)
    f(lib, "H5Sget_simple_extent_ndims", C.c_int, hid_t)
    f(lib, "H5Sget_simple_extent_dims", C.c_int, hid_t, p_hs, p_hs)
    f(lib, "H5Sselect_hyperslab", herr_t, hid_t, C.c_int, p_hs, p_hs, p_hs, p_hs)
    f(lib, "H5Dcreate2", hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t)
    f(lib, "H5Dopen2", hid_t, hid_t, C.c_char_p, hid_t)
    f(lib, "H5Dclose", herr_t, hid_t)
    f(lib, "H5Dget_space", hid_t, hid_t)
    f(lib, "H5Dget_type", hid_t, hid_t)
    f(lib, "H5Dread", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    f(lib, "H5Dwrite", herr_t, hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p)
    f(lib, "H5Dvlen_reclaim", herr_t, hid_t, hid_t, hid_t, C.c_void_p)
    f(lib, "H5Acreate2", hid_t, hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t)
    f(lib, "H5Awrite", herr_t, hid_t, hid_t, C.c_void_p)
    f(lib, "H5Aopen_by_idx", hid_t, hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, hid_t, hid_t)
    f(lib, "H5Aget_name", C.c_ssize_t, hid_t, C.c_size_t, C.c_char_p)
    f(lib, "H5Aget_type", hid_t, hid_t)
    f(lib, "H5Aget_space", hid_t, hid_t)
    f(lib, "H5Aread", herr_t, hid_t, hid_t, C.c_void_p)
    f(lib, "H5Aclose", herr_t, hid_t)
    f(lib, "H5Tcopy", hid_t, hid_t)
    f(lib, "H5Tset_size", herr_t, hid_t, C.c_size_t)
    f(lib, "H5Tget_size", C.c_size_t, hid_t)
    f(lib, "H5Tget_class", C.c_int, hid_t)
    f(lib, "H5Tget_sign", C.c_int, hid_t)
    f(lib, "H5Tget_order", C.c_int, hid_t)
    f(lib, "H5Dget_offset", C.c_uint64, hid_t)
    f(lib, "H5Dget_create_plist", hid_t, hid_t)
    f(lib, "H5Pget_layout", C.c_int, hid_t)
    f(lib, "H5Pget_nfilters", C.c_int, hid_t)
    f(lib, "H5Pclose", herr_t, hid_t)
    f(lib, "H5Tis_variable_str", C.c_int, hid_t)
    f(lib, "H5Tclose", herr_t, hid_t)
    f(lib, "H5Lget_name_by_idx", C.c_ssize_t, hid_t, C.c_char_p, C.c_int, C.c_int, hsize_t, C.c_char_p,
      C.c_size_t, hid_t)
    f(lib, "H5Iget_name", C.c_ssize_t, hid_t, C.c_char_p, C.c_size_t)
    f(lib, "H5Oopen", hid_t, hid_t, C.c_char_p, hid_t)
    f(lib, "H5Oclose", herr_t, hid_t)
    f(lib, "H5Iget_type", C.c_int, hid_t)
    f(hl, "H5DSset_scale", herr_t, hid_t, C.c_char_p)
    f(hl, "H5DSattach_scale", herr_t, hid_t, hid_t, C.c_uint)
    f(hl, "H5DSis_scale", C.c_int, hid_t)
    f(hl, "H5DSiterate_scales", herr_t, hid_t, C.c_uint, C.POINTER(C.c_int), C.c_void_p, C.c_void_p)


def _native(lib, name: str) -> int:
    return hid_t.in_dll(lib, name).value


_NP2H5 = {
    "float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g",
    "int8": "H5T_NATIVE_SCHAR_g", "uint8": "H5T_NATIVE_UCHAR_g",
    "int16": "H5T_NATIVE_SHORT_g", "uint16": "H5T_NATIVE_USHORT_g",
    "int32": "H5T_NATIVE_INT_g", "uint32": "H5T_NATIVE_UINT_g",
    "int64": "H5T_NATIVE_LLONG_g", "uint64": "H5T_NATIVE_ULLONG_g",
}


def _h5type(lib, dtype) -> int:
    return _native(lib, _NP2H5[np.dtype(dtype).name])


def _np_dtype_of(lib, tid: int):
    cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
    if cls == _H5T_FLOAT:
        return np.dtype(f"f{size}")
    if cls == _H5T_INTEGER:
        return np.dtype(("i" if lib.H5Tget_sign(tid) else "u") + str(size))
    if cls == _H5T_STRING:
        return "vstr" if lib.H5Tis_variable_str(tid) > 0 else np.dtype(f"S{size}")
    return None


def _dims(*d):
    return (hsize_t * len(d))(*d)


def _factorize(flat: np.ndarray):
    """(distinct values, index of each element) of a large 1-D unicode array.  Label coordinates
    are long runs of equal values (np.repeat / np.tile of a few names): only the run heads are
    sorted; without run structure it is a plain np.unique."""
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    if change.size > flat.size // 8:
        return np.unique(flat, return_inverse=True)
    heads = np.concatenate([[0], change])
    uniq, ids = np.unique(flat[heads], return_inverse=True)
    lengths = np.diff(np.concatenate([heads, [flat.size]]))
    return uniq, np.repeat(ids, lengths)


_host = None


def _host_lib():
    """libdmdx_host.so (dmd_era5_amd/csrc/hostutil.c) or None when it has not been built."""
    global _host
    if _host is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdmdx_host.so")
        try:
            lib = C.CDLL(path)
            lib.dmdx_host_vlen_maxlen.restype = C.c_size_t
            lib.dmdx_host_vlen_maxlen.argtypes = [C.c_void_p, C.c_size_t]
            lib.dmdx_host_vlen_to_fixed.restype = C.c_int
            lib.dmdx_host_vlen_to_fixed.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
            _host = lib
        except OSError:
            _host = False
    return _host or None


# ======================================================================================
# writer
# ======================================================================================
class Writer:
    """``with Writer(path) as w: w.dataset(name, array, dims); w.attrs(None, {...})``.
    A dataset whose name equals its single dimension becomes that dimension's scale."""

    def __init__(self, path: str):
        self.lib, self.hl = _load()
        self.fid = self.lib.H5Fcreate(path.encode(), _H5F_ACC_TRUNC, _H5P_DEFAULT, _H5P_DEFAULT)
        if self.fid < 0:
            raise OSError(f"H5Fcreate failed for {path}")
        self._dsets: dict[str, int] = {}
        self._vardims: dict[str, tuple] = {}
        self._has_ncproperties = False

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def dataset(self, name: str, arr: np.ndarray, dims: tuple, attrs: dict | None = None) -> None:
        lib = self.lib
        arr = np.asarray(arr)
        if arr.dtype.kind in "UO":
            self._vstr_dataset(name, arr, dims)
        else:
            if arr.dtype == np.bool_:
                arr = arr.astype(np.int8)
            arr = np.ascontiguousarray(arr)
            tid = _h5type(lib, arr.dtype)
            sid = lib.H5Screate_simple(arr.ndim, _dims(*arr.shape), None) if arr.ndim else lib.H5Screate(0)
            did = lib.H5Dcreate2(self.fid, name.encode(), tid, sid, _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT)
            if did < 0:
                raise OSError(f"H5Dcreate2 failed for {name}")
            if lib.H5Dwrite(did, tid, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, arr.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError(f"H5Dwrite failed for {name}")
            lib.H5Sclose(sid)
            self._dsets[name] = did
        self._vardims[name] = tuple(dims)
        if attrs:
            self.attrs(name, attrs)

    def _vstr_dataset(self, name, arr, dims):
        lib = self.lib
        flat = arr.ravel()
        # label coordinates (`original_variable` over 10^6..10^7 space points) repeat a handful of
        # values: one C string per distinct value and a numpy gather of their addresses, instead
        # of one Python bytes object per element
        if flat.dtype.kind == "U":
            uniq, inv = np.unique(flat, return_inverse=True) if flat.size < 4096 else _factorize(flat)
        else:
            uniq, inv = np.unique(np.array([str(x) for x in flat.tolist()]), return_inverse=True)
        keep = [C.create_string_buffer(str(u).encode()) for u in uniq.tolist()]
        addr = np.array([C.addressof(b) for b in keep], dtype=np.uint64)
        ptrs = np.ascontiguousarray(addr[np.asarray(inv).ravel()]) if flat.size else np.zeros(0, np.uint64)
        bufs = ptrs.ctypes.data_as(C.c_void_p)
        tid = lib.H5Tcopy(_native(lib, "H5T_C_S1_g"))
        lib.H5Tset_size(tid, _H5T_VARIABLE)
        sid = lib.H5Screate_simple(arr.ndim, _dims(*arr.shape), None)
        did = lib.H5Dcreate2(self.fid, name.encode(), tid, sid, _H5P_DEFAULT, _H5P_DEFAULT, _H5P_DEFAULT)
        if did < 0 or lib.H5Dwrite(did, tid, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, bufs) < 0:
            raise OSError(f"writing string dataset {name} failed")
        lib.H5Sclose(sid)
        lib.H5Tclose(tid)
        self._dsets[name] = did

    def attrs(self, name: str | None, attrs: dict) -> None:
        lib = self.lib
        loc = self.fid if name is None else self._dsets[name]
        for key, val in attrs.items():
            if name is None and key == "_NCProperties":
                self._has_ncproperties = True
            if isinstance(val, (bool, np.bool_)):
                val = int(val)
            if isinstance(val, (list, tuple)) and len(val) == 1 and isinstance(val[0], str):
                val = val[0]  # what netCDF4-python stores for a one-element list of str
            if isinstance(val, str):
                raw = val.encode() or b" "
                tid = lib.H5Tcopy(_native(lib, "H5T_C_S1_g"))
                lib.H5Tset_size(tid, len(raw))
                sid = lib.H5Screate(0)
                aid = lib.H5Acreate2(loc, key.encode(), tid, sid, _H5P_DEFAULT, _H5P_DEFAULT)
                lib.H5Awrite(aid, tid, C.c_char_p(raw))
                lib.H5Tclose(tid)
            elif isinstance(val, (list, tuple)) and val and all(isinstance(x, str) for x in val):
                enc = [x.encode() for x in val]
                bufs = (C.c_char_p * len(enc))(*enc)
                tid = lib.H5Tcopy(_native(lib, "H5T_C_S1_g"))
                lib.H5Tset_size(tid, _H5T_VARIABLE)
                sid = lib.H5Screate_simple(1, _dims(len(enc)), None)
                aid = lib.H5Acreate2(loc, key.encode(), tid, sid, _H5P_DEFAULT, _H5P_DEFAULT)
                lib.H5Awrite(aid, tid, bufs)
                lib.H5Tclose(tid)
            else:
                a = np.ascontiguousarray(np.asarray(val))
                if a.dtype.kind not in "iuf":
                    a = np.asarray(str(val))
                    self.attrs(name, {key: str(val)})
                    continue
                tid = _h5type(lib, a.dtype)
                # netCDF-C's layout (nc4hdf.c): every numeric attribute is a 1-D array, a single
                # value included (np.ascontiguousarray has made it one); only its own bookkeeping
                # attribute _Netcdf4Dimid sits in a scalar dataspace
                sid = lib.H5Screate(0) if key == "_Netcdf4Dimid" else lib.H5Screate_simple(1, _dims(a.size), None)
                aid = lib.H5Acreate2(loc, key.encode(), tid, sid, _H5P_DEFAULT, _H5P_DEFAULT)
                lib.H5Awrite(aid, tid, a.ctypes.data_as(C.c_void_p))
            if aid < 0:
                raise OSError(f"writing attribute {key} failed")
            lib.H5Aclose(aid)
            lib.H5Sclose(sid)

    def close(self) -> None:
        """Register dimension scales, attach them, close everything."""
        lib, hl = self.lib, self.hl
        if self.fid < 0:
            return
        dim_names = []
        for dims in self._vardims.values():
            for dn in dims:
                if dn not in dim_names:
                    dim_names.append(dn)
        sizes = {}
        for v, dims in self._vardims.items():
            sp = lib.H5Dget_space(self._dsets[v])
            nd = lib.H5Sget_simple_extent_ndims(sp)
            ext = (hsize_t * max(nd, 1))()
            lib.H5Sget_simple_extent_dims(sp, ext, None)
            lib.H5Sclose(sp)
            for i, dn in enumerate(dims):
                sizes[dn] = int(ext[i])
        for dn in dim_names:  # dimensions without a coordinate variable: netCDF's placeholder scale
            if dn not in self._dsets:
                fill = np.zeros(sizes[dn], dtype=np.float32)
                self.dataset(dn, fill, (dn,))
                self._vardims.pop(dn)
                hl.H5DSset_scale(self._dsets[dn],
                                 f"This is a netCDF dimension but not a netCDF variable.{sizes[dn]:>10d}".encode())
            elif self._vardims.get(dn) == (dn,):
                hl.H5DSset_scale(self._dsets[dn], dn.encode())
        for v, dims in self._vardims.items():
            if dims == (v,):
                continue
            for i, dn in enumerate(dims):
                if dn in self._dsets:
                    hl.H5DSattach_scale(self._dsets[v], self._dsets[dn], i)
        # what netCDF-C adds to its own files: the dimension id on every dimension scale (it
        # orders the dimensions by it instead of by creation order) and the provenance string
        for i, dn in enumerate(dim_names):
            if dn in self._dsets:
                self.attrs(dn, {"_Netcdf4Dimid": np.int32(i)})
        if not self._has_ncproperties:
            maj, mnr, rel = C.c_uint(0), C.c_uint(0), C.c_uint(0)
            lib.H5get_libversion(C.byref(maj), C.byref(mnr), C.byref(rel))
            self.attrs(None, {"_NCProperties": f"version=2,dmd_era5_amd_hdf5_lite=1,hdf5={maj.value}.{mnr.value}.{rel.value}"})
        for did in self._dsets.values():
            lib.H5Dclose(did)
        lib.H5Fclose(self.fid)
        self.fid = -1


# ======================================================================================
# reader
# ======================================================================================
_SCALE_CB = C.CFUNCTYPE(herr_t, hid_t, C.c_uint, hid_t, C.c_void_p)
_SKIP_ATTRS = {"DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates",
               "_NCProperties", "_nc3_strict"}


class Reader:
    """``r = Reader(path); r.variables -> {name: (shape, dtype, dims)}; r.read(name);
    r.read_slab(name, t0, t1); r.attrs(name or None)``."""

    def __init__(self, path: str):
        self.lib, self.hl = _load()
        self.path = path
        self.fid = self.lib.H5Fopen(path.encode(), _H5F_ACC_RDONLY, _H5P_DEFAULT)
        if self.fid < 0:
            raise OSError(f"cannot open {path} as HDF5")
        self.variables: dict[str, tuple] = {}
        self.raw_offset: dict[str, int] = {}   # name -> file address of a contiguous native dataset
        self._fd = -1
        self._pool = None
        self._scan()

    def close(self):
        if self.fid >= 0:
            self.lib.H5Fclose(self.fid)
            self.fid = -1
        if self._fd >= 0:
            os.close(self._fd)
            self._fd = -1
        if self._pool is not None:
            self._pool.shutdown(wait=True)
            self._pool = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _names(self):
        lib, out, i = self.lib, [], 0
        buf = C.create_string_buffer(1024)
        while True:
            n = lib.H5Lget_name_by_idx(self.fid, b".", _H5_INDEX_NAME, _H5_ITER_INC, i, buf, 1024, _H5P_DEFAULT)
            if n < 0:
                break
            out.append(buf.value.decode())
            i += 1
        return out

    def _scan(self):
        lib, hl = self.lib, self.hl
        for name in self._names():
            oid = lib.H5Oopen(self.fid, name.encode(), _H5P_DEFAULT)
            is_dset = lib.H5Iget_type(oid) == 5  # H5I_DATASET
            lib.H5Oclose(oid)
            if not is_dset:
                continue
            did = lib.H5Dopen2(self.fid, name.encode(), _H5P_DEFAULT)
            sp = lib.H5Dget_space(did)
            nd = lib.H5Sget_simple_extent_ndims(sp)
            ext = (hsize_t * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sp, ext, None)
            shape = tuple(int(ext[i]) for i in range(nd))
            tid = lib.H5Dget_type(did)
            dt = _np_dtype_of(lib, tid)
            dims = []
            for i in range(nd):
                found = []

                def cb(_did, _dim, dsid, _data, _found=found):
                    b = C.create_string_buffer(1024)
                    lib.H5Iget_name(dsid, b, 1024)
                    _found.append(b.value.decode().lstrip("/"))
                    return 1  # stop at the first scale

                hl.H5DSiterate_scales(did, i, None, _SCALE_CB(cb), None)
                dims.append(found[0] if found else (name if nd == 1 and hl.H5DSis_scale(did) > 0 else f"dim_{i}"))
            # contiguous, unfiltered, little-endian numeric data can be read with plain preads
            # (what netCDF4 / xarray write for fixed-size dimensions without compression)
            if isinstance(dt, np.dtype) and dt.kind in "fiu" and nd > 0 and sys.byteorder == "little":
                dcpl = lib.H5Dget_create_plist(did)
                if dcpl >= 0:
                    contiguous = lib.H5Pget_layout(dcpl) == 1 and lib.H5Pget_nfilters(dcpl) == 0
                    lib.H5Pclose(dcpl)
                    off = lib.H5Dget_offset(did)
                    if contiguous and lib.H5Tget_order(tid) == 0 and off != 0xFFFFFFFFFFFFFFFF:
                        self.raw_offset[name] = int(off)
            lib.H5Tclose(tid)
            lib.H5Sclose(sp)
            lib.H5Dclose(did)
            self.variables[name] = (shape, dt, tuple(dims))

    def is_placeholder_dimension(self, name: str) -> bool:
        a = self.attrs(name, raw=True)
        return str(a.get("NAME", "")).startswith("This is a netCDF dimension but not a netCDF variable")

    def read(self, name: str) -> np.ndarray:
        shape, dt, _ = self.variables[name]
        return self._read(name, shape, dt, None)

    def read_slab(self, name: str, start: int, stop: int, out: np.ndarray | None = None) -> np.ndarray:
        """Rows [start, stop) along the first dimension (time slab of an ERA5 variable).
        ``out``: a C-contiguous array of the slab's shape and the file's dtype to read into
        (e.g. a view of a pinned staging buffer)."""
        shape, dt, _ = self.variables[name]
        return self._read(name, (stop - start,) + shape[1:], dt, (start, stop), out)

    def read_box(self, name: str, starts, counts, out: np.ndarray | None = None) -> np.ndarray:
        """The hyperslab ``[starts[i], starts[i] + counts[i])`` of a numeric variable -- a time slab
        of ONE rank's latitude band when the rows (space points) are sharded over GPUs.  Contiguous
        native datasets are read as parallel preads of the runs the box consists of, anything
        else through H5Sselect_hyperslab."""
        shape, dt, _ = self.variables[name]
        starts, counts = [int(v) for v in starts], [int(v) for v in counts]
        if len(starts) != len(shape) or len(counts) != len(shape) or isinstance(dt, str) or dt.kind == "S":
            raise ValueError(f"read_box: {name} needs {len(shape)} starts / counts of a numeric variable")
        for a, c, full in zip(starts, counts, shape):
            if a < 0 or c < 0 or a + c > full:
                raise IndexError(f"read_box: [{a}, {a + c}) outside a dimension of {full} in {name}")
        if out is not None:
            if out.shape != tuple(counts) or out.dtype != dt or not out.flags.c_contiguous:
                raise ValueError("read_box: out must be C-contiguous with the box's shape and dtype")
        else:
            out = np.empty(counts, dtype=dt)
        if out.size == 0:
            return out
        if name in self.raw_offset and RAW_READ_THREADS > 0:
            return self._read_box_raw(name, shape, dt, starts, counts, out)
        lib = self.lib
        did = lib.H5Dopen2(self.fid, name.encode(), _H5P_DEFAULT)
        fsp = lib.H5Dget_space(did)
        msp = lib.H5Screate_simple(len(counts), _dims(*counts), None)
        try:
            lib.H5Sselect_hyperslab(fsp, _H5S_SELECT_SET, _dims(*starts), None, _dims(*counts), None)
            if lib.H5Dread(did, _h5type(lib, dt), msp, fsp, _H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError(f"H5Dread failed for {name}")
        finally:
            lib.H5Sclose(fsp)
            lib.H5Sclose(msp)
            lib.H5Dclose(did)
        return out

    def _read_box_raw(self, name, shape, dt, starts, counts, out):
        from concurrent.futures import ThreadPoolExecutor

        # the box is contiguous in the file from dimension j on (every later dimension is whole)
        j = len(shape) - 1
        while j > 0 and starts[j] == 0 and counts[j] == shape[j]:
            j -= 1
        inner = int(np.prod(shape[j + 1:], dtype=np.int64)) * dt.itemsize     # bytes per index of dim j
        run = counts[j] * inner
        strides = [int(np.prod(shape[i + 1:], dtype=np.int64)) * dt.itemsize for i in range(len(shape))]
        base = self.raw_offset[name] + sum(a * st for a, st in zip(starts, strides))
        outer = counts[:j]
        nruns = int(np.prod(outer, dtype=np.int64)) if outer else 1
        if self._fd < 0:
            self._fd = os.open(self.path, os.O_RDONLY)
        mv = memoryview(out).cast("B")

        def rd(lo, hi):
            for r in range(lo, hi):
                off, rem = base, r
                for i in range(j - 1, -1, -1):
                    rem, idx = divmod(rem, outer[i])
                    off += idx * strides[i]
                a, b = r * run, (r + 1) * run
                while a < b:
                    got = os.preadv(self._fd, [mv[a:b]], off + (a - r * run))
                    if got <= 0:
                        raise OSError(f"short read of {name} at byte {off}")
                    a += got

        per = max(1, (8 << 20) // max(run, 1))
        tasks = [(lo, min(nruns, lo + per)) for lo in range(0, nruns, per)]
        if len(tasks) == 1:
            rd(*tasks[0])
        else:
            if self._pool is None:
                self._pool = ThreadPoolExecutor(max_workers=RAW_READ_THREADS)
            list(self._pool.map(lambda t: rd(*t), tasks))
        return out

    def _read_raw(self, name, shape, dt, rng, out_buf):
        """Rows [rng) of a contiguous native dataset by parallel preads into ``out`` (the HDF5
        library reads through one thread: ~9 GB/s from the page cache; 8 preading threads
        reach the memory-copy rate)."""
        from concurrent.futures import ThreadPoolExecutor

        if out_buf is not None:
            if out_buf.shape != tuple(shape) or out_buf.dtype != dt or not out_buf.flags.c_contiguous:
                raise ValueError("read_slab: out must be C-contiguous with the slab's shape and dtype")
            out = out_buf
        else:
            out = np.empty(shape, dtype=dt)
        full = self.variables[name][0]
        row_bytes = int(np.prod(full[1:], dtype=np.int64)) * dt.itemsize
        start = self.raw_offset[name] + (rng[0] if rng is not None else 0) * row_bytes
        nbytes = out.nbytes
        if nbytes == 0:
            return out
        if self._fd < 0:
            self._fd = os.open(self.path, os.O_RDONLY)
        mv = memoryview(out).cast("B")
        piece = 8 << 20

        def rd(a):
            b = min(nbytes, a + piece)
            while a < b:
                got = os.preadv(self._fd, [mv[a:b]], start + a)
                if got <= 0:
                    raise OSError(f"short read of {name} at byte {start + a}")
                a += got

        offs = range(0, nbytes, piece)
        if len(offs) == 1:
            rd(0)
        else:
            if self._pool is None:
                self._pool = ThreadPoolExecutor(max_workers=RAW_READ_THREADS)
            list(self._pool.map(rd, offs))
        return out

    def _read(self, name, shape, dt, rng, out_buf=None):
        lib = self.lib
        if name in self.raw_offset and RAW_READ_THREADS > 0:
            return self._read_raw(name, shape, dt, rng, out_buf)
        did = lib.H5Dopen2(self.fid, name.encode(), _H5P_DEFAULT)
        fsp, msp = _H5S_ALL, _H5S_ALL
        if rng is not None:
            full = self.variables[name][0]
            fsp = lib.H5Dget_space(did)
            start = _dims(rng[0], *([0] * (len(full) - 1)))
            count = _dims(*shape)
            lib.H5Sselect_hyperslab(fsp, _H5S_SELECT_SET, start, None, count, None)
            msp = lib.H5Screate_simple(len(shape), _dims(*shape), None)
        try:
            if isinstance(dt, str):  # variable-length strings
                n = int(np.prod(shape)) if shape else 1
                bufs = (C.c_char_p * n)()
                tid = lib.H5Dget_type(did)
                if lib.H5Dread(did, tid, msp, fsp, _H5P_DEFAULT, bufs) < 0:
                    raise OSError(f"H5Dread failed for {name}")
                host = _host_lib() if n >= 4096 else None
                if host is not None:      # two C passes instead of a Python loop over n strings
                    width = max(1, int(host.dmdx_host_vlen_maxlen(bufs, n)))
                    fixed = np.empty((n, width), dtype=np.uint8)
                    ascii_only = host.dmdx_host_vlen_to_fixed(bufs, n, fixed.ctypes.data_as(C.c_void_p), width)
                    if ascii_only:
                        out = fixed.astype(np.uint32).view(f"<U{width}").reshape(shape)
                    else:  # decode the distinct values only
                        uq, inv = _factorize(fixed.view(f"S{width}").ravel())
                        out = np.array([u.decode("utf-8") for u in uq.tolist()])[inv].reshape(shape)
                else:
                    out = np.array([(b or b"").decode() for b in bufs]).reshape(shape)
                sp = lib.H5Dget_space(did)
                lib.H5Dvlen_reclaim(tid, sp, _H5P_DEFAULT, bufs)
                lib.H5Sclose(sp)
                lib.H5Tclose(tid)
                return out
            if out_buf is not None and not isinstance(dt, str) and dt.kind != "S":
                if out_buf.shape != tuple(shape) or out_buf.dtype != dt or not out_buf.flags.c_contiguous:
                    raise ValueError("read_slab: out must be C-contiguous with the slab's shape and dtype")
                out = out_buf
            else:
                out = np.empty(shape, dtype=dt)
            if dt.kind == "S":
                tid = lib.H5Dget_type(did)
                rc = lib.H5Dread(did, tid, msp, fsp, _H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
                lib.H5Tclose(tid)
                out = np.char.decode(out, "utf-8")
            else:
                rc = lib.H5Dread(did, _h5type(lib, dt), msp, fsp, _H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
            if rc < 0:
                raise OSError(f"H5Dread failed for {name}")
            return out
        finally:
            if rng is not None:
                lib.H5Sclose(fsp)
                lib.H5Sclose(msp)
            lib.H5Dclose(did)

    def attr_names(self, name: str | None = None) -> list[str]:
        """Names of ALL attributes of a dataset / of the file, including the ones whose types this
        reader does not decode (the object-reference lists of the dimension-scale API)."""
        lib = self.lib
        loc = self.fid if name is None else lib.H5Dopen2(self.fid, name.encode(), _H5P_DEFAULT)
        out, i = [], 0
        buf = C.create_string_buffer(1024)
        while True:
            aid = lib.H5Aopen_by_idx(loc, b".", _H5_INDEX_NAME, _H5_ITER_INC, i, _H5P_DEFAULT, _H5P_DEFAULT)
            if aid < 0:
                break
            i += 1
            lib.H5Aget_name(aid, 1024, buf)
            out.append(buf.value.decode())
            lib.H5Aclose(aid)
        if name is not None:
            lib.H5Dclose(loc)
        return out

    def attrs(self, name: str | None = None, raw: bool = False) -> dict:
        lib = self.lib
        loc = self.fid if name is None else lib.H5Dopen2(self.fid, name.encode(), _H5P_DEFAULT)
        out, i = {}, 0
        buf = C.create_string_buffer(1024)
        while True:
            aid = lib.H5Aopen_by_idx(loc, b".", _H5_INDEX_NAME, _H5_ITER_INC, i, _H5P_DEFAULT, _H5P_DEFAULT)
            if aid < 0:
                break
            i += 1
            lib.H5Aget_name(aid, 1024, buf)
            key = buf.value.decode()
            if not raw and key in _SKIP_ATTRS:
                lib.H5Aclose(aid)
                continue
            tid, sp = lib.H5Aget_type(aid), lib.H5Aget_space(aid)
            nd = lib.H5Sget_simple_extent_ndims(sp)
            ext = (hsize_t * max(nd, 1))()
            if nd > 0:
                lib.H5Sget_simple_extent_dims(sp, ext, None)
            n = int(np.prod([ext[k] for k in range(nd)])) if nd > 0 else 1
            dt = _np_dtype_of(lib, tid)
            val = None
            if isinstance(dt, str):
                bufs = (C.c_char_p * n)()
                lib.H5Aread(aid, tid, bufs)
                vals = [(b or b"").decode() for b in bufs]
                lib.H5Dvlen_reclaim(tid, sp, _H5P_DEFAULT, bufs)
                val = vals[0] if nd == 0 else vals
            elif dt is not None and dt.kind == "S":
                raw_b = C.create_string_buffer(dt.itemsize * n + 1)
                lib.H5Aread(aid, tid, raw_b)
                chunks = [raw_b.raw[k * dt.itemsize:(k + 1) * dt.itemsize].split(b"\x00")[0].decode("utf-8", "replace")
                          for k in range(n)]
                val = chunks[0] if n == 1 else chunks
            elif dt is not None:
                arr = np.empty(n, dtype=dt)
                lib.H5Aread(aid, _h5type(lib, dt), arr.ctypes.data_as(C.c_void_p))
                val = arr[0].item() if n == 1 else arr
            lib.H5Tclose(tid)
            lib.H5Sclose(sp)
            lib.H5Aclose(aid)
            if val is not None:
                out[key] = val
        if name is not None:
            lib.H5Dclose(loc)
        return out
