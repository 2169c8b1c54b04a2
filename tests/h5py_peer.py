"""Peer of tests/test_interop_h5py.py, run under an interpreter that has h5py (this image:
/opt/conda/bin/python3.9, h5py 3.3 on HDF5 1.10.6) -- an HDF5 implementation of the dimension-scale
conventions (H5DS, the layer netCDF-C builds NETCDF4 files on) that is NOT this repository's writer
or reader.  Plain h5py / numpy / json only; nothing of the package is imported here.

  h5py_peer.py dump  FILE            -> JSON on stdout: every dataset's shape, dtype, attributes,
                                        is_scale, the scales attached to each of its dimensions
                                        (through H5DS, i.e. DIMENSION_LIST / REFERENCE_LIST as the
                                        library resolves them), a checksum and the edge values
  h5py_peer.py write FILE            -> an ERA5-slice-shaped file the way h5py makes dimension
                                        scales (make_scale / attach_scale), values from a fixed seed
"""
import hashlib
import json
import sys

import h5py
import numpy as np


def _plain(v):
    if isinstance(v, bytes):
        return v.decode("utf-8", "replace")
    if isinstance(v, np.ndarray):
        if v.dtype.kind in "SO":
            return [_plain(x) for x in v.tolist()]
        if v.dtype.kind == "V" or v.dtype.names:
            return f"<compound x{v.size}>"
        return v.tolist()
    if isinstance(v, np.generic):
        return v.item()
    if isinstance(v, h5py.Reference):
        return "<ref>"
    return v


def dump(path):
    out = {"root_attrs": {}, "datasets": {}}
    with h5py.File(path, "r") as f:
        for k in f.attrs:
            out["root_attrs"][k] = _plain(f.attrs[k])

        def visit(name, obj):
            if not isinstance(obj, h5py.Dataset):
                return
            d = {"shape": list(obj.shape), "dtype": str(obj.dtype), "kind": obj.dtype.kind,
                 "is_scale": bool(h5py.h5ds.is_scale(obj.id)), "attrs": {}, "scales": []}
            for k in obj.attrs:
                if k in ("DIMENSION_LIST", "REFERENCE_LIST"):
                    d["attrs"][k] = "<present>"
                else:
                    d["attrs"][k] = _plain(obj.attrs[k])
            if not d["is_scale"]:
                for i in range(obj.ndim):
                    d["scales"].append([s.name.lstrip("/") for s in obj.dims[i].values()])
            a = obj[()]
            if a.dtype.kind in "OS":
                vals = [_plain(x) for x in np.asarray(a).reshape(-1).tolist()]
                d["sha"] = hashlib.sha256("\x00".join(vals).encode()).hexdigest()
                d["edge"] = [vals[0], vals[-1]] if vals else []
            else:
                a = np.ascontiguousarray(a)
                d["sha"] = hashlib.sha256(a.tobytes()).hexdigest()
                flat = a.reshape(-1)
                d["edge"] = [flat[0].item(), flat[-1].item()] if flat.size else []
            out["datasets"][name] = d

        f.visititems(visit)
    json.dump(out, sys.stdout)


def write(path):
    rs = np.random.RandomState(11)
    nt, nlev, nlat, nlon = 30, 2, 5, 8
    with h5py.File(path, "w") as f:
        f.attrs["source_path"] = np.bytes_("peer")
        f.attrs["levels"] = np.array([1000, 850], dtype=np.int64)
        hours = (np.datetime64("2019-01-01T00", "h") - np.datetime64("1900-01-01T00", "h")).astype(np.int64) + np.arange(nt)
        t = f.create_dataset("time", data=hours.astype(np.int64))
        t.attrs["units"] = np.bytes_("hours since 1900-01-01 00:00:00")
        t.attrs["calendar"] = np.bytes_("proleptic_gregorian")
        lev = f.create_dataset("level", data=np.array([1000, 850], dtype=np.int64))
        lat = f.create_dataset("latitude", data=np.linspace(90, -90, nlat))
        lon = f.create_dataset("longitude", data=np.linspace(0, 315, nlon))
        for name, ds in (("time", t), ("level", lev), ("latitude", lat), ("longitude", lon)):
            ds.make_scale(name)
        for v in ("temperature", "u_component_of_wind"):
            # the second variable chunked + shuffled + deflated (what `encoding={"zlib": True}` gives):
            # the reader's library path instead of its raw preadv path
            kw = {} if v == "temperature" else {"chunks": (7, 1, 5, 8), "compression": "gzip", "shuffle": True}
            d = f.create_dataset(v, data=rs.standard_normal((nt, nlev, nlat, nlon)).astype(np.float32), **kw)
            d.attrs["units"] = np.bytes_("K" if v == "temperature" else "m s**-1")
            for i, s in enumerate((t, lev, lat, lon)):
                d.dims[i].attach_scale(s)


if __name__ == "__main__":
    {"dump": dump, "write": write}[sys.argv[1]](sys.argv[2])
