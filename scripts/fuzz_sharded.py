"""Scratch (CPU, gloo, kernel double): random slice / config / world size -- the latitude-band sharded
device pipeline against the single-rank one.  python scripts/fuzz_sharded.py [cases]"""
import os, sys, socket, tempfile
from datetime import timedelta
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.multiprocessing as mp


def run(path, cfg, comm, stream=True):
    from kernel_double import CpuKernelDouble
    from dmd_era5_amd import era5_svd, io_netcdf
    io_netcdf.LAZY_BYTES = 1000 if cfg["_lazy"] else 1 << 40
    os.environ["DMDX_NETCDF_BACKEND"] = "hdf5"
    era5_svd.SLAB_BYTES = cfg["_slab"]
    ds = io_netcdf.open_dataset(path)
    ds = ds[cfg["_variables"]]
    if stream and cfg["_stream"]:
        os.environ["DMDX_STREAM_BYTES"] = str(cfg["_stream"])
    try:
        return era5_svd._device_pipeline(ds, cfg, comm, kern=CpuKernelDouble(), device=torch.device("cpu"))
    finally:
        os.environ.pop("DMDX_STREAM_BYTES", None)


def worker(rank, world, port, path, cfg, q):
    for p in (ROOT, os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import logging
    logging.disable(logging.CRITICAL)
    sys.stdout = open(os.devnull, "w")
    import torch.distributed as dist
    from dmd_era5_amd import svd as dsvd
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        U, s, V, coords, X, Xm, Xs = run(path, cfg, dsvd.TorchDistComm())
        q.put((rank, None) if rank else (0, (U, s, V, None if X is None else X.values, None if Xm is None else Xm.values, None if Xs is None else Xs.values)))
    except Exception as e:
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    import logging
    from dmd_era5_amd import io_netcdf, svd as dsvd
    from dmd_era5_amd.create_mock_data import create_mock_era5
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    rs = np.random.RandomState(2024 + int(os.environ.get("DMDX_FUZZ_SEED", "0")))
    tmp = tempfile.mkdtemp(prefix="dmdx_fuzz_sharded_")
    os.environ["DMDX_NETCDF_BACKEND"] = "hdf5"
    bad = 0
    for i in range(N):
        nvar = int(rs.randint(1, 4)); names = ["temperature", "u_component_of_wind", "specific_humidity"][:nvar]
        levels_all = [1000, 850, 500, 300][: int(rs.randint(1, 5))]
        hours = int(rs.choice([30, 49, 97]))
        ds = create_mock_era5("2019-01-01T00", (np.datetime64("2019-01-01T00") + np.timedelta64(hours - 1, "h")).astype(str),
                              names, levels_all, seed=100 + i, dtype=np.float32 if i % 3 else np.float64)
        t = np.arange(hours, dtype=np.float64)[:, None, None, None]
        lat = np.radians(ds.coords["latitude"].values)[None, None, :, None]; lon = np.radians(ds.coords["longitude"].values)[None, None, None, :]
        for v, name in enumerate(names):
            f = ds[name].values.astype(np.float64)
            f = f + 60 * np.sin(2 * np.pi * t / 24) * np.cos(lat) * np.cos(lon + v) + 35 * np.cos(2 * np.pi * t / 11) * np.sin(2 * lat) * np.sin(2 * lon)
            f = f + 20 * (t / hours) ** 2 * np.cos(3 * lon) * np.ones_like(lat)
            ds[name].values = f.astype(ds[name].values.dtype)
        path = os.path.join(tmp, f"slice{i}.nc")
        io_netcdf.to_netcdf(ds, path)
        sel = sorted(rs.choice(len(levels_all), size=int(rs.randint(1, len(levels_all) + 1)), replace=False).tolist())
        if rs.rand() < 0.5: sel = sel[::-1]
        use = sorted(rs.choice(nvar, size=int(rs.randint(1, nvar + 1)), replace=False).tolist())
        d = int(rs.randint(1, 4)); center = bool(rs.rand() < 0.7); scale = center and bool(rs.rand() < 0.5)
        cfg = {"delay_embedding": d, "mean_center": center, "scale": scale, "levels": [levels_all[j] for j in sel],
               "delta_time": timedelta(hours=int(rs.choice([1, 1, 3, 6]))), "n_components": 3,
               "svd_type": "standard" if rs.rand() < 0.6 else "randomized", "save_data_matrix": True, "svd_seed": 0,
               "_lazy": bool(rs.rand() < 0.7), "_slab": int(rs.choice([1 << 16, 1 << 20, 1 << 28])), "_variables": [names[j] for j in use]}
        # half of the eligible cases go through the two-pass streaming path (pieces of a few latitude rows)
        cfg["_stream"] = 0
        if (cfg["svd_type"] == "randomized" or center) and rs.rand() < 0.5:
            cfg["save_data_matrix"] = False
            cfg["_stream"] = int(rs.randint(2, 15)) * 4 * hours * len(sel) * 72
        world = int(rs.choice([1, 2, 3, 5])) if cfg["_stream"] else int(rs.choice([2, 3, 5]))
        ctx = mp.get_context("spawn"); q = ctx.Queue()
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        procs = [ctx.Process(target=worker, args=(r, world, port, path, cfg, q)) for r in range(world)]
        for p in procs: p.start()
        got = dict(q.get(timeout=300) for _ in range(world))
        for p in procs: p.join(timeout=60)
        logging.disable(logging.CRITICAL)
        so = sys.stdout; sys.stdout = open(os.devnull, "w")
        try:
            U1, s1, V1, c1, X1, Xm1, Xs1 = run(path, cfg, dsvd.Comm(), stream=False)
        finally:
            sys.stdout = so
        desc = {k: v for k, v in cfg.items() if not k.startswith("_")} | {"world": world, "vars": cfg["_variables"], "lazy": cfg["_lazy"], "hours": hours}
        if any(isinstance(v, str) for v in got.values()):
            bad += 1; print("EXC", i, desc, got); continue
        U, s, V, X, Xm, Xs = got[0]
        ok = ((X is None and X1 is None) or np.array_equal(X, X1.values)) and U.shape == U1.shape and np.allclose(s, s1, rtol=2e-5)
        ok = ok and (Xm is None) == (Xm1 is None) and (Xm is None or np.allclose(Xm, Xm1.values, atol=1e-4 * np.abs(Xm1.values).max()))
        ok = ok and (Xs is None) == (Xs1 is None) and (Xs is None or np.allclose(Xs, Xs1.values, rtol=1e-5))
        rec, rec1 = (U.astype(np.float64) * s) @ V, (U1.astype(np.float64) * s1) @ V1
        ok = ok and np.linalg.norm(rec - rec1) <= 1e-3 * np.linalg.norm(rec1)
        if not ok:
            bad += 1; print("BAD", i, desc, s, s1)
        else:
            print("ok", i, desc["world"], desc["svd_type"], "streamed" if cfg["_stream"] else "resident", "d", d, "levels", desc["levels"], "vars", len(use), "dt", desc["delta_time"], flush=True)
    print("done", N, "cases,", bad, "flagged")
