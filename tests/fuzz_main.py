"""Not collected by pytest (run by hand on the GPU box: python tests/fuzz_main.py [cases]).  Random
[era5-svd] configurations through main() against the oracle pipeline
(oracle.preprocess + numpy fp64 SVD) on seeded mock slices."""
import os, sys, shutil, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rs0 = np.random.RandomState(11 + int(os.environ.get("DMDX_FUZZ_SEED", "0")))
ALLV = ["temperature", "u_component_of_wind", "v_component_of_wind"]
ALLL = [1000, 925, 850, 500]
bad = 0
for i in range(N):
    root = tempfile.mkdtemp(prefix="dmdx_fuzz_")
    os.environ["DMD_ERA5_ROOT"] = root
    os.environ["DMDX_NETCDF_BACKEND"] = ["hdf5", "scipy"][i % 2] if i % 5 else "hdf5"
    from dmd_era5_amd import io_netcdf, era5_svd
    from dmd_era5_amd.config_parser import config_parser
    from dmd_era5_amd.create_mock_data import add_download_attributes, create_mock_era5
    from oracle import era5_oracle as orc
    nv = rs0.randint(1, 4); vs = list(rs0.choice(ALLV, nv, replace=False))
    file_levels = sorted(rs0.choice(ALLL, rs0.randint(1, 4), replace=False).tolist(), reverse=True)
    want_levels = list(rs0.permutation(file_levels)[: rs0.randint(1, len(file_levels) + 1)])
    step = int(rs0.choice([1, 1, 3, 6])); d = int(rs0.choice([1, 2, 3])); center = bool(rs0.rand() < 0.7)
    scale = bool(center and rs0.rand() < 0.4); typ = "standard" if rs0.rand() < 0.7 else "randomized"
    days = int(rs0.randint(2, 5)); k = int(rs0.randint(2, 7)); dtype = np.float32 if rs0.rand() < 0.7 else np.float64
    end = f"2019-01-0{1 + days}T00"
    if rs0.rand() < 0.15:   # a WIDE problem: one field on the 5-degree grid over ~4 months of hourly data
        vs, file_levels, want_levels, step, d, dtype = vs[:1], file_levels[:1], file_levels[:1], 1, 1, np.float32
        end = f"2019-0{int(rs0.randint(4, 6))}-{int(rs0.randint(10, 28))}T00"
    cfg = {"source_path": "synthetic", "variables": ",".join(vs), "levels": ",".join(map(str, want_levels)),
           "svd_type": typ, "delay_embedding": d, "mean_center": center, "scale": scale,
           "start_datetime": "2019-01-01T00", "end_datetime": end, "delta_time": f"{step}h",
           "n_components": k, "save_data_matrix": True, "svd_seed": 0}
    wcfg = dict(cfg, delta_time="1h", levels=",".join(map(str, file_levels)))
    try:
        p = config_parser(cfg, "era5-svd"); pw = config_parser(wcfg, "era5-svd")
        full = add_download_attributes(create_mock_era5(cfg["start_datetime"], cfg["end_datetime"], pw["variables"],
                                                        pw["levels"], seed=100 + i, dtype=dtype), pw)
        io_netcdf.to_netcdf(full, p["era5_slice_path"])
        res, _, _ = era5_svd.main(cfg, write_to_netcdf=True)
        lidx = [file_levels.index(L) for L in want_levels]
        variables = {v: full[v].values[::step][:, lidx] for v in p["variables"]}
        X, X_mean, X_std = orc.preprocess(variables, center, scale, d)
        X64 = X.astype(np.float64)
        sref = np.linalg.svd(X64, compute_uv=False)
        s = res["s"].values.astype(np.float64)
        U = res["U"].values.astype(np.float64); V = res["V"].values.astype(np.float64)
        okX = np.allclose(res["X"].values, X, rtol=0, atol=1e-3 * max(1.0, np.abs(X).max() / 30))
        tol = 5e-5 if typ == "standard" else 0.06
        oks = np.abs(s - sref[: len(s)]).max() <= tol * sref[0]
        rec = np.linalg.norm(X64 - (U * s) @ V); opt = np.sqrt((sref[len(s):] ** 2).sum())
        okr = typ != "standard" or rec <= 1.001 * opt + 2e-4 * np.linalg.norm(X64)
        if not (okX and oks and okr and U.shape == (X.shape[0], len(s))):
            bad += 1
            print("BAD", i, cfg, "okX", okX, "oks", oks, "okr", okr, np.abs(s - sref[:len(s)]).max() / sref[0], rec, opt, flush=True)
    except Exception as e:
        bad += 1
        print("EXC", i, cfg, repr(e)[:300], flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)
print("done", N, "configs,", bad, "flagged")
