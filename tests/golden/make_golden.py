"""Generate the golden fixtures under tests/golden/ (run in the build container).

    python tests/golden/make_golden.py

Sources of truth:
  * delay_embedding_cases.npz -- the four known answers the reference's own test
    holds (reference tests/test_02_slice_tools.py:215-231), transcribed as data.
  * mock_cfg1.npz     -- BASELINE config 1: seeded mock slice (1 var, 1 level, 25
    hourly stamps, mean-centred, delay 2 => X 5184 x 24 fp64), rank-4
    `np.linalg.svd` (the reference's "standard" call, era5_svd.py:251).
  * lowrank_*.npz     -- structured decaying-spectrum X (fp32, F-order):
    `np.linalg.svd` top-k in fp32 (reference arithmetic) and fp64 (truth), and
    `sklearn.utils.extmath.randomized_svd` (the reference's "randomized" call,
    era5_svd.py:258) with random_state=0, for the reference defaults and for the
    BASELINE config-4 setting (n_oversamples=20, n_iter=2), plus the Omega drawn.
The reference package itself is not importable here (xarray/netCDF4/pyprojroot/dvc
are not installed: ordinary ModuleNotFoundError), so SVD values are pinned on the
third-party calls it makes, not on reference fixtures (it holds none: "parity
unpinned" by reference tests).
  * conditioning_2048x160.npz -- temperature-like data with and without its time mean (the
    un-centred matrix has s_1 ~ 3e3 s_2): fp32 and fp64 `np.linalg.svd` of both, the inputs included.
Vectors are stored sign-normalised (largest |entry| of each U column positive).
"""
import os
import sys

import numpy as np
from sklearn.utils.extmath import randomized_svd

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import era5_oracle as orc  # noqa: E402


def main():
    # 1. delay embedding known answers (reference tests/test_02_slice_tools.py:215-231)
    np.savez(
        os.path.join(HERE, "delay_embedding_cases.npz"),
        x0=np.array([[0, 1, 2, 3, 4]]), d0=1, e0=np.array([[0, 1, 2, 3, 4]]),
        x1=np.array([[0, 1, 2, 3, 4]]), d1=2, e1=np.array([[0, 1, 2, 3], [1, 2, 3, 4]]),
        x2=np.array([[0, 1, 2, 3, 4]]), d2=3,
        e2=np.array([[0, 1, 2], [1, 2, 3], [2, 3, 4]]),
        x3=np.array([[0, 1, 2], [3, 4, 5]]), d3=2,
        e3=np.array([[0, 1], [3, 4], [1, 2], [4, 5]]),
    )

    # 2. BASELINE config 1
    seed = 20190101
    variables, _, _ = orc.mock_era5(25, ["temperature"], [1000], seed)
    X, X_mean, _ = orc.preprocess(variables, True, False, 2)
    assert X.shape == (5184, 24)
    U, s, V = orc.svd_standard(X, 4)
    U, V = orc.svd_flip(U, V)
    np.savez_compressed(
        os.path.join(HERE, "mock_cfg1.npz"),
        seed=seed, s=s, U=U, V=V, X_mean=X_mean,
        X_checksum=np.array([X.sum(), np.abs(X).sum(), (X * X).sum()]),
    )

    # 3. structured matrices
    for name, (m, n, rank, k) in {
        "lowrank_4096x192": (4096, 192, 100, 50),
        "lowrank_wide_160x1024": (160, 1024, 60, 20),
    }.items():
        X = orc.lowrank_matrix(m, n, rank, seed=0)
        U32, s32, V32 = orc.svd_standard(X, k)
        U32, V32 = orc.svd_flip(U32, V32)
        U64, s64, V64 = orc.svd_standard(X.astype(np.float64), k)
        U64, V64 = orc.svd_flip(U64, V64)
        out = dict(m=m, n=n, rank=rank, k=k, seed=0,
                   s32=s32, U32=U32, V32=V32, s64=s64,
                   U64=U64.astype(np.float32), V64=V64.astype(np.float32))
        for tag, kw in {"rdef": {}, "rcfg4": dict(n_oversamples=20, n_iter=2)}.items():
            Ur, sr, Vr = randomized_svd(X, k, random_state=0, **kw)
            p = kw.get("n_oversamples", 10)
            nmin = min(m, n)
            omega = np.random.RandomState(0).normal(size=(nmin, k + p))
            # the restatement must reproduce sklearn from the same Omega
            Uo, so, Vo = orc.svd_randomized(X, k, omega=omega, **kw)
            assert np.allclose(so, sr, rtol=1e-5), (so, sr)
            out.update({f"{tag}_s": sr, f"{tag}_U": Ur, f"{tag}_V": Vr,
                        f"{tag}_omega": omega.astype(np.float32)})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    # 4. conditioning pair: the same anomalies with and without a large
    # time-mean (temperature-like: 280 + O(1..10) anomalies).  Un-centred, s_1 is ~3e3 x s_2, which
    # is what stresses a Gram-based SVD; centred, it is the well-conditioned problem.
    m, n, k = 2048, 160, 12
    anom = orc.lowrank_matrix(m, n, 40, seed=5) * 0.05
    mean = 280.0 + 5.0 * np.random.RandomState(6).standard_normal((m, 1)).astype(np.float32)
    raw = (anom + mean).astype(np.float32)
    cen = (raw - raw.mean(axis=1, keepdims=True)).astype(np.float32)
    out = dict(m=m, n=n, k=k)
    for tag, X in (("raw", raw), ("cen", cen)):
        U32, s32, V32 = orc.svd_standard(X, k)
        U64, s64, V64 = orc.svd_standard(X.astype(np.float64), k)
        U64, V64 = orc.svd_flip(U64, V64)
        out.update({f"{tag}_X": X, f"{tag}_s32": s32, f"{tag}_s64": s64,
                    f"{tag}_U64": U64.astype(np.float32), f"{tag}_V64": V64.astype(np.float32)})
    np.savez_compressed(os.path.join(HERE, "conditioning_2048x160.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
