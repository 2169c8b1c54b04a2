"""Scratch: un-centred (280 K + anomalies) data at ranks beyond the Jacobi kernel's size (l > 96):
the mean-deflated Rayleigh-Ritz matrix is graded by alpha -- does the library eigh keep up?"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.engine import svd_numpy
rs = np.random.RandomState(3)
for (m, n, k, decay) in ((20000, 600, 100, 0.93), (50000, 900, 150, 0.95), (20000, 600, 100, 0.8)):
    r = n // 2
    X = 280.0 + 5 * rs.standard_normal((m, 1)) + rs.standard_normal((m, r)) @ (rs.standard_normal((r, n)) * (10 * decay ** np.arange(r))[:, None])
    X = X.astype(np.float32)
    U, s, V = svd_numpy(X, "standard", k, device="cuda:0")
    X64 = X.astype(np.float64)
    sref = np.linalg.svd(X64, compute_uv=False)
    ds = np.abs(s - sref[:k]) / sref[0]
    orth = np.abs(U.astype(np.float64).T @ U.astype(np.float64) - np.eye(k)).max()
    rec = np.linalg.norm(X64 - (U.astype(np.float64) * s) @ V.astype(np.float64)); opt = np.sqrt((sref[k:] ** 2).sum())
    print(f"m={m} n={n} k={k} decay={decay}: max |ds|/s1 {ds.max():.1e}, rel err of s_k {abs(s[-1]-sref[k-1])/sref[k-1]:.1e}, "
          f"s_k/s_1 {sref[k-1]/sref[0]:.1e}, U orth {orth:.1e}, recon {rec/opt:.6f} x optimal", flush=True)
