"""Scratch: ranks far beyond the usual (k = 300 .. 900): K2 column groups, K3 tile rows, full eigensolver."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.engine import svd_numpy
rs = np.random.RandomState(5)
for (m, n, k, typ) in ((50000, 2000, 600, "standard"), (50000, 2000, 600, "randomized"), (30000, 1000, 900, "standard"), (200000, 3000, 300, "randomized")):
    r = n // 2
    X = (rs.standard_normal((m, r)) @ (rs.standard_normal((r, n)) * (0.995 ** np.arange(r))[:, None])).astype(np.float32)
    U, s, V = svd_numpy(X, typ, k, device="cuda:0", **({"random_state": 0} if typ == "randomized" else {}))
    X64 = X.astype(np.float64)
    sref = np.linalg.svd(X64, compute_uv=False)
    kk = len(s)
    live = s > 1e-6 * s[0]
    Ul = U[:, live].astype(np.float64)
    print(f"m={m} n={n} k={k} {typ}: max |ds|/s1 {np.abs(s - sref[:kk]).max() / sref[0]:.1e}; U orth (live {live.sum()}) {np.abs(Ul.T @ Ul - np.eye(live.sum())).max():.1e}; "
          f"recon / optimal {np.linalg.norm(X64 - (U.astype(np.float64) * s) @ V.astype(np.float64)) / max(np.sqrt((sref[kk:] ** 2).sum()), 1e-300):.4f}", flush=True)
