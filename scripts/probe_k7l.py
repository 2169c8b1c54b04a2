"""Scratch: K7L (one-sided Jacobi, one launch) against the library calls it replaces."""
import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
from dmd_era5_amd import svd as S
K = default_kernels()
def t(fn, reps=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    t.last = ts
    return sorted(ts)[len(ts) // 2]
for n in (78, 124, 160, 250, 312, 500, 936):
    rs = np.random.RandomState(n)
    # (a) a Rayleigh-Ritz-like matrix: nearly diagonal, graded; (b) a dense graded matrix
    Qm, _ = np.linalg.qr(rs.standard_normal((n, n)))
    lam = 10.0 ** np.linspace(6, -3, n)
    E = rs.standard_normal((n, n)) * 1e-3
    near = np.diag(lam) + np.sqrt(np.outer(lam, lam)) * 0.5 * (E + E.T)
    dense = (Qm * lam) @ Qm.T; dense = 0.5 * (dense + dense.T)
    for tag, A in (("near-diagonal", near), ("dense", dense)):
        Ad = torch.from_numpy(A).cuda()
        L = torch.linalg.cholesky(Ad)
        Ct = L.T.contiguous()
        tk = t(lambda: K.svd_jacobi(Ct.clone()))
        sw = K.last_jacobi_sweeps
        reps_k = ["%.2f" % v for v in t.last]
        te = t(lambda: torch.linalg.eigh(Ad))
        tall = t(lambda: S._eigh_desc(Ad, K))
        ts = t(lambda: torch.linalg.svd(Ct), reps=2) if n <= 312 else float("nan")
        spread = max(t.last) if False else 0
        print(f"n={n:4d} {tag:14s}: K7L {tk:6.2f} ms ({sw} sweeps) | chol+K7L (_eigh_desc) {tall:6.2f} ms | torch eigh {te:6.2f} ms | torch svd {ts:6.2f} ms | K7L reps {reps_k}", flush=True)
