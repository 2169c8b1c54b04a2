"""Scratch: BASELINE config 5 -- optimized DMD on rank-200 reduced coordinates (n = 8760)."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd import bopdmd as bop
rs = np.random.RandomState(0)
r, n = 200, 8760
t = np.arange(n) / 24.0
freq = np.sort(rs.uniform(0.02, 6.0, r // 2))
alpha = -rs.uniform(1e-4, 3e-3, r // 2) + 1j * 2 * np.pi * freq
alpha = np.concatenate([alpha, alpha.conj()])
modes = rs.standard_normal((r, r)) + 1j * rs.standard_normal((r, r))
H = np.exp(np.outer(t, alpha)) @ modes + 1e-6 * rs.standard_normal((n, r))
for dev in ("cuda",):
    for dt in (torch.complex128, torch.complex64):
        Hd = torch.from_numpy(H).to(dev).to(dt)
        td = torch.from_numpy(t).to(dev)
        a0 = torch.from_numpy(alpha * (1 + 1e-4 * rs.standard_normal(r))).to(dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = bop.optdmd(Hd, td, r, alpha0=a0, tol=1e-9 if dt == torch.complex128 else 1e-4, maxiter=20)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        err = max(np.min(np.abs(res.eigs.cpu().numpy() - a)) for a in alpha)
        print(f"cfg5 {dt}: {res.n_iter} LM iterations in {el*1e3:.1f} ms ({el/max(res.n_iter,1)*1e3:.1f} ms/iter), rel residual {res.rel_error:.2e}, max |d alpha| {err:.2e}", flush=True)
