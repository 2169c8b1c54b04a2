"""Scratch: the eigen stage (svd.top_eigh, n = 8760, l = 62 as at cfg2 rank 50) over a range of
spectra -- time, method taken, accuracy of the leading 50 pairs against the library's full solver."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd import svd as S
from dmd_era5_amd.kernels import default_kernels
kern = default_kernels(); dev = torch.device("cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8760
k = int(sys.argv[2]) if len(sys.argv) > 2 else 50
l = k + max(8, k // 4)
g = torch.Generator(device=dev); g.manual_seed(5)
Q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device=dev, generator=g))
i = torch.arange(1, n + 1, dtype=torch.float64, device=dev)
spectra = {
    "sigma i^-0.25": i ** -0.5, "sigma i^-0.5": i ** -1.0, "sigma i^-1": i ** -2.0, "sigma i^-2": i ** -4.0,
    "sigma 0.99^i": 0.99 ** (2 * i), "sigma 0.999^i": 0.999 ** (2 * i),
    "flat 100 then 1e-3": torch.where(i <= 100, torch.ones_like(i), 1e-6 * torch.ones_like(i)),
    "flat (identity + 1e-3 i/n)": 1.0 + 1e-3 * (1 - i / n),
    "two clusters of 40 (rel. split 1e-6)": torch.where(i <= 40, 1.0 + 1e-6 * i, torch.where(i <= 80, 0.5 + 1e-6 * i, 1e-4 / i)),
}
X = torch.randn(4 * n, n, dtype=torch.float64, device=dev, generator=g)
wish = X.T @ X; del X
for name, lam in list(spectra.items()) + [("Wishart m = 4n (white noise)", None)]:
    if lam is None:
        G = wish
    else:
        lam = torch.sort(lam, descending=True).values
        G = (Q * lam) @ Q.T; G = 0.5 * (G + G.T)
    torch.cuda.synchronize(); t0 = time.perf_counter(); ref = torch.linalg.eigvalsh(G).flip(0); torch.cuda.synchronize(); tfull = time.perf_counter() - t0
    for rep in range(2):
        info = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        w, V = S.top_eigh(G, l, info=info, kern=kern)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    err = float(((w[:k] - ref[:k]).abs() / ref[0]).max())
    res = float((G @ V[:, :k] - V[:, :k] * w[:k]).norm(dim=0).max() / ref[0])
    print(f"{name:40s} top_eigh {dt*1e3:7.1f} ms ({info.get('eig_method')}, {info.get('eig_products')} products, degrees {info.get('eig_degrees')}) "
          f"| full eigvalsh {tfull*1e3:6.0f} ms | max |dlam|/lam1 {err:.1e}, residual/lam1 {res:.1e}", flush=True)
