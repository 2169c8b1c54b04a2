"""Scratch: top_eigh on slowly decaying spectra (no gap behind the block), n = 8760."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd import svd as S
from dmd_era5_amd.kernels import default_kernels
kern = default_kernels()
n = 8760
g = torch.Generator(device="cuda").manual_seed(0)
for name, decay in (("power law i^-1 (sigma)", lambda i: 1.0 / i), ("power law i^-0.5", lambda i: i ** -0.5), ("geometric 0.98^i", lambda i: 0.98 ** i)):
    A = torch.randn((20000, n), generator=g, device="cuda", dtype=torch.float64)
    Qm, _ = torch.linalg.qr(torch.randn((n, n), generator=g, device="cuda", dtype=torch.float64))
    sig = decay(torch.arange(1, n + 1, device="cuda", dtype=torch.float64))
    G = (Qm * sig ** 2) @ Qm.T
    G = 0.5 * (G + G.T)
    del A
    for l in (62, 250):
        info = {}
        S.top_eigh(G, l, info=info, kern=kern); torch.cuda.synchronize()
        t0 = time.perf_counter(); lam, V = S.top_eigh(G, l, info=info, kern=kern); torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) * 1e3
        ref = (sig ** 2)[:l]
        print(f"{name:24s} l={l:3d}: {dt:7.1f} ms  {info}  max rel eig err {float(((lam - ref) / ref).abs().max()):.1e}", flush=True)
