"""Scratch: pure-noise matrix (flat Marchenko-Pastur spectrum): where does the eigen stage spend its time?"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
from dmd_era5_amd import svd as dsvd
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
for (m, n, k) in ((259560, 8760, 50), (100000, 4000, 50), (50000, 2000, 20)):
    X = torch.randn((n, m), generator=g, device="cuda", dtype=torch.float32)
    G = K.syrk(X)
    for meth in ("auto", "full"):
        info = {}
        torch.cuda.synchronize(); t0 = time.perf_counter()
        lam, V = dsvd.top_eigh(G, k + 12, method=meth, info=info, kern=K)
        torch.cuda.synchronize()
        print(f"m={m} n={n} k={k} top_eigh[{meth}]: {(time.perf_counter()-t0)*1e3:.0f} ms {info}", flush=True)
    del X, G
