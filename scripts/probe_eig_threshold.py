"""Scratch: where should top_eigh switch from the full solver to the block power / Krylov path?"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
for n in (384, 512, 768, 1024, 1536, 2048):
    for kind in ("lowrank+noise", "slow decay"):
        m = 60000
        if kind == "lowrank+noise":
            A = torch.randn((m, 64), generator=g, device="cuda"); B = torch.randn((n, 64), generator=g, device="cuda")
            X = (A * (100 * 0.9 ** torch.arange(64, device="cuda"))) @ B.T + 0.01 * torch.randn((m, n), generator=g, device="cuda")
        else:
            X = torch.randn((m, n), generator=g, device="cuda") * (0.99 ** torch.arange(n, device="cuda"))
        G = K.syrk(X.T.contiguous())
        out = []
        for meth in ("full", "krylov"):
            if 3 * (62 + 15) > n and meth == "krylov": out.append("krylov n/a"); continue
            info = {}
            dsvd.top_eigh(G, 62, method=meth, info=info, kern=K); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): dsvd.top_eigh(G, 62, method=meth, info=info, kern=K)
            torch.cuda.synchronize(); out.append(f"{meth} {(time.perf_counter()-t0)/3*1e3:.1f} ms ({info['eig_method']})")
        print(f"n={n} {kind}: " + "; ".join(out), flush=True)
