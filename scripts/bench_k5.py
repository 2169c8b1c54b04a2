"""Scratch: K5 (row mean / std, centre, scale in place) alone: TB/s of traffic (2 reads + 1 write, 3 + 1 with scale)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
for (m, n) in ((129780, 8760), (130872, 3653), (129779, 8760)):
    X = torch.randn((n, m), generator=g, device="cuda", dtype=torch.float32) * 10 + 280
    ref = X[:, :4096].double()
    for scale in (False, True):
        Y = X.clone(); torch.cuda.synchronize(); t0 = time.perf_counter()
        mu, sd = K.row_center_scale_(Y, scale); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        c = ref - ref.mean(dim=0)
        emu = float((mu[:4096].double() - ref.mean(dim=0)).abs().max())
        if scale: c = c / c.std(dim=0, unbiased=False)
        err = float((Y[:, :4096].double() - c).abs().max())
        passes = 4 if scale else 3
        print(f"K5 m={m} n={n} scale={scale}: {dt*1e3:.2f} ms = {passes*m*n*4/dt/1e12:.2f} TB/s ({passes} passes); max err mean {emu:.1e} values {err:.1e}", flush=True)
