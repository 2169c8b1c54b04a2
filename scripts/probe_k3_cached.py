"""Scratch: is K3 at l <= 32 limited by the HBM access pattern or by the kernel itself?  The same
product on a matrix that fits the Infinity Cache (second run) vs one that does not."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
n = 8760
g = torch.Generator(device="cuda").manual_seed(1)
for l in (20, 60):
    for mb in (4096, 129780):
        nb = 1 if mb == 4096 else 8
        Xb = [torch.randn((n, mb), generator=g, device="cuda") for _ in range(nb)]
        Yb = [torch.randn((l, mb), generator=g, device="cuda") for _ in range(nb)]
        f = (lambda: K.gemm_tn_blocks(Xb, Yb)) if nb > 1 else (lambda: K.gemm_tn(Xb[0], Yb[0]))
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); e1.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"l={l} rows={nb*mb}: {ms:.3f} ms -> {nb*mb*n*4/ms/1e9:.2f} TB/s of X ({nb*mb*n*4/1e6:.0f} MB)", flush=True)
