"""Scratch: the l x l Gram of a tall m x l block (CholeskyQR of the randomized path) through K1."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
nb, mb = 119, 130872
for l in (60, 70, 128, 220, 250):
    Yb = [torch.randn((l, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(nb)]
    K.syrk_blocks(Yb); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): G = K.syrk_blocks(Yb)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    gb = nb * mb * l * 4 / 1e9
    # torch reference: one big matmul per block in fp32 (rocBLAS)
    t0 = time.perf_counter()
    for _ in range(3):
        acc = torch.zeros((l, l), device="cuda", dtype=torch.float64)
        for Y in Yb: acc += (Y @ Y.T).double()
    torch.cuda.synchronize(); dt2 = (time.perf_counter() - t0) / 3
    print(f"l={l}: syrk_blocks {dt*1e3:.1f} ms = {gb/dt/1e3:.2f} TB/s of Y ({gb:.1f} GB); rocBLAS per-block sgemm {dt2*1e3:.1f} ms; "
          f"rel diff {float((G-acc).abs().max()/acc.abs().max()):.1e}", flush=True)
    del Yb
