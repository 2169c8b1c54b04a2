"""Scratch: first-touch costs of a fresh process on the GPU box (context, libdmdx, pinned buffers, dense libraries)."""
import time
t0 = time.perf_counter()
import torch
t1 = time.perf_counter(); print(f"import torch {t1-t0:.2f} s")
torch.cuda.init(); x = torch.zeros(1, device="cuda"); torch.cuda.synchronize()
t2 = time.perf_counter(); print(f"context + first tensor {t2-t1:.2f} s")
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
k = default_kernels()
t3 = time.perf_counter(); print(f"default_kernels {t3-t2:.2f} s")
a = torch.empty(256 << 20, dtype=torch.uint8).pin_memory(); b = torch.empty(256 << 20, dtype=torch.uint8).pin_memory()
t4 = time.perf_counter(); print(f"2 x 256 MB pinned {t4-t3:.2f} s")
y = torch.empty((8760, 129780), device="cuda"); k.row_center_scale_(y, False); torch.cuda.synchronize()
t5 = time.perf_counter(); print(f"first K5 {t5-t4:.2f} s")
g = torch.randn(512, 512, device="cuda", dtype=torch.float64); g = g @ g.T; torch.cuda.synchronize()
t6 = time.perf_counter(); print(f"first fp64 gemm {t6-t5:.2f} s")
torch.linalg.cholesky_ex(g); torch.cuda.synchronize()
t7 = time.perf_counter(); print(f"first cholesky {t7-t6:.2f} s")
torch.linalg.eigh(g); torch.cuda.synchronize()
t8 = time.perf_counter(); print(f"first eigh {t8-t7:.2f} s")
torch.linalg.solve_triangular(g, g, upper=False); torch.cuda.synchronize()
t9 = time.perf_counter(); print(f"first trsm {t9-t8:.2f} s")
k.syrk(torch.randn(512, 4096, device="cuda")); torch.cuda.synchronize()
t10 = time.perf_counter(); print(f"first K1 {t10-t9:.2f} s")
