"""Scratch: K10 (one-launch Cholesky + triangular inverse) and K11 (Y = Q Mt^T) against the library
calls they replace (torch.linalg.cholesky_ex + solve_triangular = rocSOLVER potrf + rocBLAS trsm;
torch matmul = rocBLAS dgemm), ms per call (median of 20, HIP events)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
def tm(fn, reps=20):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)
g = torch.Generator(device="cuda").manual_seed(0)
for n in (20, 62, 78, 124, 220, 250, 312, 500, 1024):
    B = torch.randn((n + 50, n), generator=g, device="cuda", dtype=torch.float64)
    A = B.T @ B
    eye = torch.eye(n, dtype=torch.float64, device="cuda")
    Q = torch.randn((8760, n), generator=g, device="cuda", dtype=torch.float64)
    t_k = tm(lambda: K.chol_inv(A))
    t_kl = tm(lambda: K.chol_inv(A, want_inv=False))
    def lib():
        L, err = torch.linalg.cholesky_ex(A)
        return torch.linalg.solve_triangular(L, eye, upper=False)
    t_lib = tm(lib)
    t_pot = tm(lambda: torch.linalg.cholesky_ex(A))
    L, Linv, _ = K.chol_inv(A)
    t_nt = tm(lambda: K.gemm_nt64(Q, Linv))
    t_trsm = tm(lambda: torch.linalg.solve_triangular(L, Q.T, upper=False))
    t_mm = tm(lambda: Q @ Linv.T)
    print(f"n={n:5d}: K10 chol+inv {t_k:7.3f} ms (chol only {t_kl:7.3f}) | lib potrf {t_pot:7.3f}, potrf+trsm(I) {t_lib:7.3f} || "
          f"8760 x n: K11 {t_nt:7.3f} ms | lib trsm {t_trsm:7.3f}, dgemm {t_mm:7.3f}", flush=True)
