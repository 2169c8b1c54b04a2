import time, torch
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for n in (62, 210, 512, 1024):
    A = torch.randn(n, n, dtype=torch.float64, device="cuda"); T = A @ A.T
    Tc = T.cpu()
    print(f"eigh {n}: gpu {t(lambda: torch.linalg.eigh(T)):.2f} ms, cpu(+copies) {t(lambda: [x.cuda() for x in torch.linalg.eigh(T.cpu())]):.2f} ms", flush=True)
    print(f"chol {n}: gpu {t(lambda: torch.linalg.cholesky_ex(T + n * torch.eye(n, dtype=torch.float64, device='cuda'))):.2f} ms", flush=True)
B = torch.randn(60, 8760, dtype=torch.float64, device="cuda")
print(f"svd 60x8760: gpu {t(lambda: torch.linalg.svd(B, full_matrices=False)):.2f} ms, cpu {t(lambda: [x.cuda() for x in torch.linalg.svd(B.cpu(), full_matrices=False)]):.2f} ms")
Z = torch.randn(8760, 70, dtype=torch.float64, device="cuda")
print(f"qr 8760x70: gpu {t(lambda: torch.linalg.qr(Z)):.2f} ms")
G = torch.randn(8760, 8760, dtype=torch.float64, device="cuda")
print(f"G@Z: {t(lambda: G @ Z):.2f} ms ; Z.T@Z {t(lambda: Z.T @ Z):.3f} ms; randn cpu->gpu {t(lambda: torch.randn((8760,70), dtype=torch.float64).cuda()):.2f} ms")
# stage timings inside svd.top_eigh at the cfg2 shape
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd import svd as S
A = torch.randn(8760, 300, dtype=torch.float64, device="cuda")
Gm = A @ A.T + 1e-3 * torch.eye(8760, dtype=torch.float64, device="cuda")
Q = torch.randn(8760, 77, dtype=torch.float64, device="cuda")
print(f"top_eigh(8760, l=62): {t(lambda: S.top_eigh(Gm, 62)):.2f} ms", flush=True)
print(f"  G@Q(77): {t(lambda: Gm @ Q):.2f}  _orth(77): {t(lambda: S._orth(Q)):.2f}  _orth(231): {t(lambda: S._orth(torch.cat([Q, Q.flip(0), Q.roll(5, 0)], 1))):.2f} ms", flush=True)
Q3 = torch.randn(8760, 231, dtype=torch.float64, device="cuda")
print(f"  G@S(231): {t(lambda: Gm @ Q3):.2f}  S.T@GS: {t(lambda: Q3.T @ Q3):.2f} ms", flush=True)
T = Q3.T @ Q3
print(f"  eigh(231): gpu {t(lambda: torch.linalg.eigh(T)):.2f}  host roundtrip {t(lambda: [x.cuda() for x in torch.linalg.eigh(T.cpu())]):.2f} ms", flush=True)
print(f"  chol(231) {t(lambda: torch.linalg.cholesky_ex(T)):.2f}  trsm {t(lambda: torch.linalg.solve_triangular(T, Q3.T, upper=False)):.2f} ms", flush=True)
