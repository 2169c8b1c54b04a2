"""Scratch: the standard path on a gap-free (power-law) spectrum at cfg2 size, stage split, and
the unit costs the eigen stage is built from (products with G, orthonormalisation, small eigensolves)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as S
from dmd_era5_amd.kernels import default_kernels

kern = default_kernels()
dev = torch.device("cuda")
m, n, r, _ = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
rank = int(sys.argv[2]) if len(sys.argv) > 2 else r
t0 = time.perf_counter()
blocks = bench.make_powerlaw_blocks(m, n, 1234, dev)
for B in blocks:
    kern.row_center_scale_(B, False)
torch.cuda.synchronize()
print("generated in %.1f s" % (time.perf_counter() - t0), flush=True)


def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3


for it in range(2):
    res = S.svd_snapshots(blocks, rank, kern=kern, timings=True)
    print("standard powerlaw:", json.dumps({k: (round(v, 5) if isinstance(v, float) else v) for k, v in res.info.items()}), flush=True)
print("s head", res.s[:4].tolist(), "s tail", res.s[-2:].tolist(), "expected ~", [100.0 / i * m ** 0.5 for i in (1, 2, rank)])
G = S._gram_blocks(blocks, kern, S.Comm())
for b in (77, 124, 160, 312):
    Q = torch.randn(n, b, dtype=torch.float64, device=dev)
    Qe = Q[:, : b & ~1].contiguous()
    print("b=%d: G@Q %.3f ms | K8 %.3f ms | orth %.3f ms | Q^T(GQ) %.3f ms" % (b, t(lambda: G @ Q), t(lambda: kern.symm_skinny(G, Qe, 1.0)), t(lambda: S._orth(Q)), t(lambda: Q.T @ Q)), flush=True)
for nn in (77, 96, 124, 160, 231, 250, 312, 936):
    A = torch.randn(nn, nn, dtype=torch.float64, device=dev); A = A @ A.T
    line = "n=%d: torch eigh %.2f ms | cholesky %.2f ms | svd %.2f ms" % (nn, t(lambda: torch.linalg.eigh(A)), t(lambda: torch.linalg.cholesky_ex(A)), t(lambda: torch.linalg.svd(A)))
    if nn <= kern.eigh_small_max_n:
        line += " | K7 %.2f ms" % t(lambda: kern.eigh_small(A))
    print(line, flush=True)
info = {}
torch.cuda.synchronize(); t0 = time.perf_counter()
lam, V = S.top_eigh(G, rank + max(8, rank // 4), info=info, kern=kern)
torch.cuda.synchronize(); print("top_eigh: %.1f ms" % ((time.perf_counter() - t0) * 1e3), info, flush=True)
