"""Scratch: the mean-deflated route at cfg2's full size: X = 280 + per-row offset + (rank-64 + noise), NOT
centred.  Size-independent checks in fp64 torch products: U^T U = I, V V^T = I, X^T u_j = s_j v_j."""
import sys, os, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
m, n, r, _ = bench.WORKLOADS["cfg2"]
blocks = bench.make_snapshot_blocks(m, n, 1234, torch.device("cuda"))
g = torch.Generator(device="cuda").manual_seed(7)
for B in blocks:
    B.mul_(1e-4)                                               # anomalies of O(10)
    B.add_(280.0 + 5.0 * torch.randn((1, B.shape[1]), generator=g, device="cuda"))
torch.cuda.synchronize(); t0 = time.perf_counter()
res = dsvd.svd_snapshots(blocks, r, kern=K)
torch.cuda.synchronize(); print(f"svd_snapshots (un-centred): {(time.perf_counter()-t0)*1e3:.0f} ms, info { {k: v for k, v in res.info.items() if not k.startswith('t_')} }")
s, Vh, Ut = res.s, res.Vh, res.Ut
eye = torch.eye(r, dtype=torch.float64, device="cuda")
UtU = torch.zeros((r, r), dtype=torch.float64, device="cuda"); XtU = torch.zeros((n, r), dtype=torch.float64, device="cuda")
r0 = 0
for B in blocks:
    Ub = Ut[:, r0:r0 + B.shape[1]].double(); UtU += Ub @ Ub.T
    for j0 in range(0, n, 2190): XtU[j0:j0 + 2190] += B[j0:j0 + 2190].double() @ Ub.T
    r0 += B.shape[1]
err = (XtU - (Vh.T * s)).norm(dim=0)
print("s head", s[:4].tolist(), "s_2/s_1 %.2e s_50/s_1 %.2e" % (float(s[1] / s[0]), float(s[-1] / s[0])))
print("V V^T - I %.1e   U^T U - I %.1e   max |X^T u - s v| / s_1 %.1e   / s_2 %.1e   max_j / s_j %.1e" % (
    float((Vh @ Vh.T - eye).abs().max()), float((UtU - eye).abs().max()), float((err / s[0]).max()), float((err[1:] / s[1]).max()), float((err / s).max())))
