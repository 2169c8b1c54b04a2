"""Scratch: delay embedding d = 2, 3 at cfg2's full size (zero-copy embedded views, K6 shifted Gram sum):
E^T u_j = s_j v_j and U^T U = I in fp64 torch products on the materialised embedding, one block at a time."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
m, n, r, _ = bench.WORKLOADS["cfg2"]
blocks = bench.make_snapshot_blocks(m, n, 1234, torch.device("cuda"))
for B in blocks: K.row_center_scale_(B, False)
for d in (2, 3):
    for typ in ("standard", "randomized"):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = dsvd.svd_snapshots(blocks, r, delay=d, kern=K) if typ == "standard" else dsvd.svd_randomized(blocks, r, delay=d, random_state=0, kern=K)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        s, Vh, Ut = res.s, res.Vh, res.Ut
        nd = n - d + 1
        eye = torch.eye(r, dtype=torch.float64, device="cuda")
        UtU = torch.zeros((r, r), dtype=torch.float64, device="cuda"); EtU = torch.zeros((nd, r), dtype=torch.float64, device="cuda")
        M = sum(B.shape[1] for B in blocks)
        r0 = 0
        for B in blocks:
            mb = B.shape[1]
            for kd in range(d):                       # rows k*M + (r0 .. r0+mb) of the embedded matrix <- X[rows, t + k]
                Ub = Ut[:, kd * M + r0: kd * M + r0 + mb].double()
                UtU += Ub @ Ub.T
                for j0 in range(0, nd, 2190):
                    j1 = min(nd, j0 + 2190)
                    EtU[j0:j1] += B[j0 + kd:j1 + kd].double() @ Ub.T
            r0 += mb
        err = (EtU - (Vh.T * s)).norm(dim=0)
        print(f"d={d} {typ}: {dt*1e3:.0f} ms; U {tuple(Ut.shape)}; V V^T - I {float((Vh @ Vh.T - eye).abs().max()):.1e}  U^T U - I {float((UtU - eye).abs().max()):.1e}  "
              f"max |E^T u - s v| / s_1 {float((err / s[0]).max()):.1e}  / s_j {float((err / s).max()):.1e}", flush=True)
