"""Scratch: stage timings of the standard path at sizes between the mock and cfg2 (where launch-bound
small stages, not the Gram, decide the time)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
for (m, n, k, d) in ((1038240, 720, 20, 1), (1038240, 720, 20, 2), (1038240, 2000, 50, 1), (259560, 8760, 50, 1), (5184, 120, 10, 2)):
    blocks = bench.make_snapshot_blocks(m, n, 5, torch.device("cuda"))
    for B in blocks: K.row_center_scale_(B, False)
    dsvd.svd_snapshots(blocks, k, delay=d, kern=K); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): res = dsvd.svd_snapshots(blocks, k, delay=d, kern=K)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    r = dsvd.svd_snapshots(blocks, k, delay=d, kern=K, timings=True)
    t = {a: round(b * 1e3, 2) for a, b in r.info.items() if a.startswith("t_")}
    print(f"m={m} n={n} k={k} d={d}: {dt*1e3:.2f} ms per SVD ({m*n*4/dt/1e9:.1f} GB/s); stages {t}; eig {r.info.get('eig_method')}", flush=True)
    del blocks
