"""Scratch: host-side profile (cProfile) of one randomized SVD at cfg4 scale, k = 200."""
import sys, os, time, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
kern = default_kernels()
m = int(sys.argv[1]) if len(sys.argv) > 1 else 15 * 721 * 1440
blocks = bench.make_snapshot_blocks(m, 3653, 99, torch.device("cuda"))
for B in blocks: kern.row_center_scale_(B, False)
res = dsvd.svd_randomized(blocks, 200, n_oversamples=20, n_iter=2, random_state=0, kern=kern)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter(); pr.enable()
res = dsvd.svd_randomized(blocks, 200, n_oversamples=20, n_iter=2, random_state=0, kern=kern)
torch.cuda.synchronize(); pr.disable()
print(f"wall {(time.perf_counter()-t0)*1e3:.0f} ms")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
