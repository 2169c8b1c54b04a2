"""Scratch: where the non-kernel time of svd_randomized (cfg2, sklearn defaults) goes."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as S
from dmd_era5_amd.kernels import default_kernels
kern = default_kernels()
m, n, r, _ = bench.WORKLOADS["cfg2"]
blocks = bench.make_snapshot_blocks(m, n, 1234, torch.device("cuda"))
for B in blocks: kern.row_center_scale_(B, False)
for _ in range(2): S.svd_randomized(blocks, r, random_state=0, kern=kern)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    S.svd_randomized(blocks, r, random_state=0, kern=kern); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=22, max_name_column_width=70))
