"""Scratch: one GPU's share of cfg3 (15 573 600 / 8 rows x 8760, rank 200, standard), stage timings."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
kern = default_kernels()
m, n, r = 15573600 // 8, 8760, 200
blocks = bench.make_snapshot_blocks(m, n, 1234, torch.device("cuda"))
for B in blocks: kern.row_center_scale_(B, False)
torch.cuda.synchronize()
for _ in range(2):
    res = dsvd.svd_snapshots(blocks, r, kern=kern, timings=True)
print("cfg3 shard:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.info.items()}, flush=True)
print("GB/s", m * n * 4 / res.info["t_total"] / 1e9, "peak HBM GB", torch.cuda.max_memory_allocated() / 1e9)
