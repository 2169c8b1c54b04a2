"""Scratch: where a randomized cfg2 step goes, by kernel and shape (HIP events around every launch)."""
import os, sys, time, json, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
kern = default_kernels()
m, n, _, _ = bench.WORKLOADS["cfg2"]
blocks = bench.make_snapshot_blocks(m, n, 1234, torch.device("cuda"))
for B in blocks: kern.row_center_scale_(B, False)
step = lambda: dsvd.svd_randomized(blocks, k, random_state=0, kern=kern)
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
kern.events = []; step(); torch.cuda.synchronize(); ev, kern.events = kern.events, None
agg = collections.OrderedDict()
for name, shape, e0, e1 in ev:
    key = (name, tuple(shape[1:]) if name in ("skinny",) else tuple(shape))
    a = agg.setdefault(key, [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
print(f"k={k}: {dt*1e3:.1f} ms per step; kernel time by (name, shape): count, total ms")
tot = 0.0
for key, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  ", key, c, round(t, 2)); tot += t
print("   sum of our kernels", round(tot, 1), "ms; rest (torch small dense, gaps)", round(dt * 1e3 - tot, 1))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
