"""Scratch: GPU idle gaps inside one standard cfg2 step (torch.profiler device timeline)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
from torch.profiler import profile, ProfilerActivity
from torch.autograd import DeviceType
k = int(sys.argv[1]) if len(sys.argv) > 1 else 50
kern = default_kernels()
m, n, _, _ = bench.WORKLOADS["cfg2"]
blocks = bench.make_snapshot_blocks(m, n, 1234, torch.device("cuda"))
for B in blocks: kern.row_center_scale_(B, False)
step = lambda: dsvd.svd_snapshots(blocks, k, kern=kern)
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter(); step(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    step(); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == DeviceType.CUDA]
ev.sort(key=lambda e: e.time_range.start)
busy = 0.0; end = ev[0].time_range.start; gaps = []
for i, e in enumerate(ev):
    s, t = e.time_range.start, e.time_range.end
    if s > end:
        gaps.append((s - end, ev[i - 1].name[:60] if i else "", e.name[:60]))
        busy += t - s
    elif t > end:
        busy += t - end
    end = max(end, t)
span = end - ev[0].time_range.start
print(f"k={k}: step {dt*1e3:.1f} ms wall; device span {span/1e3:.1f} ms, busy {busy/1e3:.1f} ms, idle {(span-busy)/1e3:.1f} ms in {len(gaps)} gaps; {len(ev)} device events")
gaps.sort(key=lambda g: -g[0])
for g in gaps[:40]:
    print(f"  {g[0]:8.1f} us   after {g[1]}   before {g[2]}")
import collections
small = collections.Counter(); tsmall = collections.Counter()
for e in ev:
    d = e.time_range.end - e.time_range.start
    small[e.name[:70]] += 1; tsmall[e.name[:70]] += d
print("device time by kernel:")
for name, t in tsmall.most_common(25):
    print(f"  {t/1e3:8.2f} ms  x{small[name]:4d}  {name}")
