"""Scratch: where the eigen stage goes at rank 200 (l = 250, block 500) on a power-law Gram, n = 8760."""
import os, sys, time, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd import svd as S
from dmd_era5_amd.kernels import default_kernels
from torch.profiler import profile, ProfilerActivity
from torch.autograd import DeviceType
kern = default_kernels(); dev = torch.device("cuda")
n, k = 8760, int(sys.argv[1]) if len(sys.argv) > 1 else 200
l = k + max(8, k // 4)
g = torch.Generator(device=dev); g.manual_seed(5)
Q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64, device=dev, generator=g))
i = torch.arange(1, n + 1, dtype=torch.float64, device=dev)
G = (Q * i ** -2.0) @ Q.T; G = 0.5 * (G + G.T)
for rep in range(2):
    info = {}; torch.cuda.synchronize(); t0 = time.perf_counter()
    w, V = S.top_eigh(G, l, info=info, kern=kern); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"top_eigh l={l}: {dt*1e3:.1f} ms", info)
kern.events = []
S.top_eigh(G, l, kern=kern); torch.cuda.synchronize()
agg = collections.OrderedDict()
for name, shape, e0, e1 in kern.events:
    a = agg.setdefault((name, tuple(shape)), [0, 0.0]); a[0] += 1; a[1] += e0.elapsed_time(e1)
kern.events = None
for key, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]): print("  ours", key, c, round(t, 2))
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    S.top_eigh(G, l, kern=kern); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == DeviceType.CUDA]
ev.sort(key=lambda e: e.time_range.start)
busy = sum(e.time_range.end - e.time_range.start for e in ev); span = ev[-1].time_range.end - ev[0].time_range.start
print(f"device span {span/1e3:.1f} ms, sum of kernel time {busy/1e3:.1f} ms, {len(ev)} events")
tk = collections.Counter(); ck = collections.Counter()
for e in ev: tk[e.name[:80]] += e.time_range.end - e.time_range.start; ck[e.name[:80]] += 1
for name, t in tk.most_common(16): print(f"  {t/1e3:7.2f} ms x{ck[name]:4d} {name}")
