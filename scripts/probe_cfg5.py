"""Scratch: cfg5 (optimized DMD, n = 8760, r = 200) -- repeated timings per dtype and a kernel-level
breakdown of one fit (torch profiler)."""
import sys, os, time, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd import bopdmd as bop
from torch.profiler import profile, ProfilerActivity
from torch.autograd import DeviceType
rs = np.random.RandomState(0)
r, n = 200, 8760
t = np.arange(n) / 24.0
freq = np.sort(rs.uniform(0.02, 6.0, r // 2))
alpha = -rs.uniform(1e-4, 3e-3, r // 2) + 1j * 2 * np.pi * freq
alpha = np.concatenate([alpha, alpha.conj()])
modes = rs.standard_normal((r, r)) + 1j * rs.standard_normal((r, r))
H = np.exp(np.outer(t, alpha)) @ modes + 1e-2 * rs.standard_normal((n, r))
td = torch.from_numpy(t).cuda()
for dt in (torch.complex64, torch.complex128, torch.complex64, torch.complex128):
    Hd = torch.from_numpy(H).cuda().to(dt)
    a0 = bop.initial_eigs(Hd, td, r)
    for rep in range(3):
        torch.cuda.synchronize(); t1 = time.perf_counter()
        res = bop.optdmd(Hd, td, r, alpha0=a0, tol=1e-9, maxiter=40)
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(f"{dt} rep {rep}: {res.n_iter} iterations, {res.info['projections']} projections in {1e3*(t2-t1):.1f} ms "
              f"({1e3*(t2-t1)/res.info['projections']:.2f} ms / projection), route {res.info['route']}, lambda {res.info['lambda']:.2e}", flush=True)
Hd = torch.from_numpy(H).cuda()
a0 = bop.initial_eigs(Hd, td, r)
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    res = bop.optdmd(Hd, td, r, alpha0=a0, tol=1e-9, maxiter=40); torch.cuda.synchronize()
ev = [e for e in prof.events() if e.device_type == DeviceType.CUDA]
ev.sort(key=lambda e: e.time_range.start)
busy = sum(e.time_range.end - e.time_range.start for e in ev); span = ev[-1].time_range.end - ev[0].time_range.start
print(f"device span {span/1e3:.1f} ms, sum of kernel time {busy/1e3:.1f} ms, {len(ev)} events, {res.info['projections']} projections")
tk = collections.Counter(); ck = collections.Counter()
for e in ev: tk[e.name[:90]] += e.time_range.end - e.time_range.start; ck[e.name[:90]] += 1
for name, tt in tk.most_common(14): print(f"  {tt/1e3:7.2f} ms x{ck[name]:4d} {name}")
