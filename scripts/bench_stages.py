"""Scratch: stage timings of the SVD drivers on the cfg2 workload (and randomized)."""
import sys, os, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
kern = default_kernels()
m, n, r, _ = bench.WORKLOADS["cfg2"]
blocks = bench.make_snapshot_blocks(m, n, 1234, torch.device("cuda"))
for B in blocks: kern.row_center_scale_(B, False)
torch.cuda.synchronize()
for _ in range(2):
    res = dsvd.svd_snapshots(blocks, r, kern=kern, timings=True)
print("standard:", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.info.items()}, flush=True)
# finer: eigensolve pieces
G = dsvd._gram_blocks(blocks, kern, dsvd.Comm())
torch.cuda.synchronize(); t = time.perf_counter()
info = {}
lam, V = dsvd.top_eigh(G, 62, info=info)
torch.cuda.synchronize(); print("top_eigh krylov: %.1f ms" % ((time.perf_counter() - t) * 1e3), info, flush=True)
for which in ("randomized",):
    for it in range(2):
        kern.events = []
        torch.cuda.synchronize(); t = time.perf_counter()
        res = dsvd.svd_randomized(blocks, r, random_state=0, kern=kern)
        torch.cuda.synchronize(); dt = time.perf_counter() - t
        ev = {}
        for name, shape, e0, e1 in kern.events:
            ev.setdefault(name, []).append(e0.elapsed_time(e1))
        kern.events = None
    print("randomized (sklearn defaults, n_iter=%d): %.1f ms total -> %.1f GB/s; kernel ms:" % (res.info["n_iter"], dt * 1e3, m * n * 4 / dt / 1e9),
          {k: round(sum(v), 1) for k, v in ev.items()}, {k: len(v) for k, v in ev.items()}, flush=True)
    print(" s head", res.s[:3].tolist())
