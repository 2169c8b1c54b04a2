"""Scratch: the stages of the slab ingest in isolation: parallel pread into pinned memory, H2D, device scatter."""
import os, sys, time, tempfile
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd import hdf5_lite
n, nlat, nlon = 1000, 721, 1440
path = os.path.join(tempfile.mkdtemp(prefix="dmdx_parts_"), "slice.nc")
field = np.random.RandomState(0).standard_normal((n, 1, nlat, nlon)).astype(np.float32)
with hdf5_lite.Writer(path) as w:
    w.dataset("time", np.arange(n, dtype=np.int64), ("time",))
    w.dataset("t", field, ("time", "level", "latitude", "longitude"))
del field
r = hdf5_lite.Reader(path)
rows = 64
m_v = nlat * nlon
pin = [torch.empty((rows, m_v), dtype=torch.float32).pin_memory() for _ in range(2)]
dev = torch.empty((rows, m_v), dtype=torch.float32, device="cuda")
dst = torch.empty((n, m_v), dtype=torch.float32, device="cuda")
for thr in (1, 4, 8, 16, 32):
    hdf5_lite.RAW_READ_THREADS = thr
    if r._pool is not None:
        r._pool.shutdown(); r._pool = None
    t0 = time.perf_counter(); nb = 0
    for j0 in range(0, n, rows):
        j1 = min(n, j0 + rows)
        view = pin[0][: j1 - j0].numpy().reshape(j1 - j0, 1, nlat, nlon)
        r.read_slab("t", j0, j1, view); nb += view.nbytes
    dt = time.perf_counter() - t0
    print(f"pread into pinned, {thr:2d} threads: {nb/dt/1e9:.1f} GB/s", flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(16): dev.copy_(pin[0], non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"H2D pinned: {16*pin[0].numel()*4/dt/1e9:.1f} GB/s")
t0 = time.perf_counter()
for j0 in range(0, n - rows, rows): dst[j0:j0 + rows].copy_(dev)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"device slab copy: {(n // rows - 1) * dev.numel()*4/dt/1e9:.0f} GB/s")
# H2D with several streams (several SDMA engines?) and with a kernel copy from mapped host memory
for ns in (1, 2, 4):
    sts = [torch.cuda.Stream() for _ in range(ns)]
    chunks = torch.chunk(pin[0], ns, dim=0); dch = torch.chunk(dev, ns, dim=0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(16):
        for s, a, b in zip(sts, chunks, dch):
            with torch.cuda.stream(s): b.copy_(a, non_blocking=True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"H2D pinned, {ns} streams: {16*pin[0].numel()*4/dt/1e9:.1f} GB/s", flush=True)
big = torch.empty((1 << 30,), dtype=torch.float32).pin_memory()
dbig = torch.empty_like(big, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter(); dbig.copy_(big, non_blocking=True); torch.cuda.synchronize()
print(f"H2D pinned 4 GiB single copy: {big.numel()*4/(time.perf_counter()-t0)/1e9:.1f} GB/s")
