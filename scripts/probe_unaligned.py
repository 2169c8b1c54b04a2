"""Scratch: what an odd number of space points costs (register-staged K1/K3, scalar-load K2)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
from dmd_era5_amd import svd as dsvd
K = default_kernels()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8760
g = torch.Generator(device="cuda").manual_seed(1)
for m in (259560, 259559):
    X = torch.randn((n, m), generator=g, device="cuda", dtype=torch.float32)
    W = torch.randn((62, n), generator=g, device="cuda", dtype=torch.float32)
    for name, fn in (("syrk", lambda: K.syrk(X)), ("skinny", lambda: K.skinny(X, W))):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        print(f"m={m} {name}: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    Y = K.skinny(X, W)
    fn = lambda: K.gemm_tn(X, Y)
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    print(f"m={m} gemm_tn: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    t0 = time.perf_counter(); r = dsvd.svd_snapshots([X], 50, kern=K); torch.cuda.synchronize()
    print(f"m={m} svd_snapshots: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
    del X, Y
