"""Scratch probe: cost of the dense small-matrix pieces on the device."""
import time, torch
dev = "cuda"
for n in (2048, 4096, 8760):
    A = torch.randn(n, n + 64, dtype=torch.float64, device=dev)
    G = A @ A.T
    torch.cuda.synchronize(); t = time.perf_counter()
    lam, V = torch.linalg.eigh(G)
    torch.cuda.synchronize(); print(f"eigh fp64 n={n}: {time.perf_counter()-t:.3f}s", flush=True)
    Q = torch.randn(n, 210, dtype=torch.float64, device=dev)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): Y = G @ Q
    torch.cuda.synchronize(); print(f"  G@Q(210) fp64: {(time.perf_counter()-t)/10*1e3:.3f} ms", flush=True)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5): Qo, _ = torch.linalg.qr(Y)
    torch.cuda.synchronize(); print(f"  qr n x 210 fp64: {(time.perf_counter()-t)/5*1e3:.3f} ms", flush=True)
