"""Scratch micro-benchmark of the tall-skinny projection kernel (K2): GB/s of X streamed."""
import argparse, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=1038240)
ap.add_argument("--n", type=int, default=8760)
ap.add_argument("--l", type=int, default=62)
ap.add_argument("--blocks", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--streams", type=int, default=1)
a = ap.parse_args()
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
mb = a.m // a.blocks
Xb = [torch.randn((a.n, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(a.blocks)]
Wt = torch.randn((a.l, a.n), generator=g, device="cuda", dtype=torch.float32)
for B in Xb: K.skinny(B, Wt)
torch.cuda.synchronize()
ms = []
for _ in range(a.reps):
    K.events = []
    for B in Xb: Y = K.skinny(B, Wt)
    torch.cuda.synchronize()
    ms.append(sum(e0.elapsed_time(e1) for _, _, e0, e1 in K.events))
gb = a.m * a.n * 4 / 1e9
fl = 2.0 * a.m * a.n * a.l
if a.streams > 1:
    import time
    sts = [torch.cuda.Stream() for _ in range(a.streams)]
    K.events = None
    best = 1e9
    for _ in range(a.reps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i, B in enumerate(Xb):
            with torch.cuda.stream(sts[i % a.streams]):
                Y = K.skinny(B, Wt)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) * 1e3)
    print(f"  with {a.streams} streams (wall): {best:.2f} ms")
print(f"skinny m={a.m} n={a.n} l={a.l} blocks={a.blocks}: {min(ms):.2f} ms best -> {gb/min(ms)*1e3/1e3:.2f} TB/s of X, {fl/min(ms)/1e9:.1f} TFLOP/s", flush=True)
