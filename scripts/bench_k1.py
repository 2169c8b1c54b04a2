"""Scratch micro-benchmark of the Gram kernel (K1) alone: TFLOP/s vs the fp32 MFMA peak."""
import argparse, sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=259560)
ap.add_argument("--n", type=int, default=8760)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--blocks", type=int, default=1)
ap.add_argument("--batched", action="store_true", help="one launch for all blocks (dmdx_syrk_blocks_f32)")
ap.add_argument("--group", type=int, default=0, help="with --batched: blocks per launch (0 = all)")
a = ap.parse_args()
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
mb = a.m // a.blocks
Xb = [torch.randn((a.n, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(a.blocks)]
def gram():
    if a.batched:
        if a.group <= 0:
            return K.syrk_blocks(Xb)
        G = K.syrk_blocks(Xb[:a.group])
        for i in range(a.group, len(Xb), a.group):
            K.syrk_blocks(Xb[i:i + a.group], out=G)
        return G
    G = K.syrk(Xb[0])
    for B in Xb[1:]: K.syrk(B, out=G)
    return G
G = gram(); torch.cuda.synchronize()
ms = []
for _ in range(a.reps):
    K.events = []
    G = gram(); torch.cuda.synchronize()
    ms.append(sum(e0.elapsed_time(e1) for _, _, e0, e1 in K.events))
fl = a.m * a.n * (a.n + 1)
print(("batched " if a.batched else "") + f"syrk m={a.m} n={a.n} blocks={a.blocks}: {min(ms):.2f} ms best, {sum(ms)/len(ms):.2f} avg -> {fl/min(ms)/1e9:.1f} TFLOP/s best ({fl/min(ms)/1e9/157.3*100:.1f}% of 157.3)", flush=True)
