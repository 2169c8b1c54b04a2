"""Scratch: K1 at cfg2 shape against the chunks-per-unit cap (DMDX_TN_MAX_CPS), interleaved on one box;
times include the reduce kernel (HIP events around the whole batched call)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
n, m, nb = 8760, 1038240, 8
g = torch.Generator(device="cuda").manual_seed(1)
Xb = [torch.randn((n, m // nb), generator=g, device="cuda", dtype=torch.float32) for _ in range(nb)]
def run():
    K.events = []
    K.syrk_blocks(Xb); torch.cuda.synchronize()
    t = sum(e0.elapsed_time(e1) for _, _, e0, e1 in K.events); K.events = None
    return t
res = {}
for rep in range(3):
    for cps in ("1024", "2048", "4096", "1024", "2048", "4096"):
        os.environ["DMDX_TN_MAX_CPS"] = cps
        if rep == 0: run()            # workspace growth outside the timing
        res.setdefault(cps, []).append(run())
for cps, ts in res.items():
    print(f"max_cps {cps:5s}: " + " ".join(f"{t:.1f}" for t in ts) + f"  best {min(ts):.1f} ms; workspace {sum(w.numel() for w in K._ws.values())/1e9:.1f} GB (largest so far)")
