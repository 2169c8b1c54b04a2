"""Scratch: what padding row blocks to a multiple of 4 space points buys (era5_svd._upload_variable
pad4): standard and randomized SVD on 4 row blocks of 129 781 space points x 4380 snapshots,
as they are (leading dimension 129 781: register-staged kernel bodies) and widened by 3 zero rows."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd import svd as dsvd
from dmd_era5_amd.kernels import default_kernels
kern = default_kernels()
n, r, nb = 4380, 129781, 4
g = torch.Generator(device="cuda").manual_seed(0)
plain = []
for j in range(nb):
    B = torch.randn((n, r), device="cuda", dtype=torch.float32, generator=g)
    B += 3 * torch.sin(torch.arange(n, device="cuda", dtype=torch.float32) / 50)[:, None] * torch.cos(torch.arange(r, device="cuda", dtype=torch.float32) / 999)[None, :]
    plain.append(B)
padded = []
for B in plain:
    P = torch.zeros((n, r + 3), device="cuda", dtype=torch.float32)
    P[:, :r].copy_(B)
    padded.append(P)
for name, fn in (("standard rank 20", lambda bl: dsvd.svd_snapshots(bl, 20, kern=kern)),
                 ("randomized rank 20", lambda bl: dsvd.svd_randomized(bl, 20, random_state=0, kern=kern))):
    out = {}
    for tag, bl in (("as is", plain), ("padded", padded)):
        fn(bl); torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); res = fn(bl); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        out[tag] = (min(ts), res)
    s0, s1 = out["as is"][1].s, out["padded"][1].s
    print(f"{name}: as is {out['as is'][0]:.1f} ms, padded {out['padded'][0]:.1f} ms; max rel. difference of s {float(((s0 - s1).abs() / s0).max()):.1e}", flush=True)
