"""Scratch: the Gram of cfg2 (n = 8760 = 68 x 128 + 56) as K1 on the first 8704 columns + K3 for the last 56
(64-row tiles) instead of K1 with a 128-wide edge tile that is 56 % padding."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(0)
n, m, nb = 8760, 129780, 8
blocks = [torch.randn((n, m), device="cuda", dtype=torch.float32, generator=g) for _ in range(nb)]
n0 = (n // 128) * 128
def tm(fn, reps=4):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return min(ts)
G = torch.empty((n, n), dtype=torch.float64, device="cuda")
t_full = tm(lambda: K.syrk_blocks(blocks, out=None))
G0 = torch.empty((n0, n0), dtype=torch.float64, device="cuda")
heads = [B[:n0] for B in blocks]
tails = [B[n0:] for B in blocks]
t_main = tm(lambda: K.syrk_blocks(heads))
t_edge = tm(lambda: K.gemm_tn_blocks(blocks, tails))
ref = K.syrk_blocks(blocks)
Gm = K.syrk_blocks(heads)
Ge = K.gemm_tn_blocks(blocks, tails)          # (56, 8760) = G[n0:, :]
err = max(float((Gm - ref[:n0, :n0]).abs().max()), float((Ge - ref[n0:, :]).abs().max())) / float(ref.abs().max())
print(f"K1 on {n} columns {t_full:.1f} ms; K1 on {n0} columns {t_main:.1f} ms + K3 56 x {n} {t_edge:.1f} ms = {t_main + t_edge:.1f} ms "
      f"({100 * (t_full - t_main - t_edge) / t_full:+.2f} %); max difference {err:.1e} of max |G|", flush=True)
