"""Scratch: are the per-block K2 launches of a pass back to back?  16 cfg4 blocks, l = 70 / 220 (fused Gram as in
the range finder): wall time of the 16 launches against the sum of their kernel times (rocprofv3 --kernel-trace
gives the same split; here: one event pair around the loop, and the loop repeated under a HIP graph-free stream)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
mb, n, NB = 130872, 3653, 16
g = torch.Generator(device="cuda").manual_seed(1)
blocks = [torch.randn((n, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(NB)]
for l in (70, 220):
    Wt = K.pitch(torch.randn((l, n), generator=g, device="cuda", dtype=torch.float32))
    outs = [torch.empty((l, mb), device="cuda", dtype=torch.float32) for _ in range(NB)]
    G = torch.zeros((l, l), device="cuda", dtype=torch.float64)
    fused = l <= K.skinny_gram_max_l
    def one(j):
        K.skinny(blocks[j], Wt, out=outs[j], gram=G if fused else None)
    for j in range(NB): one(j)
    torch.cuda.synchronize()
    walls, sums = [], []
    for rep in range(5):
        K.events = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for j in range(NB): one(j)
        e1.record(); e1.synchronize(); walls.append(e0.elapsed_time(e1))
        K.events = []
        for j in range(NB): one(j)
        torch.cuda.synchronize()
        sums.append(sum(a.elapsed_time(b) for _, _, a, b in K.events)); K.events = None
    print(f"l={l} ({'fused Gram' if fused else 'plain'}): {NB} launches back to back {statistics.median(walls):.2f} ms; sum of per-launch event pairs {statistics.median(sums):.2f} ms", flush=True)
