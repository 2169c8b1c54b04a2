"""Scratch: the Gram of Y at small l -- fused into K2 (the old 32x32x2 body's epilogue) against a plain K2 pass
followed by one batched Gram launch over the Y blocks.  8 cfg2 blocks."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
mb, n, NB = 129780, 8760, 8
g = torch.Generator(device="cuda").manual_seed(1)
blocks = [torch.randn((n, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(NB)]
def tm(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
    return statistics.median(ts)
for l in (10, 20, 32):
    W = K.pitch(torch.randn((l, n), generator=g, device="cuda", dtype=torch.float32))
    outs = [torch.empty((l, mb), device="cuda", dtype=torch.float32) for _ in range(NB)]
    G = torch.zeros((l, l), dtype=torch.float64, device="cuda")
    def fused():
        for B, o in zip(blocks, outs): K.skinny(B, W, out=o, gram=G)
    def plain():
        for B, o in zip(blocks, outs): K.skinny(B, W, out=o)
    def gram_only():
        return K.syrk_blocks(outs)
    tf, tp, tg = tm(fused), tm(plain), tm(gram_only)
    G.zero_(); fused(); G2 = gram_only()
    err = float((G - G2).abs().max() / G.abs().max())
    print(f"l={l}: fused {tf:.2f} ms; plain {tp:.2f} + batched Gram of Y {tg:.3f} = {tp + tg:.2f} ms; Grams differ by {err:.1e}", flush=True)
