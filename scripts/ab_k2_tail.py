"""Scratch: K2 with 4-column tail blocks (16 c + 4 t columns of MFMA work) against the padded bodies
(DMDX_K2_NO_TAIL=1), plain products (no fused Gram): correctness against fp64 on a small case, then ms per pass over
NB row blocks of cfg2 / cfg4 shape.  Usage: python scripts/ab_k2_tail.py [cfg2|cfg4] [NB]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 8
mb, n = (130872, 3653) if which == "cfg4" else (129780, 8760)
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
def mode(tail):
    if tail: os.environ.pop("DMDX_K2_NO_TAIL", None)
    else: os.environ["DMDX_K2_NO_TAIL"] = "1"
# ---- correctness: ragged shapes, every eligible l
for (ms, ns) in ((1000, 77), (4099, 130), (70003, 301)):
    Xs = torch.randn((ns, ms + (-ms) % 4), generator=g, device="cuda", dtype=torch.float32)[:, :ms]
    Xp = torch.zeros((ns, ms + 4), device="cuda", dtype=torch.float32); Xp[:, :ms] = Xs
    for view, tag in ((Xp[:, :ms], "padded ld"), (Xs.contiguous(), "contiguous")):
        for l in (17, 18, 20, 21, 24, 33, 36, 40, 50, 52, 56, 65, 70, 72):
            W = torch.randn((l, ns), generator=g, device="cuda", dtype=torch.float32)
            ref = (W.double() @ view.double())
            mode(True); Y = K.skinny(view, W)
            err = float((Y.double() - ref).abs().max() / ref.abs().max())
            if err > 2e-6: print(f"MISMATCH m={ms} n={ns} l={l} {tag}: {err:.2e}", flush=True)
print("correctness sweep done", flush=True)
blocks = [torch.randn((n, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(NB)]
def run(l):
    W = K.pitch(torch.randn((l, n), generator=g, device="cuda", dtype=torch.float32))
    outs = [torch.empty((l, mb), device="cuda", dtype=torch.float32) for _ in range(2)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i, B in enumerate(blocks): K.skinny(B, W, out=outs[i & 1])
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)
print(f"{which}: {NB} blocks of {mb} x {n} ({NB*mb*n*4/1e9:.1f} GB per pass)")
for l in (20, 24, 36, 50, 52, 70, 72):
    res = {True: [], False: []}
    for rnd in range(5):
        for tail in (False, True):
            mode(tail); t = run(l)
            if rnd: res[tail].append(t)
    a, b = statistics.median(res[False]), statistics.median(res[True])
    print(f"l={l:3d}: padded body {a:7.2f} ms, with 4-column blocks {b:7.2f} ms  ({100*(b-a)/a:+.1f} %), {NB*mb*n*4/b/1e6:.0f} GB/s of X", flush=True)
