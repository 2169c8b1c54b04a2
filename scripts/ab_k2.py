"""Scratch: K2 bodies A/B in one process (cdna_hip_programming.md rule 24): the 32x32x2 body
(DMDX_K2_IMPL=old) against the 16x16x4 body, plain and with the fused Gram, on row blocks of
cfg4's shape (130 870 x 3653) and cfg2's (129 780 x 8760), interleaved rounds, median ms per pass
over NB row blocks.  Usage: python scripts/ab_k2.py [cfg4|cfg2] [NB]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
which = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mb, n = (130872, 3653) if which == "cfg4" else (129780, 8760)
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
blocks = [torch.randn((n, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(NB)]
def run(impl, l, gram):
    os.environ["DMDX_K2_IMPL"] = impl
    W = K.pitch(torch.randn((l, n), generator=g, device="cuda", dtype=torch.float32))
    G = torch.zeros((l, l), dtype=torch.float64, device="cuda") if gram else None
    outs = [torch.empty((l, mb), device="cuda", dtype=torch.float32) for _ in range(2)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i, B in enumerate(blocks):
        K.skinny(B, W, out=outs[i & 1], gram=G)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)
def clock(impl, l, gram):
    ctr = torch.zeros(3, dtype=torch.int64, device="cuda")
    K.clock_probe(ctr)
    try:
        run(impl, l, gram)
    finally:
        K.clock_probe(None)
    cyc, ticks, _ = (int(v) for v in ctr.tolist())
    return 100.0 * cyc / max(ticks, 1)
cases = [(20, False), (32, False), (60, True), (64, False), (70, True), (96, True), (112, False), (128, False), (192, False), (208, False), (220, False), (220, True)]
print(f"{which}: {NB} blocks of {mb} x {n} ({NB*mb*n*4/1e9:.1f} GB per pass)")
for l, gram in cases:
    res = {"old": [], "new": []}
    for rnd in range(5):
        for impl in ("old", "new"):
            if impl == "old" and gram and l > 96:
                continue
            try:
                t = run(impl, l, gram)
            except Exception as e:
                t = float("nan")
            if rnd:
                res[impl].append(t)
    def med(v): return statistics.median(v) if v else float("nan")
    fl = 2.0 * NB * mb * n * l
    o, nw = med(res["old"]), med(res["new"])
    ck = {i: (clock(i, l, gram) if (i == "new" or not (gram and l > 96)) else float("nan")) for i in ("old", "new")}
    print(f"l={l:4d} gram={int(gram)}: old {o:8.2f} ms ({fl/o/1e9 if o==o else 0:6.1f} TF)  new {nw:8.2f} ms ({fl/nw/1e9:6.1f} TF, "
          f"{NB*mb*n*4/nw/1e9:5.2f} TB/s)  ratio {nw/o if o==o else float('nan'):.3f}  clock old {ck['old']:.0f} new {ck['new']:.0f} MHz", flush=True)
