"""Scratch: K3 (Z = X^T Y over row blocks) on cfg4-shaped blocks: ms per pass for several l.
DMDX_TN_FORCE_TM (read by make_plan when set) forces the tile height for A/B.
Usage: python scripts/ab_k3.py [NB]"""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
NB = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mb, n = (129780, 8760) if os.environ.get("AB_K3_SHAPE") == "cfg2" else (130872, 3653)
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
blocks = [torch.randn((n, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(NB)]
def run(l, tm):
    if tm: os.environ["DMDX_TN_FORCE_TM"] = str(tm)
    else: os.environ.pop("DMDX_TN_FORCE_TM", None)
    Y = [torch.randn((l, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(NB)]
    K.gemm_tn_blocks(blocks, Y)
    ts = []
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); K.gemm_tn_blocks(blocks, Y); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    os.environ.pop("DMDX_TN_FORCE_TM", None)
    return statistics.median(ts)
print(f"{NB} blocks of {mb} x {n} ({NB*mb*n*4/1e9:.1f} GB per pass)")
cases = [(10, (0,)), (16, (0,)), (20, (0,)), (24, (0,)), (32, (0,)), (60, (0,))] if os.environ.get("AB_K3_SHAPE") == "cfg2" else \
    [(40, (64, 48)), (60, (64,)), (70, (96, 80)), (80, (96, 80)), (100, (128, 112)), (220, (0,))]
for l, tms in cases:
    for tm in tms:
        t = run(l, tm)
        print(f"l={l:4d} tile {tm or 'auto':>4}: {t:8.2f} ms  {2.0*NB*mb*n*l/t/1e9:6.1f} TF algorithmic, {NB*mb*n*4/t/1e9:5.2f} TB/s", flush=True)
