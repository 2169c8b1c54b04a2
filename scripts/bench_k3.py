"""Scratch micro-benchmark of K3 (Z = X^T Y over row blocks, one batched launch): TB/s of X streamed."""
import argparse, sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
ap = argparse.ArgumentParser()
ap.add_argument("--m", type=int, default=2076480)
ap.add_argument("--n", type=int, default=3653)
ap.add_argument("--l", type=int, nargs="+", default=[60, 70, 96, 128, 220])
ap.add_argument("--blocks", type=int, default=16)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
K = default_kernels()
g = torch.Generator(device="cuda").manual_seed(1)
mb = a.m // a.blocks
Xb = [torch.randn((a.n, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(a.blocks)]
for l in a.l:
    Yb = [torch.randn((l, mb), generator=g, device="cuda", dtype=torch.float32) for _ in range(a.blocks)]
    Zt = K.gemm_tn_blocks(Xb, Yb)                      # (l, n)
    ref = sum(Y[:, :4096].double() @ X[:, :4096].double().T for X, Y in zip(Xb, Yb))
    chk = K.gemm_tn_blocks([X[:, :4096] for X in Xb], [Y[:, :4096] for Y in Yb])
    err = float((chk - ref).abs().max() / ref.abs().max())
    torch.cuda.synchronize()
    ms = []
    for _ in range(a.reps):
        K.events = []
        K.gemm_tn_blocks(Xb, Yb)
        torch.cuda.synchronize()
        ms.append(sum(e0.elapsed_time(e1) for _, _, e0, e1 in K.events))
    K.events = None
    t = min(ms)
    print(f"K3 m={a.m} n={a.n} l={l}: {t:.2f} ms -> {a.m*a.n*4/t/1e9:.2f} TB/s of X, {2.0*a.m*a.n*l/t/1e9:.1f} TFLOP/s algorithmic; "
          f"rel err (4096-row check) {err:.1e}", flush=True)
