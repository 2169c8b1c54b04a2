"""Scratch: K1 / K2 / K3 at production-like shapes (row blocks of 10^5 rows, n up to 8760) with random row
padding / alignment, against torch fp64 products of the same operands."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
import _ws_guard
_ws_guard.install(K)     # exact-size workspaces with a sentinel band behind them
N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rs = np.random.RandomState(7 + int(os.environ.get("DMDX_FUZZ_SEED", "0")))
g = torch.Generator(device="cuda").manual_seed(11)
bad = 0

def view(rows, cols):
    pad = int(rs.choice([0, 0, 4, 8, 1, 3])); off = int(rs.choice([0, 0, 4, 1]))
    buf = torch.randn(rows * (cols + pad) + off + 8, generator=g, device="cuda", dtype=torch.float32)
    return buf[off: off + rows * (cols + pad)].view(rows, cols + pad)[:, :cols]

def check(name, got, ref, absref, tol):
    global bad
    err = (got.double() - ref).abs()
    if not (bool((err <= tol * absref + 1e-30).all()) and bool(torch.isfinite(got).all())):
        bad += 1
        print("BAD", name, float((err / (absref + 1e-300)).max()), flush=True)

for i in range(N):
    m = int(rs.choice([100000, 129780, 130872, 129779, 262144])); n = int(rs.choice([1000, 3653, 8760, 2049]))
    if m > 200000 and n > 4000: n = 3653
    l = int(rs.choice([60, 62, 70, 96, 128, 220, 250]))
    Xt = view(n, m); X = Xt.double()
    aX = X.abs()
    G = K.syrk(Xt)
    check(f"syrk m={m} n={n}", G, X @ X.T, aX @ aX.T, 3e-6)
    nb = int(rs.randint(2, 5)); cuts = np.linspace(0, m, nb + 1).astype(int); cuts = (cuts // 4) * 4; cuts[-1] = m
    Gb = K.syrk_blocks([Xt[:, a:b] for a, b in zip(cuts[:-1], cuts[1:])])
    check(f"syrk_blocks m={m} n={n} nb={nb}", Gb, X @ X.T, aX @ aX.T, 3e-6)
    Wt = view(l, n); Y = K.skinny(Xt, Wt)
    check(f"skinny m={m} n={n} l={l}", Y, Wt.double() @ X, Wt.double().abs() @ aX, (4 + np.sqrt(n)) * 6e-8)
    Z = K.gemm_tn(Xt, Y)                                # (l, n) = (X^T Y)^T
    Yd = Y.double()
    check(f"gemm_tn m={m} n={n} l={l}", Z, Yd @ X.T, Yd.abs() @ aX.T, 3e-6)
    # the batched products at row-block scale, small l included (K3s) and more blocks than one launch takes
    l2 = int(rs.choice([1, 7, 20, 32])); nb2 = int(rs.choice([3, 17]))
    cuts2 = np.linspace(0, m, nb2 + 1).astype(int); cuts2 = (cuts2 // 4) * 4; cuts2[-1] = m
    Xb = [Xt[:, a:b] for a, b in zip(cuts2[:-1], cuts2[1:])]
    Yb = [Y[:l2, a:b].contiguous() for a, b in zip(cuts2[:-1], cuts2[1:])]
    Zb = K.gemm_tn_blocks(Xb, Yb)
    check(f"gemm_tn_blocks m={m} n={n} l={l2} nb={nb2}", Zb, Yd[:l2] @ X.T, Yd[:l2].abs() @ aX.T, 3e-6)
    bad += _ws_guard.check(f"case {i} m={m} n={n} l={l}")
    del X, aX, G, Gb, Y, Z, Yd, Xt
    print("ok", i, m, n, l, flush=True)
print("done", N, "cases,", bad, "flagged")
