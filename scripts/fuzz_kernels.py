"""Scratch: random shapes / leading dimensions / pointer alignments straight at the kernel provider
(K1 syrk + syrk_blocks, K3 gemm_tn + gemm_tn_blocks, K2 skinny, K5, K6, K7) against torch fp64."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
K = default_kernels()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
rs = np.random.RandomState(123 + int(os.environ.get("DMDX_FUZZ_SEED", "0")))
g = torch.Generator(device="cuda").manual_seed(5)
bad = 0

import _ws_guard
_ws_guard.install(K)     # exact-size workspaces with a sentinel band behind them

def view(rows, cols):
    """(rows, cols) fp32 view with random row padding and a random element offset (alignment 4..16 B)."""
    pad = int(rs.choice([0, 0, 1, 3, 4, 8])); off = int(rs.choice([0, 0, 1, 2, 4]))
    buf = torch.randn(rows * (cols + pad) + off + 8, generator=g, device="cuda", dtype=torch.float32)
    return buf[off: off + rows * (cols + pad)].view(rows, cols + pad)[:, :cols]

def check(name, got, ref, absref, tol=2e-6):
    global bad
    err = (got.double() - ref).abs()
    ok = bool((err <= tol * absref + 1e-30).all()) and bool(torch.isfinite(got).all())
    if not ok:
        bad += 1
        print("BAD", name, float((err / (absref + 1e-300)).max()), flush=True)

for i in range(N):
    m = int(rs.choice([1, 5, 31, 32, 33, 127, 129, 1000, 4097, 30001])); n = int(rs.choice([1, 2, 31, 64, 65, 128, 129, 260, 500]))
    l = int(rs.choice([1, 2, 16, 17, 20, 24, 25, 31, 32, 33, 36, 50, 64, 65, 70, 72, 100, 128, 130, 200]))
    if i % 100 == 99:
        print("...", i + 1, "rounds,", bad, "flagged so far", flush=True)
    try:
        Xt = view(n, m)                                   # (n, m): X is m x n
        X = Xt.double()
        G = K.syrk(Xt)
        check(f"syrk m={m} n={n}", G, X @ X.T, X.abs() @ X.abs().T)
        nb = int(rs.randint(2, 20)); sizes = [int(rs.randint(1, 3000)) for _ in range(nb)]
        blocks = [view(n, mb) for mb in sizes]
        Gb = K.syrk_blocks(blocks)
        ref = sum(B.double() @ B.double().T for B in blocks); aref = sum(B.double().abs() @ B.double().abs().T for B in blocks)
        check(f"syrk_blocks n={n} nb={nb}", Gb, ref, aref)
        At = view(l, m); C = K.gemm_tn(At, Xt)            # (n, l)
        check(f"gemm_tn m={m} na={l} nb={n}", C, X @ At.double().T, X.abs() @ At.double().abs().T)
        Ab = [view(l, mb) for mb in sizes]
        Cb = K.gemm_tn_blocks(Ab, blocks)
        ref = sum(B.double() @ A.double().T for A, B in zip(Ab, blocks)); aref = sum(B.double().abs() @ A.double().abs().T for A, B in zip(Ab, blocks))
        check(f"gemm_tn_blocks n={n} l={l} nb={nb}", Cb, ref, aref)
        Wt = view(l, n); Y = K.skinny(Xt, Wt)             # (l, m)
        check(f"skinny m={m} n={n} l={l}", Y, Wt.double() @ X, Wt.double().abs() @ X.abs(), tol=4e-6)
        if n >= 3:
            d = int(rs.randint(1, min(n, 4) + 1))
            Gd = K.delay_shift_sum(G, d); nd = n - d + 1
            ref = sum(G[k:k + nd, k:k + nd] for k in range(d))
            check(f"shift_sum n={n} d={d}", Gd, ref, ref.abs() + 1e-300, tol=1e-14)
        if n <= 96:
            w, V = K.eigh_small(G)
            R = G @ V - V * w
            check(f"eigh_small n={n}", R, torch.zeros_like(R), torch.full_like(R, float(G.abs().max())), tol=1e-12)
        Xc = Xt.contiguous().clone(); mean, std = K.row_center_scale_(Xc, bool(i % 2))
        refm = X.mean(dim=0)
        check(f"center m={m} n={n}", mean, refm, refm.abs() + X.abs().mean(dim=0), tol=1e-6)
        # guard bands: in-place K5 on a padded view and K2 into a padded `out` view must leave the
        # padding columns (and the words before / behind the view) untouched
        pad = int(rs.choice([1, 3, 4, 8])); off = int(rs.choice([0, 1, 2, 4]))
        SENT = 12345.678
        buf = torch.full((n * (m + pad) + off + 8,), SENT, device="cuda", dtype=torch.float32)
        Xg = buf[off: off + n * (m + pad)].view(n, m + pad)[:, :m]
        Xg.copy_(Xt)
        K.row_center_scale_(Xg, bool(i % 2))
        whole = buf[off: off + n * (m + pad)].view(n, m + pad)
        if not (bool((whole[:, m:] == SENT).all()) and bool((buf[:off] == SENT).all()) and bool((buf[off + n * (m + pad):] == SENT).all())):
            bad += 1; print("BAD guard K5", m, n, pad, off, flush=True)
        if n > 1 or not (i % 2):     # (one sample, scaled: 0 / 0 = NaN on both paths, as in numpy)
            check(f"center view m={m} n={n}", Xg, Xc.double(), Xc.double().abs() + 1e-30, tol=1e-6)
        obuf = torch.full((l * (m + pad) + off + 8,), SENT, device="cuda", dtype=torch.float32)
        Og = obuf[off: off + l * (m + pad)].view(l, m + pad)[:, :m]
        K.skinny(Xt, Wt, out=Og)
        ow = obuf[off: off + l * (m + pad)].view(l, m + pad)
        if not (bool((ow[:, m:] == SENT).all()) and bool((obuf[:off] == SENT).all()) and bool((obuf[off + l * (m + pad):] == SENT).all())):
            bad += 1; print("BAD guard K2", m, n, l, pad, off, flush=True)
        check(f"skinny out view m={m} n={n} l={l}", Og, Wt.double() @ X, Wt.double().abs() @ X.abs(), tol=4e-6)
        Gf = torch.zeros((l, l), dtype=torch.float64, device="cuda")
        if l <= K.skinny_gram_max_l:
            Yg = K.skinny(Xt, Wt, gram=Gf)
            check(f"skinny+gram m={m} n={n} l={l}", Gf, Yg.double() @ Yg.double().T, Yg.double().abs() @ Yg.double().abs().T, tol=4e-6)
        bad += _ws_guard.check(f"round {i} m={m} n={n} l={l} nb={nb}")
    except Exception as e:
        bad += 1
        print("EXC", i, m, n, l, repr(e)[:300], flush=True)
print("done", N, "rounds,", bad, "flagged")
