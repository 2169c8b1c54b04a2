"""Scratch: random shapes / views at the fp64 kernels (K8 symm_skinny, K9 gemm_tn64, pack / unpack of the
Gram triangle, K7 / K7L small eigen / one-sided Jacobi, exp_basis) against torch fp64."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dmd_era5_amd.kernels import default_kernels
from dmd_era5_amd import svd as S
K = default_kernels()
import _ws_guard
_ws_guard.install(K)     # exact-size workspaces with a sentinel band behind them
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rs = np.random.RandomState(977 + int(os.environ.get("DMDX_FUZZ_SEED", "0")))
g = torch.Generator(device="cuda").manual_seed(7)
bad = 0

def mat(r, c, pad_choices=(0, 0, 1, 2, 4), off_choices=(0, 0, 1, 2)):
    pad = int(rs.choice(pad_choices)); off = int(rs.choice(off_choices))
    buf = torch.randn(r * (c + pad) + off + 4, generator=g, device="cuda", dtype=torch.float64)
    return buf[off: off + r * (c + pad)].view(r, c + pad)[:, :c]

def flag(name, err, tol):
    global bad
    if not (err <= tol) or not np.isfinite(err):
        bad += 1
        print("BAD", name, err, flush=True)

for i in range(N):
    n = int(rs.choice([2, 3, 16, 63, 64, 130, 257, 1000, 2048, 4099, 8760])); b = int(rs.choice([1, 2, 3, 8, 30, 62, 78, 124, 160, 250, 312]))
    b = min(b, n)
    if i % 20 == 19:
        print("...", i + 1, "rounds,", bad, "flagged so far", flush=True)
    try:
        A = mat(n, n); G = (A + A.T).contiguous() if rs.rand() < 0.5 else mat(n, n)
        G = 0.5 * (G + G.T) if G.is_contiguous() else G            # strided views stay unsymmetric-safe below
        Gs = 0.5 * (G + G.T)
        Gv = Gs if rs.rand() < 0.7 else Gs.clone()
        Q = mat(n, b)
        shift = float(rs.choice([0.0, 1.0, -2.5]))
        Y = K.symm_skinny(Gv, Q, shift)
        ref = Gs @ Q - shift * Q
        flag(f"symm_skinny n={n} b={b}", float((Y - ref).abs().max() / (ref.abs().max() + 1e-300)), 1e-12)
        b2 = int(rs.choice([1, 2, 7, 20, 62, 124])); b2 = min(b2, n)
        B2 = mat(n, b2)
        Cm = K.gemm_tn64(Q, B2)
        ref = Q.T @ B2
        flag(f"gemm_tn64 n={n} b1={b} b2={b2}", float((Cm - ref).abs().max() / (Q.abs().T @ B2.abs()).max()), 1e-13)
        if n <= 4099:
            p = K.pack_triu(Gv); U = K.unpack_triu(p, n)
            flag(f"pack n={n}", float((U - Gs).abs().max()), 0.0)
        nn = int(rs.choice([1, 2, 3, 17, 62, 78, 96, 97, 124, 200, 312, 500]))
        M = mat(nn, nn); T = M @ M.T + 1e-3 * torch.eye(nn, dtype=torch.float64, device="cuda")
        sc = torch.logspace(0, -float(rs.choice([0, 3, 6])), nn, dtype=torch.float64, device="cuda")
        T = sc[:, None] * T * sc[None, :]; T = 0.5 * (T + T.T)
        w, V = S._eigh_desc(T.clone(), K)
        flag(f"eigh_desc n={nn}", float((T @ V - V * w).abs().max() / T.abs().max()), 1e-12 * max(1, nn / 64))
        flag(f"eigh_desc orth n={nn}", float((V.T @ V - torch.eye(nn, dtype=torch.float64, device='cuda')).abs().max()), 1e-12 * max(1, nn / 64))
        if 2 <= nn <= K.svd_jacobi_max_n:
            Cm = mat(nn, nn) * sc[None, :]
            sig, Zt = K.svd_jacobi(Cm.T.contiguous().clone())
            ref = torch.from_numpy(np.linalg.svd(Cm.cpu().numpy(), compute_uv=False)).cuda()   # LAPACK on the host: rocSOLVER's gesvd is 1e-11 s_1 off at n = 500
            lib = torch.linalg.svdvals(Cm)
            if float(((lib - ref).abs() / ref[0]).max()) > 1e-12: print(f"   (library svdvals n={nn}: {float(((lib - ref).abs() / ref[0]).max()):.1e} of s_1 from LAPACK)", flush=True)
            flag(f"svd_jacobi n={nn}", float(((sig - ref).abs() / ref[0]).max()), 1e-13 * max(1, nn / 64))
            flag(f"svd_jacobi orth n={nn}", float((Zt @ Zt.T - torch.eye(nn, dtype=torch.float64, device='cuda')).abs().max()), 1e-12 * max(1, nn / 64))
        r = int(rs.choice([1, 2, 7, 40, 200])); nt = int(rs.choice([1, 5, 100, 8760]))
        al = (torch.randn(r, generator=g, device="cuda", dtype=torch.float64) * 0.01 + 1j * torch.randn(r, generator=g, device="cuda", dtype=torch.float64) * 3).to(torch.complex128)
        t = torch.sort(torch.rand(nt, generator=g, device="cuda", dtype=torch.float64) * 50).values
        Phi, W = K.exp_basis(al, t, torch.complex64 if i % 2 else torch.complex128)
        ref = torch.exp(t[:, None] * al[None, :])
        flag(f"exp_basis n={nt} r={r}", float((Phi.to(torch.complex128) - ref).abs().max()), 2e-6 if i % 2 else 1e-13)
        bad += _ws_guard.check(f"round {i} n={n} b={b}")
    except Exception as e:
        bad += 1
        print("EXC", i, n, b, repr(e)[:300], flush=True)
print("done", N, "rounds,", bad, "flagged")
