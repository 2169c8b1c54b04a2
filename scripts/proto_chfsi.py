"""Scratch (CPU): Chebyshev-root shifted block iteration for the top-l eigenpairs of a PSD matrix
with a gap-free spectrum; counts products with G, orthonormalisations and Rayleigh-Ritz steps."""
import math, sys, time
import numpy as np
import torch

torch.manual_seed(0)


def make(n, kind):
    Q, _ = torch.linalg.qr(torch.randn(n, n, dtype=torch.float64))
    i = torch.arange(1, n + 1, dtype=torch.float64)
    if kind == "power":
        lam = 1.0 / i ** 2
    elif kind == "slow":
        lam = 1.0 / i
    elif kind == "noise":        # Marchenko-Pastur like bulk, m/n = 118
        g = torch.randn(n, 20 * n, dtype=torch.float64)
        return (g @ g.T) / (20 * n)
    elif kind == "lowrank":
        lam = torch.cat([100.0 ** 2 * 0.81 ** torch.arange(64, dtype=torch.float64), 1e-4 * torch.ones(n - 64, dtype=torch.float64)])
    G = (Q * lam) @ Q.T
    G = 0.5 * (G + G.T)
    # rounding noise of a Gram of fp32 products
    E = torch.randn(n, n, dtype=torch.float64) * 1e-10 * float(lam[0]) / math.sqrt(n) * 3
    return G + 0.5 * (E + E.T)


def orth(Y):
    Q = Y
    for _ in range(2):
        Gm = Q.T @ Q
        L, err = torch.linalg.cholesky_ex(Gm)
        if int(err) != 0:
            Qh, _ = torch.linalg.qr(Y)
            return Qh
        Q = torch.linalg.solve_triangular(L, Q.T, upper=False).T
    return Q


def chfsi(G, l, tol=1e-9, bfac=1.25, max_deg=24, log=print):
    n = G.shape[0]
    b = min(n // 3, l + max(8, int(l * (bfac - 1))))
    nprod = north = nrr = 0
    Q = torch.randn(n, b, dtype=torch.float64)
    Q = orth(G @ Q); nprod += 1; north += 1

    def ritz(Q, Y):
        T = Q.T @ Y
        T = 0.5 * (T + T.T)
        th, Z = torch.linalg.eigh(T)
        th, Z = th.flip(0), Z.flip(1)
        Qn = Q @ Z
        R = Y @ Z - Qn * th
        res = torch.linalg.vector_norm(R, dim=0) / th[0]
        return th, Qn, res

    for it in range(2):
        Q = orth(G @ Q); nprod += 1; north += 1
        Y = G @ Q; nprod += 1
        th, Q, res = ritz(Q, Y); nrr += 1
        log(f"power {it}: max res[:l] {float(res[:l].max()):.2e} nconv {int((res[:l] <= tol).sum())}")
        if float(res[:l].max()) <= tol:
            return th[:l], Q[:, :l], dict(nprod=nprod, north=north, nrr=nrr)
    for outer in range(30):
        c = float(th[b - 1])                    # damp [0, c]
        # slowest wanted column: the last unconverged among the leading l
        bad = torch.nonzero(res[:l] > tol).squeeze(1)
        j = int(bad.max())
        x = 2.0 * float(th[j]) / c - 1.0
        rho = x + math.sqrt(max(x * x - 1.0, 0.0))
        need = math.log(float(res[j]) / tol * 3.0) / math.log(max(rho, 1.0 + 1e-6))
        deg = int(min(max_deg, max(2, math.ceil(need))))
        roots = [0.5 * c * (1.0 + math.cos(math.pi * (2 * i + 1) / (2 * deg))) for i in range(deg)]
        # interleave large / small roots
        order = []
        lo, hi = 0, deg - 1
        while lo <= hi:
            order.append(roots[lo]); lo += 1
            if lo <= hi:
                order.append(roots[hi]); hi -= 1
        for rt in order:
            Q = orth(G @ Q - rt * Q); nprod += 1; north += 1
        Y = G @ Q; nprod += 1
        th, Q, res = ritz(Q, Y); nrr += 1
        log(f"outer {outer}: deg {deg} rho {rho:.3f} cut {c:.3e} max res[:l] {float(res[:l].max()):.2e} at {int(res[:l].argmax())} nconv {int((res[:l] <= tol).sum())}")
        if float(res[:l].max()) <= tol:
            break
    return th[:l], Q[:, :l], dict(nprod=nprod, north=north, nrr=nrr, outer=outer + 1)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    l = int(sys.argv[2]) if len(sys.argv) > 2 else 62
    for kind in sys.argv[3:] or ["power", "slow", "lowrank", "noise"]:
        G = make(n, kind)
        for bfac in (1.25, 2.0):
            t0 = time.perf_counter()
            lam, V, info = chfsi(G, l, bfac=bfac, log=lambda s: None)
            dt = time.perf_counter() - t0
            ref = torch.linalg.eigvalsh(G).flip(0)[:l]
            print(kind, "bfac", bfac, info, "max rel err lam %.2e" % float(((lam - ref).abs() / ref[0]).max()),
                  "orth %.1e" % float((V.T @ V - torch.eye(l, dtype=torch.float64)).abs().max()), f"{dt:.2f}s", flush=True)
