"""Optimized DMD (variable projection) on the reduced coordinates of the snapshot SVD.

The reference only *announces* this step ("makes use of the optimized DMD algorithm",
/root/reference/README.md:85, citation :139 -- Askham & Kutz, "Variable projection methods for an
optimized dynamic mode decomposition", SIADS 2018); it contains no code for it and pydmd is not a
dependency (SURVEY.md header table).  BASELINE config 5 asks for it on the rank-r reduced
coordinates H = (U_r^T X)^T = V_r S  (n snapshots x r), which is what this module fits:

        min over alpha in C^r, B in C^{r x r}   || H - Phi(alpha) B ||_F ,   Phi_ij = exp(alpha_j t_i)

by variable projection: B = Phi^+ H is eliminated, the residual R(alpha) = (I - Phi Phi^+) H is
minimised over alpha with Levenberg-Marquardt.  With Phi = U S V^H, W = diag(t) Phi, C = W^H R,
the Gauss-Newton matrix and gradient of the full Golub-Pereyra Jacobian reduce to r x r pieces
(the cross terms vanish because U^H R = 0):

        J^H J = [(P W)^H (P W)] o [conj(B) B^T]  +  [V S^-2 V^H]^T-type term o [conj(C) C^T]
        J^H rho = -g ,   g_j = sum_s C[j, s] conj(B[j, s]) ,        P = I - U U^H

so one iteration costs an orthonormal basis of the n x r matrix Phi (CholeskyQR2; the economy SVD
only when Phi is too ill-conditioned for it) and a handful of (n x r)^H (n x r) products -- all
dense torch ops on whatever device H lives on (complex128 by default; launch-bound small dense
work at n = 8760, r = 200, not a roofline kernel).
Initial eigenvalues come from the trapezoidal-rule DMD of the same data (Askham & Kutz section 3.3).
``num_trials > 0`` adds the bagging of BOP-DMD (Sashidhar & Kutz 2022): refits on random subsets
of the snapshots, eigenvalue mean / std over the trials.

Parity: **unpinned** (no pydmd, no reference code, no reference fixtures).  The tests pin it on
known answers instead: planted damped complex exponentials, uneven sampling, noise.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np
import torch

__all__ = ["OptDMDResult", "trapezoidal_dmd_eigs", "exact_dmd_eigs", "initial_eigs", "optdmd", "bopdmd",
           "reduced_coordinates"]


@dataclass
class OptDMDResult:
    eigs: torch.Tensor          # (r,) continuous-time eigenvalues alpha
    modes: torch.Tensor         # (n_s, r) unit-norm modes in the coordinates of H's columns
    amplitudes: torch.Tensor    # (r,) b_j >= 0, H ~ Phi(alpha) diag(b) modes^T
    rel_error: float
    n_iter: int
    converged: bool
    eigs_std: torch.Tensor | None = None   # bagging only
    info: dict = field(default_factory=dict)

    def reconstruct(self, t: torch.Tensor) -> torch.Tensor:
        """(len(t), n_s) = Phi(alpha) diag(b) modes^T."""
        phi = torch.exp(t.to(self.eigs.real.dtype)[:, None].to(self.eigs.dtype) * self.eigs[None, :])
        return (phi * self.amplitudes.to(self.eigs.dtype)) @ self.modes.T


def reduced_coordinates(s: torch.Tensor, Vh: torch.Tensor) -> torch.Tensor:
    """H = (U_r^T X)^T = V_r S from the SVD factors: (n, r), row i = snapshot i in the U_r basis."""
    return (Vh * s[:, None].to(Vh.dtype)).T.contiguous()


def _phi(alpha: torch.Tensor, t: torch.Tensor, dtype: torch.dtype | None = None) -> torch.Tensor:
    """Phi_ij = exp(alpha_j t_i).  The exponent is always formed in complex128 -- |alpha| t reaches
    ~1.4e4 rad over a year of hourly snapshots at 6 cycles / day, which fp32 would carry to 1e-3 rad
    only -- and the result rounded once to the working dtype."""
    out = dtype or alpha.dtype
    a = alpha.to(torch.complex128)
    tt = (t.real if t.is_complex() else t).to(torch.float64)
    if a.is_cuda and out in (torch.complex64, torch.complex128):
        # one launch of the HIP kernel (dmdx_exp_basis) instead of the outer product / exp / cast chain
        from .kernels import default_kernels

        return default_kernels().exp_basis(a, tt.to(a.device), out, want_w=False)[0]
    return torch.exp(tt[:, None] * a[None, :]).to(out)


def trapezoidal_dmd_eigs(H: torch.Tensor, t: torch.Tensor, r: int) -> torch.Tensor:
    """Initial guess: eigenvalues of the rank-r operator fitted to the trapezoidal rule
    (h_{i+1} - h_i) / dt_i ~ A (h_i + h_{i+1}) / 2   (works for uneven sampling)."""
    X1, X2 = H[:-1].T, H[1:].T                        # (n_s, m-1)
    dt = (t[1:] - t[:-1]).to(H.dtype)
    dX = (X2 - X1) / dt[None, :]
    Xm = 0.5 * (X1 + X2)
    U, S, Vh = torch.linalg.svd(Xm, full_matrices=False)
    r = min(r, int((S > S[0] * 1e-12).sum()))
    U, S, Vh = U[:, :r], S[:r], Vh[:r]
    At = U.conj().T @ dX @ Vh.conj().T / S[None, :]
    return torch.linalg.eigvals(At)


def exact_dmd_eigs(H: torch.Tensor, dt: float, r: int) -> torch.Tensor:
    """Continuous-time eigenvalues log(mu) / dt of the rank-r exact DMD operator h_{i+1} = A h_i
    (uniform sampling).  The trapezoidal rule sees mu through the bilinear map
    lambda = (2 / dt) (mu - 1) / (mu + 1): at omega dt = pi / 2 (6 cycles / day, hourly data) that
    is 27 % off in frequency, far outside the basin of the optimisation (its width is ~1 / T), so
    on uniformly sampled data the one-step operator itself is used."""
    X1, X2 = H[:-1].T, H[1:].T
    U, S, Vh = torch.linalg.svd(X1, full_matrices=False)
    r = min(r, int((S > S[0] * 1e-12).sum()))
    U, S, Vh = U[:, :r], S[:r], Vh[:r]
    At = U.conj().T @ X2 @ Vh.conj().T / S[None, :]
    mu = torch.linalg.eigvals(At)
    return torch.log(mu) / dt


def initial_eigs(H: torch.Tensor, t: torch.Tensor, r: int) -> torch.Tensor:
    """Cold-start eigenvalues (complex128): exact DMD when the sampling is uniform, the
    trapezoidal-rule DMD of Askham & Kutz (section 3.3) otherwise."""
    tr = (t.real if t.is_complex() else t).to(torch.float64)
    dts = tr[1:] - tr[:-1]
    dt = float(dts.mean())
    H128 = H.to(torch.complex128)
    if float((dts - dt).abs().max()) <= 1e-9 * abs(dt):
        return exact_dmd_eigs(H128, dt, r)
    return trapezoidal_dmd_eigs(H128, tr.to(torch.complex128), r)


def _project(alpha, t, H, rank_tol=None, use_qr: bool = True):
    """Variable projection at alpha: (Phi, Q, Ginv, B, R) with Q an orthonormal basis of
    range(Phi), Ginv = (Phi^H Phi)^-1, B = Phi^+ H and R = H - Phi B -- all in H's dtype (alpha
    complex128).  Returns None when Phi is not finite (a trial step into Re(alpha) t > the dtype's
    range).

    Route (round 2): CholeskyQR2 of Phi -- two r x r Grams accumulated in complex128, two
    Cholesky factorisations, two triangular solves: ~10 small launches, ~2 ms at n = 8760,
    r = 200 -- instead of the economy SVD (rocSOLVER gesvd of the 8760 x 200 matrix: ~90 ms of
    launch-bound time per evaluation, i.e. the whole iteration).  Exponentials whose frequencies
    are separated by more than ~1 / T are nearly orthogonal, which is the regime of the fit; when
    the basis is too ill-conditioned for the Gram route (Cholesky fails, or the factor's diagonal
    spans more than 1 / rank_tol) the SVD route takes over, with its truncation of the null
    directions."""
    Phi = _phi(alpha, t, H.dtype)
    if not bool(torch.isfinite(Phi.real).all() and torch.isfinite(Phi.imag).all()):
        return None
    if rank_tol is None:
        rank_tol = 1e-12 if H.dtype == torch.complex128 else 1e-6
    if use_qr:
        Q, Rt = Phi, None
        ok = True
        for _ in range(2):
            Qd = Q.to(torch.complex128)          # the r x r Gram in complex128 whatever the working dtype:
            G = Qd.conj().T @ Qd                  # fp32 sums would cap the usable cond(Phi) at ~3e3
            L, err = torch.linalg.cholesky_ex(G)
            d = torch.diagonal(L).real
            if int(err) != 0 or not bool(torch.isfinite(d).all()) or float(d.min()) < math.sqrt(rank_tol) * float(d.max()):
                ok = False
                break
            Rk = L.conj().T                                           # Q_old = Q_new Rk
            Q = torch.linalg.solve_triangular(Rk.to(H.dtype), Q, upper=True, left=False)
            Rt = Rk if Rt is None else Rk @ Rt
        if ok:
            Rinv = torch.linalg.solve_triangular(Rt, torch.eye(Rt.shape[0], dtype=Rt.dtype, device=Rt.device), upper=True)
            QhH = Q.conj().T @ H
            B = Rinv.to(H.dtype) @ QhH
            R = H - Q @ QhH
            Ginv = (Rinv @ Rinv.conj().T).to(H.dtype)
            return Phi, Q, Ginv, B, R
    U, S, Vh = torch.linalg.svd(Phi, full_matrices=False)
    keep = int((S > S[0] * rank_tol).sum())
    U, S, Vh = U[:, :keep], S[:keep], Vh[:keep]
    UhH = U.conj().T @ H
    B = Vh.conj().T @ (UhH / S[:, None].to(H.dtype))
    R = H - U @ UhH
    Ginv = Vh.conj().T @ (Vh / (S[:, None].to(H.dtype) ** 2))
    return Phi, U, Ginv, B, R


def optdmd(H: torch.Tensor, t: torch.Tensor, r: int, alpha0: torch.Tensor | None = None,
           maxiter: int = 30, tol: float = 1e-6, eps_stall: float | None = None, init_lambda: float = 1.0,
           lamup: float = 2.0, maxlam: int = 52) -> OptDMDResult:
    """Optimized DMD of the rows of H (snapshots at times t) with r exponentials.

    Defaults (maxiter 30, tol 1e-6, eps_stall 1e-12, lambda 1, x2 up to 52 times) are the ones
    commonly used for this algorithm; tol is on the relative residual ||R||_F / ||H||_F.
    ``eps_stall`` (relative gain below which the iteration counts as stalled): None = 1e-12 in
    complex128 and 1e-6 in complex64 -- there the residual norm itself carries ~1e-7 of rounding,
    and a tighter test makes every late iteration walk lambda through all its 52 doublings (one
    projection each) before it gives up: measured 59 instead of 9 ms per iteration at cfg5."""
    cdtype = torch.complex128 if H.dtype in (torch.float64, torch.complex128) else torch.complex64
    rdtype = torch.float64 if cdtype == torch.complex128 else torch.float32
    if eps_stall is None:
        eps_stall = 1e-12 if cdtype == torch.complex128 else 1e-6
    H = H.to(cdtype)
    t = (t.real if t.is_complex() else t).to(device=H.device, dtype=torch.float64)
    # the parameters stay complex128 whatever the working dtype of the n x r matrices is
    alpha = (initial_eigs(H, t, r) if alpha0 is None else alpha0).to(device=H.device, dtype=torch.complex128)
    r = alpha.numel()
    tw = t.to(rdtype).to(cdtype)
    normH = torch.linalg.norm(H)
    lam = float(init_lambda)

    pieces = _project(alpha, t, H)
    if pieces is None:
        raise ValueError("optdmd: exp(alpha0 * t) is not finite in the working dtype")
    Phi, U, Ginv, B, R = pieces
    err = float(torch.linalg.norm(R) / normH)
    n_iter, converged = 0, err < tol
    errs = [err]
    n_proj = 1
    while n_iter < maxiter and not converged:
        n_iter += 1
        W = tw[:, None] * Phi
        PW = W - U @ (U.conj().T @ W)
        C = W.conj().T @ R                                           # (r, n_s)
        A1 = (PW.conj().T @ PW) * (B.conj() @ B.T)
        A2 = Ginv * (C.conj() @ C.T)                                  # (Phi^H Phi)^-1 = V S^-2 V^H
        JtJ = (A1 + A2).to(torch.complex128)
        g = (C * B.conj()).sum(dim=1).to(torch.complex128)
        dg = torch.diagonal(JtJ).real.clamp_min(1e-300).to(torch.complex128)

        def trial(lmb):
            M = JtJ + lmb * torch.diag(dg)
            delta = torch.linalg.solve(M, g)
            a_new = alpha + delta
            pieces = _project(a_new, t, H)
            nonlocal n_proj
            n_proj += 1
            if pieces is None or not bool(torch.isfinite(pieces[4].real).all()):
                return a_new, None, math.inf
            return a_new, pieces, float(torch.linalg.norm(pieces[4]) / normH)

        a_new, pieces, e_new = trial(lam)
        if e_new < err:
            lam = max(lam / lamup, 1e-12)
        else:
            improved = False
            for _ in range(maxlam):
                lam *= lamup
                a_new, pieces, e_new = trial(lam)
                if e_new < err:
                    improved = True
                    break
            if not improved:
                break                                               # stalled: keep the current alpha
        gain = err - e_new
        alpha, (Phi, U, Ginv, B, R), err = a_new, pieces, e_new
        errs.append(err)
        if err < tol:
            converged = True
        elif gain < eps_stall * max(err, 1e-300):
            break
    alpha = alpha.to(cdtype)
    amp = torch.linalg.norm(B, dim=1)
    modes = (B / amp[:, None].clamp_min(1e-300).to(cdtype)).T.contiguous()
    order = torch.argsort(-amp)
    return OptDMDResult(eigs=alpha[order], modes=modes[:, order], amplitudes=amp[order].to(rdtype),
                        rel_error=err, n_iter=n_iter, converged=converged,
                        info={"errors": errs, "lambda": lam, "projections": n_proj})


def _match(reference: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
    """Greedy nearest-neighbour assignment of ``values`` to ``reference`` (both (r,))."""
    out = torch.empty_like(reference)
    free = list(range(values.numel()))
    for i in torch.argsort(-reference.abs()).tolist():
        d = (values[free] - reference[i]).abs()
        j = int(torch.argmin(d))
        out[i] = values[free[j]]
        free.pop(j)
    return out


def bopdmd(H: torch.Tensor, t: torch.Tensor, r: int, num_trials: int = 0, trial_size: float = 0.6,
           seed: int = 0, **kwargs) -> OptDMDResult:
    """Optimized DMD with optional bagging: ``num_trials`` refits on random subsets
    (``trial_size`` of the snapshots, without replacement, kept in time order) started from the
    full-data eigenvalues; the reported eigenvalues are the trial mean, ``eigs_std`` their spread,
    modes / amplitudes come from the projection of the full data on the averaged eigenvalues."""
    base = optdmd(H, t, r, **kwargs)
    if num_trials <= 0:
        return base
    rs = np.random.RandomState(seed)
    m = H.shape[0]
    size = max(2 * r, int(round(trial_size * m))) if trial_size <= 1 else int(trial_size)
    size = min(size, m)
    trials = []
    for _ in range(num_trials):
        idx = torch.from_numpy(np.sort(rs.choice(m, size=size, replace=False))).to(H.device)
        res = optdmd(H[idx], t[idx], r, alpha0=base.eigs, **kwargs)
        trials.append(_match(base.eigs, res.eigs))
    A = torch.stack(trials)
    mean = A.mean(dim=0)
    std = torch.sqrt(((A - mean).abs() ** 2).mean(dim=0))
    cdtype = mean.dtype
    _, _, _, B, R = _project(mean.to(torch.complex128), t.to(H.device), H.to(cdtype))
    amp = torch.linalg.norm(B, dim=1)
    modes = (B / amp[:, None].clamp_min(1e-300).to(cdtype)).T.contiguous()
    order = torch.argsort(-amp)
    return OptDMDResult(eigs=mean[order], modes=modes[:, order], amplitudes=amp[order].to(std.dtype),
                        rel_error=float(torch.linalg.norm(R) / torch.linalg.norm(H)),
                        n_iter=base.n_iter, converged=base.converged, eigs_std=std[order],
                        info={"num_trials": num_trials, "trial_size": size, "base_error": base.rel_error})
