"""Host <-> device glue for the SVD boundary: numpy matrix in, numpy (U, s, V) out.

This is what ``svd_on_era5`` (the reference boundary,
/root/reference/src/dmd_era5/era5_svd/era5_svd.py:230-263) calls.  The matrix is
uploaded once as the ``(time, space)`` fp32 tensor the kernels want -- for the
F-ordered X the reference produces (slice_tools.py:207-211 ends in ``.T``) that
is a plain contiguous copy of ``X.T`` -- and U, s, V come back as numpy arrays
in X's dtype.  No CPU arithmetic path exists: without libdmdx.so / a GPU this
raises.
"""

from __future__ import annotations

import numpy as np
import torch

from . import svd as _svd

SUPPORTED_SVD_TYPES = ("standard", "randomized")


def to_device_matrix(X: np.ndarray, device: torch.device | str = "cuda") -> torch.Tensor:
    """numpy (space, time) -> device fp32 tensor Xt of shape (time, space)."""
    if X.ndim != 2:
        raise ValueError("X must be 2-D (space, time)")
    Xt_host = np.ascontiguousarray(X.T, dtype=np.float32)  # no copy if X is F-ordered fp32
    return torch.from_numpy(Xt_host).to(device)


def to_device_blocks(X: np.ndarray, device: torch.device | str = "cuda") -> list[torch.Tensor]:
    """numpy (space, time) -> list of device row blocks, each (time, rows) fp32 (the
    layout the kernels want for large m, see svd.BLOCK_ROWS).  Uploads block by block."""
    if X.ndim != 2:
        raise ValueError("X must be 2-D (space, time)")
    m = X.shape[0]
    if m <= 2 * _svd.BLOCK_ROWS:
        return [to_device_matrix(X, device)]
    return [to_device_matrix(X[a:b], device) for a, b in _svd.split_rows(m)]


def _flip_by_u(U: np.ndarray, V: np.ndarray):
    idx = np.argmax(np.abs(U), axis=0)
    sg = np.sign(U[idx, np.arange(U.shape[1])])
    sg[sg == 0] = 1
    return U * sg[None, :], V * sg[:, None]


# float64 input up to this many bytes is factored in fp64 (the reference's mock slices and test
# inputs are float64: a few MB); anything larger is ERA5-sized and computed in fp32 like the
# float32 slices the reference downloads
FP64_MAX_BYTES = 2 << 30


def _svd_fp64(X: np.ndarray, svd_type: str, n_components: int, device, random_state=None, omega=None,
              n_oversamples: int = 10, n_iter="auto", power_iteration_normalizer: str = "auto", **_ignored):
    """The same two algorithms in fp64, for float64 input small enough to hold twice: the Gram
    X^T X on the fp64 MFMA path (K9), the eigen stage of the fp32 engine (already fp64), the
    projections through the library's fp64 GEMM.  Singular values agree with LAPACK's to
    ~1e-16 s_1^2 / s_i (the Gram route squares the condition number -- in fp64 that leaves 1e-12
    at s_i / s_1 = 1e-4), vectors to rounding; float64 in, float64 out, as the reference returns
    for its float64 mock data (era5_svd.py:246-259)."""
    from .kernels import default_kernels

    kern = default_kernels()
    dev = torch.device(device)
    m, n = X.shape
    wide = m < n
    A = torch.from_numpy(np.ascontiguousarray(X.T if wide else X, dtype=np.float64)).to(dev)   # (rows, cols), tall
    rows, cols = A.shape
    k = min(n_components, cols)
    if svd_type == "standard":
        G = kern.gemm_tn64(A, A)
        G = 0.5 * (G + G.T)
        if not bool(torch.isfinite(torch.diagonal(G)).all()):
            raise np.linalg.LinAlgError("SVD did not converge")
        l = min(cols, k + max(8, k // 4))
        lam, V = _svd.top_eigh(G, l, tol=1e-14, kern=kern)
        good = lam > lam[0].clamp_min(1e-300) * 1e-28
        s0 = torch.sqrt(torch.where(good, lam, torch.ones_like(lam)))
        inv_s0 = torch.where(good, 1.0 / s0, torch.zeros_like(s0))
        s0 = torch.where(good, s0, torch.zeros_like(s0))
        Up = A @ (V * inv_s0)                                    # (rows, l) = X V S^-1
        Mm = kern.gemm_tn64(Up, Up)
        mu_, Z = _svd._graded_eigh(s0, 0.5 * (Mm + Mm.T), kern)
        mu_, Z = mu_[:k], Z[:, :k]
        s = torch.sqrt(mu_.clamp_min(0.0))
        ok = s > s[0].clamp_min(1e-300) * 1e-14
        inv_s = torch.where(ok, 1.0 / torch.where(ok, s, torch.ones_like(s)), torch.zeros_like(s))
        U = Up @ ((s0[:, None] * Z) * inv_s[None, :])
        Vh = (V @ Z).T
    else:
        l = min(cols, n_components + n_oversamples)
        n_it = _svd.resolve_n_iter(n_components, rows, cols, n_iter)
        if omega is None:
            rs = random_state if isinstance(random_state, np.random.RandomState) else np.random.RandomState(random_state)
            omega = rs.normal(size=(cols, n_components + n_oversamples))[:, :l]
        Q = torch.as_tensor(np.asarray(omega, dtype=np.float64)).to(dev)
        # the normaliser sklearn would take (extmath.py:314-326, 342-353): 'auto' = none for n_iter <= 2,
        # LU above; in fp64 its arithmetic is followed literally (the fp32 engine re-orthonormalises
        # where sklearn's 'auto' would not, because un-normalised fp32 iterates lose the trailing
        # directions; in fp64 they do not)
        norm = power_iteration_normalizer
        if norm == "auto":
            norm = "none" if n_it <= 2 else "LU"
        if norm not in ("none", "LU", "QR"):
            raise ValueError(f"power_iteration_normalizer must be 'auto', 'none', 'LU' or 'QR', got {norm!r}")

        def normalise(M):
            if norm == "QR":
                return torch.linalg.qr(M)[0]
            if norm == "LU":
                P, L, _ = torch.linalg.lu(M)                     # scipy.linalg.lu(..., permute_l=True)[0] = P L
                return P @ L
            return M

        for _ in range(n_it):                                    # extmath.py:349-351
            Q = normalise(A @ Q)
            Q = normalise(A.T @ Q)
        Q, _ = torch.linalg.qr(A @ Q)                            # extmath.py:355
        Uh, s, Vh = torch.linalg.svd(Q.T @ A, full_matrices=False)
        U, s, Vh = (Q @ Uh)[:, :k], s[:k], Vh[:k]
    U, s, Vh = U.cpu().numpy(), s.cpu().numpy(), Vh.cpu().numpy()
    if wide:
        U, Vh = np.ascontiguousarray(Vh.T), np.ascontiguousarray(U.T)
    U, Vh = _flip_by_u(U, Vh)
    return U, s, Vh


def svd_numpy(X: np.ndarray, svd_type: str, n_components: int, device="cuda", **opts):
    """Rank-``n_components`` SVD of X (space x time) on the GPU.

    Returns (U (m, k), s (k,), V (k, n)) like the reference's svd_on_era5.
    ``opts`` are forwarded to :func:`svd_snapshots` / :func:`svd_randomized`
    (e.g. ``random_state``, ``omega``, ``n_oversamples``, ``n_iter``, ``refine``).
    """
    if svd_type not in SUPPORTED_SVD_TYPES:
        raise ValueError(f"SVD type {svd_type} is not supported.")
    out_dtype = X.dtype if X.dtype in (np.float32, np.float64) else np.float64
    if X.dtype == np.float64 and X.nbytes <= FP64_MAX_BYTES and not opts.get("comm"):
        return _svd_fp64(X, svd_type, n_components, device, **opts)
    m, n = X.shape
    wide = m < n
    if wide:
        # sklearn transposes wide inputs (extmath.py:562-566); LAPACK does not care.
        # The tall algorithms run on X^T and the factors swap roles.
        Xt = to_device_blocks(np.ascontiguousarray(X.T), device)  # (space, time) tensors
    else:
        Xt = to_device_blocks(X, device)
    if svd_type == "standard":
        res = _svd.svd_snapshots(Xt, n_components, flip_sign=not wide, **opts)
    else:
        res = _svd.svd_randomized(Xt, n_components, flip_sign=not wide, **opts)
    A = res.Ut.cpu().numpy().T.astype(out_dtype, copy=False)       # (rows of Xt's matrix, k)
    s = res.s.cpu().numpy().astype(out_dtype, copy=False)
    B = res.Vh.cpu().numpy().astype(out_dtype, copy=False)          # (k, cols)
    if wide:
        U, V = np.ascontiguousarray(B.T), np.ascontiguousarray(A.T)
        U, V = _flip_by_u(U, V)
    else:
        U, V = A, B
    return U, s, V
