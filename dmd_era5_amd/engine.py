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


def svd_numpy(X: np.ndarray, svd_type: str, n_components: int, device="cuda", **opts):
    """Rank-``n_components`` SVD of X (space x time) on the GPU.

    Returns (U (m, k), s (k,), V (k, n)) like the reference's svd_on_era5.
    ``opts`` are forwarded to :func:`svd_snapshots` / :func:`svd_randomized`
    (e.g. ``random_state``, ``omega``, ``n_oversamples``, ``n_iter``, ``refine``).
    """
    if svd_type not in SUPPORTED_SVD_TYPES:
        raise ValueError(f"SVD type {svd_type} is not supported.")
    out_dtype = X.dtype if X.dtype in (np.float32, np.float64) else np.float64
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
