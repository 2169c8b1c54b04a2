"""CPU oracle for the ERA5 slice -> snapshot matrix -> rank-r SVD path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dmd_era5_amd/`` may import this
module; it is used by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` as the checker, never as the product.

It restates, on plain numpy arrays (the reference works on xarray objects,
which are not installed here), every arithmetic stage of
``python -m dmd_era5.era5_svd.era5_svd`` in the reference
(paths relative to /root/reference):

=====================  =====================================================
oracle function        reference lines it follows
=====================  =====================================================
standardize            src/dmd_era5/slice_tools/slice_tools.py:144-179
flatten                src/dmd_era5/slice_tools/slice_tools.py:277-365
delay_embed            src/dmd_era5/slice_tools/slice_tools.py:182-211
delay_labels           src/dmd_era5/slice_tools/slice_tools.py:254-272
svd_standard           src/dmd_era5/era5_svd/era5_svd.py:249-255
svd_randomized         src/dmd_era5/era5_svd/era5_svd.py:256-259, which calls
                       sklearn.utils.extmath.randomized_svd (scikit-learn is a
                       third-party dependency, unpinned in pyproject.toml:40;
                       version installed here: 1.7.2, extmath.py:287-357,
                       531-604, 895-953)
svd_flip               sklearn/utils/extmath.py:895-953 (u-based branch)
mock_era5              src/dmd_era5/create_mock_data/create_mock_data.py:26-155
preprocess             src/dmd_era5/era5_svd/era5_svd.py:384-414
=====================  =====================================================

Pinning (see tests/test_oracle_golden.py):
  * delay_embed   -- the four known answers of
                     tests/test_02_slice_tools.py:215-231 (tests/golden/
                     delay_embedding_cases.npz) and the invalid-input errors
                     of :234-262.
  * standardize   -- mean 0 / std 1 at atol 1e-6, tests/test_02:108-174.
  * flatten       -- ordering spot checks, tests/test_02:291-333.
  * svd_standard  -- IS the reference's call (np.linalg.svd + truncation).
  * svd_randomized-- checked against sklearn's randomized_svd(random_state=s)
                     itself (bitwise for the same seed on the same BLAS).
The reference's own tests hold no golden U/s/V (tests/test_03_era5_svd.py:
165-176 checks shapes only), so SVD *values* are pinned by numpy/scikit-learn
outputs generated here by tests/golden/make_golden.py, not by reference
fixtures.
"""

from __future__ import annotations

import numpy as np
from scipy import linalg as sla

__all__ = [
    "standardize",
    "flatten",
    "delay_embed",
    "delay_labels",
    "preprocess",
    "svd_standard",
    "svd_randomized",
    "svd_flip",
    "mock_era5",
    "lowrank_matrix",
]


# --------------------------------------------------------------------------
# pre-processing (defines X exactly)
# --------------------------------------------------------------------------
def standardize(data: np.ndarray, axis: int = 0, scale: bool = True):
    """Mean-centre (and optionally scale) along ``axis`` (time).

    slice_tools.py:171-179: ``mean = data.mean(dim)``; ``data = data - mean``;
    ``std = data.std(dim)`` of the *centred* data with ddof=0; ``data / std``.
    dtype is preserved (float32 in -> float32 arithmetic, as xarray does).
    Returns (data, mean, std_or_None).
    """
    mean = data.mean(axis=axis, keepdims=True, dtype=data.dtype)
    out = data - mean
    if scale:
        std = out.std(axis=axis, keepdims=True, dtype=data.dtype)
        out = out / std
        return out, np.squeeze(mean, axis=axis), np.squeeze(std, axis=axis)
    return out, np.squeeze(mean, axis=axis), None


def flatten(variables: dict[str, np.ndarray], has_time: bool = True):
    """Stack (level, latitude, longitude) -> space, variables along space.

    slice_tools.py:311,323-336: per variable an array (time, level, lat, lon)
    becomes (space, time) with level slowest and longitude fastest; the
    variables are concatenated along space in dict order.  Row index is
    ``v*m_v + ((l*n_lat + i)*n_lon + j)``.
    Returns (X, original_variable_labels).
    """
    mats, labels = [], []
    for name, arr in variables.items():
        if has_time:
            t = arr.shape[0]
            mats.append(np.moveaxis(arr, 0, -1).reshape(-1, t))
        else:
            mats.append(arr.reshape(-1))
        labels.append(np.repeat(name, mats[-1].shape[0]))
    return np.concatenate(mats, axis=0), np.concatenate(labels)


def delay_embed(X: np.ndarray, d: int) -> np.ndarray:
    """Delay embedding, slice_tools.py:182-211.

    Row ``k*m + s``, column ``t`` of the result equals ``X[s, t+k]``
    (k = 0 is the oldest snapshot); shape (d*m, n-d+1); the reference's
    result is F-ordered (it ends with ``.T``), so is this one.
    """
    if X.ndim != 2:
        raise ValueError("Input array must be 2D.")
    if not isinstance(d, (int, np.integer)) or isinstance(d, bool) or d <= 0:
        raise ValueError("Delay must be an integer greater than 0.")
    m, n = X.shape
    nt = n - d + 1
    out = np.empty((d * m, nt), dtype=X.dtype, order="F")
    for k in range(d):
        out[k * m : (k + 1) * m, :] = X[:, k : k + nt]
    return out


def delay_labels(m: int, d: int) -> np.ndarray:
    """``delay`` coordinate: block k carries delay d-1-k (slice_tools.py:265-268)."""
    return np.repeat(np.flip(np.arange(d)), m)


def preprocess(
    variables: dict[str, np.ndarray],
    mean_center: bool,
    scale: bool,
    delay_embedding: int,
):
    """era5_svd.py:389-414 on plain arrays.

    Returns (X, X_mean, X_std): X is (d*m, n-d+1).  The reference keeps
    X_mean/X_std only when mean-centring was requested *and* d > 1
    (era5_svd.py:400,412-414) -- reproduced, quirk included.
    """
    means, stds = {}, {}
    if mean_center:
        out = {}
        for k, a in variables.items():
            out[k], means[k], stds[k] = standardize(a, 0, scale=scale)
        variables = out
    X, _ = flatten(variables)
    X = delay_embed(X, delay_embedding)
    X_mean = X_std = None
    if mean_center and delay_embedding > 1:
        fm, _ = flatten(means, has_time=False)
        X_mean = np.concatenate([fm] * delay_embedding)
        if scale:
            fs, _ = flatten(stds, has_time=False)
            X_std = np.concatenate([fs] * delay_embedding)
    return X, X_mean, X_std


# --------------------------------------------------------------------------
# the SVD boundary (era5_svd.py:230-263)
# --------------------------------------------------------------------------
def svd_standard(X: np.ndarray, n_components: int):
    """era5_svd.py:251-254: LAPACK gesdd, then slice the first k."""
    U, s, V = np.linalg.svd(X, full_matrices=False)
    return U[:, :n_components], s[:n_components], V[:n_components, :]


def svd_flip(u: np.ndarray, v: np.ndarray):
    """u-based sign convention (extmath.py:935-943): the largest-|.| entry
    of every column of u becomes positive."""
    idx = np.argmax(np.abs(u), axis=0)
    signs = np.sign(u[idx, np.arange(u.shape[1])])
    return u * signs[np.newaxis, :], v * signs[:, np.newaxis]


def svd_randomized(
    X: np.ndarray,
    n_components: int,
    n_oversamples: int = 10,
    n_iter="auto",
    power_iteration_normalizer: str = "auto",
    random_state=None,
    omega: np.ndarray | None = None,
    flip_sign: bool = True,
):
    """Restatement of sklearn's randomized_svd for m >= n (transpose='auto'
    is False there, extmath.py:562-563) and for m < n (operates on X.T).

    ``omega`` (n x (k+p)) may be supplied instead of ``random_state`` so a
    GPU run can use exactly the same test matrix.
    """
    m, n = X.shape
    n_random = n_components + n_oversamples
    if n_iter == "auto":
        n_iter = 7 if n_components < 0.1 * min(X.shape) else 4
    transpose = m < n
    M = X.T if transpose else X
    if omega is None:
        rs = (
            random_state
            if isinstance(random_state, np.random.RandomState)
            else np.random.RandomState(random_state)
        )
        omega = rs.normal(size=(M.shape[1], n_random))
    Q = omega.astype(M.dtype, copy=False)
    if power_iteration_normalizer == "auto":
        power_iteration_normalizer = "none" if n_iter <= 2 else "LU"
    if power_iteration_normalizer == "LU":
        norm = lambda a: sla.lu(a, permute_l=True, check_finite=False)[0]  # noqa: E731
    elif power_iteration_normalizer == "QR":
        norm = lambda a: sla.qr(a, mode="economic", check_finite=False)[0]  # noqa: E731
    else:
        norm = lambda a: a  # noqa: E731
    for _ in range(n_iter):
        Q = norm(M @ Q)
        Q = norm(M.T @ Q)
    Q, _ = sla.qr(M @ Q, mode="economic", check_finite=False)
    B = Q.T @ M
    Uhat, s, Vt = sla.svd(B, full_matrices=False, lapack_driver="gesdd")
    U = Q @ Uhat
    if flip_sign:
        if not transpose:
            U, Vt = svd_flip(U, Vt)
        else:  # flip on the rows of Vt (extmath.py:598-600)
            vt_t, u_t = svd_flip(np.ascontiguousarray(Vt.T), np.ascontiguousarray(U.T))
            Vt, U = vt_t.T, u_t.T
    if transpose:
        return Vt[:n_components, :].T, s[:n_components], U[:, :n_components].T
    return U[:, :n_components], s[:n_components], Vt[:n_components, :]


# --------------------------------------------------------------------------
# fixtures generators
# --------------------------------------------------------------------------
def mock_era5(n_time: int, variables, levels, seed: int, dtype=np.float64):
    """Seeded restatement of create_mock_data.py:26-155 (5 deg grid, 36x72).

    The reference draws from the unseeded global ``np.random``; here the same
    draws come from ``RandomState(seed)`` in the same order.
    Returns (dict var -> (time, level, lat, lon), lats, lons).
    """
    rs = np.random.RandomState(seed)
    lats = np.arange(90, -90, -5.0)
    lons = np.arange(-180, 180, 5.0)
    shape = (n_time, len(levels), len(lats), len(lons))
    out = {}
    for var in variables:
        if var == "temperature":
            data = rs.rand(*shape) * 30 + 250
            for i, level in enumerate(levels):
                data[:, i, :, :] -= (1000 - level) / 100
            data = data * np.cos(np.radians(lats))[None, None, :, None]
        elif "wind" in var:
            data = rs.rand(*shape) * 20 - 10
        else:
            data = rs.rand(*shape) * 100
        out[var] = data.astype(dtype)
    return out, lats, lons


def lowrank_matrix(m: int, n: int, rank: int, seed: int, decay=0.9, s0=100.0,
                   noise=0.01, dtype=np.float32, center=True):
    """Structured test matrix of SURVEY.md section 8(d): X = A diag(sigma) B^T + eps,
    sigma_i = s0*decay^i, then row-centred; F-ordered like the reference's X."""
    rs = np.random.RandomState(seed)
    A = rs.standard_normal((m, rank)) / np.sqrt(m)
    B = rs.standard_normal((n, rank)) / np.sqrt(n)
    sig = s0 * decay ** np.arange(rank)
    X = (A * sig) @ B.T  # columns of A, B ~orthonormal => singular values ~ sig
    X = X + noise * rs.standard_normal((m, n)) / np.sqrt(m)
    if center:
        X = X - X.mean(axis=1, keepdims=True)
    return np.asfortranarray(X.astype(dtype))
